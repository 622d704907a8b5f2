// g2g_device.h -- device-side view of one group-to-group DP (internal to libg2g.so).
//
// HBM layout (all per-DP regions are carved out of one arena by the host, g2g_engine.hip):
//   inputs   : exactly the g2g_problem arrays (include/g2g.h), copied verbatim, 16-B aligned
//   state    : the reference's diagonal-indexed row buffer hh[0..Noll) + f1/f2 (fwd2c.h:90-98) kept as
//              structure-of-arrays over the diagonal index r = n - m, r in [lw-1, up+1]:
//                val[X][r]  f64      dir[X][r] u8      X in {H, G, G2, F, F2}
//                dla[X][k][r], dlb[X][k][r]  {glen,nins} pairs, ENTRY-major so that the k-th entries of
//                neighbouring diagonals are contiguous (the anti-diagonal sweep reads them coalesced)
//   trace    : one byte per in-band cell, anti-diagonal major: trace[(d - d0) * tstride + (m - mlo(d))]
#ifndef G2G_DEVICE_H_
#define G2G_DEVICE_H_
#include <stdint.h>

// The library is built from several translation units (prrn_aln_amd/build.py compiles them side by side): every unit sees every
// device function, but EMITS only the kernels of its own group and merely declares the others -- the host stubs are ordinary
// symbols, resolved at link time.  G2G_TU_ALL puts everything into one unit (what a one-off resource listing wants).
//   G2G_TU_V1  g2g_forward_kernel, g2g_traceback_kernel, g2g_spscore_kernel + the f3 / pairsum kernels (the engine's own unit)
//   G2G_TU_V2  the 8-lanes-per-cell strips and the tile-mode helpers     G2G_TU_V3  _hf strips
//   G2G_TU_V6  _pf strips                                               G2G_TU_V78 strips of the records without gap profiles
#ifdef G2G_TU_ALL
#define G2G_TU_V1 1
#define G2G_TU_V2 1
#define G2G_TU_V3 1
#define G2G_TU_V6 1
#define G2G_TU_V78 1
#endif

struct DevSide {
    int many, len, left, right, nils, nelm, felm, hetero;
    int maxlist;         // longest static gap-profile list incl. its terminator (any view, any position)
    int r_from_t;        // every r list = [optional head {glen 0}] + the t list of the position with glen + 1 (how Gfq builds it,
                         // gfreq.cc:218-226): checked at pack time; lets a kernel keep s and t only
    const uint8_t *seq;
    const double  *weight;
    const double  *pseq;
    const double  *thk;
    const int     *off[3];
    const int     *glen[3];
    const double  *freq[3];
    const double  *gapdens;
    const double  *postgapdens;
    double         sumwt;            // Seq::sumwt (read by the stt?? statistics)
    int            npfq, pfq_step;   // exon-boundary annotation (SigII::pfq): positions ascending, density per position
    const int     *pfq_pos;
    const double  *pfq_dns;
};

// XBT / XBL: running records of the top / left boundary chains (2 ping-pong slots each)
enum { XH = 0, XG = 1, XG2 = 2, XF = 3, XF2 = 4, XBT = 5, XBL = 6, NX = 7 };

struct DevProb {
    int kind;            // 0 DPunit, 1 _hf, 2 _pf, 3 _nv   (reference src/dpunit.h:31-51)
    int rect;            // 1: the rectangular engine Fwd2c::forwardA (the _ALN modes, `-A`; src/fwd2c.h:232-356) -- g2g_forward_kernel, DPunit only
    int noll, sim2_kind, crg2_kind, codonk1, lw, up, width;
    double basic_gop, weighted_gop, u, u2divu1, v2divv1;
    int dvsp;            // PwdB::DvsP (0: nucleotide x nucleotide)
    double spb_fact;     // SpbFact; > 0 with both sides annotated: the intron-position bonus (fwd2c.h:446-452) is live
    int nbonus;          // cells of the DP that receive it, ascending in (m, n): precomputed by the host from PfqItr's walk
    const int *bon_m, *bon_n;
    const double *bon_h, *bon_mx;     // what is added to H, and to the best non-diagonal record, of that cell
    const double *simmtx;
    int simdim;
    int capa, capb;      // entries per dynamic list (hetero+1) / members (nv)
    DevSide a, b;
    // state, indexed [r - (lw-1)]
    double  *val[NX];
    uint8_t *dir[NX];
    int2    *dla[NX];
    int2    *dlb[NX];
    int     *glb[NX];    // _hf: running gap length of b ; _nv: gla/glb member arrays [(an+bn)][width]
    int      spw;        // stride of the two scratch lists g2g_spscore_kernel keeps in dla/dlb[XH] (width for v1 DPs, else 1)
    // v2 kernel: strip boundary (one record per column) and left boundary chain (one per row) in HBM
    void    *v2_rowH, *v2_rowG, *v2_rowG2, *v2_colH;   // rowX: 3 buffers of v2_rowstride records
    void    *v2_cbH, *v2_cbF, *v2_cbF2;                // per row: records at the current column-block edge
    int      v2_rowstride;
    double  *v2_sim;       // sim2 of every in-band cell, row-major, row m at v2_rowoff[m - a.left] (- nlo(m))
    long long *v2_rowoff;
    int      v2_ok;      // 1: handled by g2g_forward_kernel_v2
    // trace
    uint8_t *trace;
    int      tstride;
    int      d0, d1;     // first / last anti-diagonal (m + n) holding cells
    // outputs
    double  *score;
    int     *ntrace;
    int2    *otrace;     // up to tcap records {m, n}
    int      tcap;
    long long cells;
};


#if defined(__HIPCC__)
#define G2G_HD __host__ __device__
#else
#define G2G_HD
#endif
// ---- band geometry ---------------------------------------------------------------------------
G2G_HD inline int g2g_floordiv2(int x) { return x >= 0 ? x / 2 : -((-x + 1) / 2); }
G2G_HD inline int g2g_ceildiv2(int x) { return x >= 0 ? (x + 1) / 2 : -((-x) / 2); }
// rows holding a cell on anti-diagonal d = m + n, where row m spans
// n in [max(m+lw, b.left), min(m+up+1, b.right))   (reference src/fwd2c.h:373-374)
G2G_HD inline void diag_rows(int d, int al, int ar, int bl, int br, int lw, int up, int *mlo, int *mhi)
{
    int lo = al, hi = ar - 1, t;
    t = d - br + 1;              if (t > lo) lo = t;       // n <= b.right - 1
    t = g2g_ceildiv2(d - up);    if (t > lo) lo = t;       // n - m <= up
    t = d - bl;                  if (t < hi) hi = t;       // n >= b.left
    t = g2g_floordiv2(d - lw);   if (t < hi) hi = t;       // n - m >= lw
    *mlo = lo; *mhi = hi;
}

#endif
