// g2g_kernels_v2.hip -- the production forward kernel for the gap-profile engines (_hf, _pf; Noll 2/3).
//
// Same recurrence, same arithmetic order as g2g_kernels.hip (Fwd2c::forwardB, reference
// src/fwd2c.h:359-482 + src/fwd2c.cc:152-251 + src/gfreq.cc:493-605), different machine mapping:
//
//  * ROW STRIPS.  A workgroup sweeps one DP in strips of R = blockDim/8 rows.  Inside a strip row t runs
//    one column behind row t-1 (skewed wavefront, one barrier per step), so every record a cell needs
//    was produced one or two steps earlier by the row itself or by the row above.
//  * STATE IN LDS.  Each row keeps small rings of its own DP records -- H: 3 slots (corners n-1, n, n+1),
//    G: 2, F: 1 (+G2: 2, F2: 1 for the double-affine penalty) -- in LDS, dynamic gap-state lists packed
//    as 16+16-bit {glen, nins}.  Nothing but the strip boundary (the last row's H/G records, one record
//    per column) and the trace bytes goes back to HBM; v1 streamed every record through HBM.
//  * A TEAM OF 8 LANES PER CELL.  The six gap-open costs of a cell (two diagonal, two vertical, two
//    horizontal list merges) and the column score are independent, as are the list updates of the
//    three produced records: lanes of a team take one each, exchange scalars with wave shuffles, and
//    all lanes replay the (cheap) scalar decision logic.  This cuts the dependent-access chain per cell
//    ~10x, which is what bounds a single DP's sweep time.
// The boundary chains (initB) run once in a prologue and are parked in HBM (top row: RowH, left column:
// ColH).  Trace bytes and the backtrack kernel are shared with v1.
#include <hip/hip_runtime.h>

#define TEAM 8
#define DL_END 0xFFFFu                       // packed terminator glen (INT_MAX in the reference)
#define DL_GUARD 4096                        // no list is this long: a corrupted one must not hang the wave

// ---- LDS record: {f64 val; i32 dir; i32 glb; u32 dla[capa]; u32 dlb[capb]} -----------------------
struct LRec { char *p; };
__device__ __forceinline__ double &lval(LRec r) { return *(double *) r.p; }
__device__ __forceinline__ int &ldir(LRec r) { return *(int *) (r.p + 8); }
__device__ __forceinline__ int &lglb(LRec r) { return *(int *) (r.p + 12); }
__device__ __forceinline__ unsigned *ldla(LRec r) { return (unsigned *) (r.p + 16); }
__device__ __forceinline__ unsigned *ldlb(LRec r, int capa) { return (unsigned *) (r.p + 16) + capa; }

__device__ __forceinline__ void team_sync()
{   // lanes of a team live in one wave: ordering LDS traffic between phases is a compiler matter only
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// GapLenSD, gfreq.h:67, on a packed list
__device__ __forceinline__ int p_gaplen(int g, const unsigned *dl)
{
    int k = 0;
    while (g >= (int) (dl[k + 1] >> 16) && k < DL_GUARD) ++k;
    return g + (int) (dl[k] & 0xFFFFu);
}
// newgap(cf, dlc, df, dld), gfreq.cc:507-521
__device__ double p_newgap4(const SList cf, const unsigned *dlc, const SList df, const unsigned *dld)
{
    double g = 0;
    int ci = 0;
    for (int di = 0; df.glen[di] >= 0; ++di) {
        const int j = p_gaplen(df.glen[di], dld);
        for ( ; cf.glen[ci] >= 0; ++ci) {
            const int i = p_gaplen(cf.glen[ci], dlc);
            if (i >= j) break;
        }
        if (cf.glen[ci] < 0) break;
        g += cf.freq[ci] * df.freq[di];
    }
    return g;
}
// newgap1 / newgap2, maln.h:296-308 (+ newgapc/newgapd :280-291)
__device__ double p_newgap1(const DevProb &P, const SList acf, const unsigned *dla, int glb)
{
    if (acf.glen[0] < 0) return 0;
    if (acf.glen[1] >= 0) {
        for (int ci = 0; acf.glen[ci] >= 0; ++ci)          // newgap(cf, dlc, j), gfreq.cc:523-532
            if (p_gaplen(acf.glen[ci], dla) >= glb) return P.weighted_gop * acf.freq[ci];
        return P.weighted_gop * 0.;
    }
    return ((int) (dla[0] & 0xFFFFu) + acf.glen[0] >= glb) ? (P.weighted_gop * acf.freq[0]) : 0;
}
__device__ double p_newgap2(const DevProb &P, const SList adf, int glb, const unsigned *dla)
{
    if (adf.glen[0] < 0) return 0;
    if (adf.glen[1] >= 0) {                                 // newgap(df, i, dld), gfreq.cc:534-545
        double g = 0;
        int k = 0;
        for (int di = 0; adf.glen[di] >= 0; ++di) {
            while (adf.glen[di] >= (int) (dla[k + 1] >> 16) && k < DL_GUARD) ++k;
            if (glb < adf.glen[di] + (int) (dla[k] & 0xFFFFu)) break;
            g += adf.freq[di];
        }
        return P.weighted_gop * g;
    }
    return (glb >= (int) (dla[0] & 0xFFFFu) + adf.glen[0]) ? (P.weighted_gop * adf.freq[0]) : 0;
}
// newdelta(dlt, df, dln, 1), gfreq.cc:570-587; up to two destinations (the record itself and, when that
// record wins the cell, the new H).  dst may alias src (stores trail the loads they could affect).
__device__ void p_newdelta(unsigned *dst, unsigned *dst2, const SList df, const unsigned *src)
{
    int kd = 0, ks = 0;
    unsigned tg = 0, tn = 0;
    for (int di = 0; df.glen[di] >= 0; ++di) {
        const int g = df.glen[di];
        if (g >= (int) (src[ks] >> 16)) {
            while (g >= (int) (src[ks + 1] >> 16) && ks < DL_GUARD) ++ks;
            const unsigned sn = src[ks] & 0xFFFFu;
            if (sn > tn) {
                const unsigned e = (tg << 16) | tn;
                dst[kd] = e; if (dst2) dst2[kd] = e;
                ++kd;
                tn = sn;
                tg = (unsigned) (g + 1);
            }
        }
    }
    const unsigned e = (tg << 16) | tn;
    dst[kd] = e; dst[kd + 1] = DL_END << 16;
    if (dst2) { dst2[kd] = e; dst2[kd + 1] = DL_END << 16; }
}
// incdelta(dlt, dln, 1), gfreq.cc:598-605
__device__ void p_incdelta(unsigned *dst, unsigned *dst2, const unsigned *src)
{
    int k = 0;
    for ( ; k < DL_GUARD; ++k) {
        unsigned e = src[k];
        if ((e >> 16) == DL_END) { dst[k] = e; if (dst2) dst2[k] = e; break; }
        e += 1;                                             // nins + 1
        dst[k] = e; if (dst2) dst2[k] = e;
    }
}
__device__ void p_copylist(unsigned *dst, const unsigned *src)
{
    for (int k = 0; ; ++k) { const unsigned e = src[k]; dst[k] = e; if ((e >> 16) == DL_END) break; }
}
__device__ __forceinline__ void p_clearlist(unsigned *d) { d[0] = 0; d[1] = DL_END << 16; }

// global <-> LDS record moves by one team (lane j moves dwords j, j+8, ...)
__device__ __forceinline__ void rec_g2l(LRec dst, const unsigned *src, int ndw, int lane)
{
    unsigned *d = (unsigned *) dst.p;
    for (int k = lane; k < ndw; k += TEAM) d[k] = src[k];
}
__device__ __forceinline__ void rec_l2g(unsigned *dst, LRec src, int ndw, int lane)
{
    const unsigned *s = (const unsigned *) src.p;
    for (int k = lane; k < ndw; k += TEAM) dst[k] = s[k];
}

struct V2Geom {
    int capa, capb, recsz, ndw;       // list capacities, record bytes / dwords
    int nslot;                        // ring slots per row: 6 (Noll 2) or 9 (Noll 3)
    int R;                            // rows per strip
    char *lds;
    // slot ids inside a row
    __device__ __forceinline__ LRec row(int t, int slot) const { LRec r; r.p = lds + ((size_t) t * nslot + slot) * recsz; return r; }
    __device__ __forceinline__ LRec extra(int k) const { LRec r; r.p = lds + ((size_t) R * nslot + k) * recsz; return r; }
};
// ring slots: H corner c -> c mod 3 (0..2); G corner c -> 3 + (c & 1); F -> 5; G2 -> 6 + (c & 1); F2 -> 8
__device__ __forceinline__ int mod3(int c) { return ((c % 3) + 3) % 3; }
#define SLOT_H(c) (mod3(c))
#define SLOT_G(c) (3 + ((c) & 1))
#define SLOT_F 5
#define SLOT_G2(c) (6 + ((c) & 1))
#define SLOT_F2 8
// extras: 0 black; staging of the strip's first row, whose upper neighbours live in HBM: 1,2 H corners
// (corner c in slot c & 1: this step's Hup is the next step's Hdiag), 3 Gup, 4 G2up
enum { EX_BLACK = 0, EX_H0 = 1, EX_H1 = 2, EX_GU = 3, EX_G2U = 4, EX_N = 5 };

template <int KIND>
__device__ __forceinline__ void lrec_black(LRec r, int capa)
{
    lval(r) = NEVSEL; ldir(r) = 0; lglb(r) = 0;
    p_clearlist(ldla(r));
    if (KIND == 2) p_clearlist(ldlb(r, capa));
}

// gap-open cost of record rc for move d3 -- Fwd2c<_hf/_pf>::gapopen (fwd2c.cc:152-160, 203-212), one
// list merge per call; `part` selects the first/second merge of the _pf diagonal case
template <int KIND>
__device__ double v2_gapopen(const DevProb &P, LRec rc, int capa, int m, int n, int d3, int part)
{
    if (KIND == 1) {
        if (d3 > 0) return p_newgap1(P, gfq_at(P.a, 0, m), ldla(rc), lglb(rc));
        return p_newgap2(P, gfq_at(P.a, d3 == 0 ? 1 : 2, m), lglb(rc), ldla(rc));
    } else {
        const unsigned *dla = ldla(rc), *dlb = ldlb(rc, capa);
        if (d3 == 0) {
            if (part == 0) return p_newgap4(gfq_at(P.a, 0, m), dla, gfq_at(P.b, 1, n), dlb) * P.basic_gop;
            return p_newgap4(gfq_at(P.b, 0, n), dlb, gfq_at(P.a, 1, m), dla) * P.basic_gop;
        } else if (d3 > 0) return p_newgap4(gfq_at(P.a, 0, m), dla, gfq_at(P.b, 2, n), dlb) * P.basic_gop;
        return p_newgap4(gfq_at(P.b, 0, n), dlb, gfq_at(P.a, 2, m), dla) * P.basic_gop;
    }
}

// ---- prologue: the boundary chains of initB (fwd2c.h:138-176), one lane each ----------------------
template <int KIND>
__device__ void v2_chain_top(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *rowH)
{
    const DevSide &a = P.a, &b = P.b;
    int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;
    const int nlast = a.left + rrt, ai = a.left - 1;
    lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG;          // origin
    { const unsigned *s = (const unsigned *) s0.p; unsigned *d = rowH + (size_t) b.left * G.ndw; for (int k = 0; k < G.ndw; ++k) d[k] = s[k]; }
    LRec prv = s0, cur = s1;
    for (int n = b.left + 1; n <= nlast; ++n) {
        const int bi = n - 1;
        const double pub = unpb(P, bi, ai);
        double gnp = (KIND == 1) ? v2_gapopen<KIND>(P, prv, G.capa, ai, bi, -1, 0)
                                 : v2_gapopen<KIND>(P, prv, G.capa, ai, bi, -1, 0);
        gnp = (n - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
        // update(h, h-1, ..., -1): fwd2c.cc:175-178 / 226-229
        ldir(cur) = isvert(ldir(prv)) ? D_NEWH : D_HORI;
        if (KIND == 1) { p_incdelta(ldla(cur), 0, ldla(prv)); lglb(cur) = 0; }
        else { p_newdelta(ldlb(cur, G.capa), 0, gfq_at(P.b, 1, bi), ldlb(prv, G.capa)); p_incdelta(ldla(cur), 0, ldla(prv)); lglb(cur) = 0; }
        lval(cur) = lval(prv) + gnp;
        { const unsigned *s = (const unsigned *) cur.p; unsigned *d = rowH + (size_t) n * G.ndw; for (int k = 0; k < G.ndw; ++k) d[k] = s[k]; }
        LRec t = prv; prv = cur; cur = t;
    }
}
template <int KIND>
__device__ void v2_chain_left(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *colH)
{
    const DevSide &a = P.a, &b = P.b;
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int mlast = b.left - rrl, bi = b.left - 1;
    lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG;
    LRec prv = s0, cur = s1;
    for (int m = a.left + 1; m <= mlast; ++m) {
        const int ai = m - 1;
        const double pua = unpa(P, ai, bi);
        double gnp = v2_gapopen<KIND>(P, prv, G.capa, ai, bi, 1, 0);
        gnp = (m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
        // update(h, h+1, ..., 1): fwd2c.cc:170-173 / 222-225
        ldir(cur) = ishori(ldir(prv)) ? D_NEWV : D_VERT;
        const int g1 = lglb(prv) + 1;
        p_newdelta(ldla(cur), 0, gfq_at(P.a, 1, ai), ldla(prv));
        if (KIND == 1) lglb(cur) = g1;
        else { p_incdelta(ldlb(cur, G.capa), 0, ldlb(prv, G.capa)); lglb(cur) = 0; }
        lval(cur) = lval(prv) + gnp;
        { const unsigned *s = (const unsigned *) cur.p; unsigned *d = colH + (size_t) (m - a.left) * G.ndw; for (int k = 0; k < G.ndw; ++k) d[k] = s[k]; }
        LRec t = prv; prv = cur; cur = t;
    }
}

// ---- one cell by one team ------------------------------------------------------------------------
struct CellSrc { LRec hd, hu, gu, g2u, hl, fl, f2l; };   // records the cell reads
struct CellDst { LRec h, g, g2, f, f2; };                 // records it writes

template <int KIND, bool NOLL3>
__device__ void v2_cell(const DevProb &P, const V2Geom &G, int m, int n, int lane,
                        const CellSrc &S, const CellDst &D, bool do_vert, bool do_hori, uint8_t *tr)
{
    const DevSide &a = P.a, &b = P.b;
    const int capa = G.capa;
    // ---- phase A: one independent cost per lane -----------------------------------------------
    double r = 0;
    switch (lane) {
    case 0: r = v2_gapopen<KIND>(P, S.hd, capa, m, n, 0, 0); break;
    case 1: if (KIND == 2) r = v2_gapopen<KIND>(P, S.hd, capa, m, n, 0, 1); break;
    case 2: if (do_vert) r = v2_gapopen<KIND>(P, S.gu, capa, m, n, 1, 0); break;
    case 3: if (do_vert) r = v2_gapopen<KIND>(P, S.hu, capa, m, n, 1, 0); break;
    case 4: if (do_hori) r = v2_gapopen<KIND>(P, S.fl, capa, m, n, -1, 0); break;
    case 5: if (do_hori) r = v2_gapopen<KIND>(P, S.hl, capa, m, n, -1, 0); break;
    case 6: r = sim2(P, m, n); break;
    default: if (NOLL3 && do_vert) r = v2_gapopen<KIND>(P, S.g2u, capa, m, n, 1, 0); break;
    }
    double r8 = 0;
    if (NOLL3 && do_hori && lane == 1 && KIND == 1) r8 = v2_gapopen<KIND>(P, S.f2l, capa, m, n, -1, 0);
    if (NOLL3 && do_hori && lane == 6 && KIND == 2) r8 = v2_gapopen<KIND>(P, S.f2l, capa, m, n, -1, 0);
    const double c_d0 = __shfl(r, 0, TEAM), c_d1 = __shfl(r, 1, TEAM);
    const double c_gnpv = __shfl(r, 2, TEAM), c_gopv = __shfl(r, 3, TEAM);
    const double c_gnph = __shfl(r, 4, TEAM), c_goph = __shfl(r, 5, TEAM);
    const double dab = __shfl(r, 6, TEAM);
    const double c_gnpv2 = NOLL3 ? __shfl(r, 7, TEAM) : 0;
    const double c_gnph2 = NOLL3 ? __shfl(r8, KIND == 1 ? 1 : 6, TEAM) : 0;
    // ---- scalar decisions, replayed by every lane (fwd2c.h:395-453) ------------------------------
    double gop = (KIND == 2) ? c_d0 + c_d1 : c_d0;
    const double hval = lval(S.hd) + (dab + gop);
    const int hdir = isdiag(ldir(S.hd)) ? D_DIAG : D_NEWD;
    int bits = 0, win = 0;                          // win: 0 diag, 1 G, 2 G2, 3 F, 4 F2
    double mxval = NEVSEL;                          // mx = g: at the first row G is a black record
    double gval = 0, g2val = 0, fval = 0, f2val = 0;
    int gdir = 0, g2dir = 0, fdir = 0, f2dir = 0;
    bool g_from_h = false, g2_from_h = false, f_from_h = false, f2_from_h = false;
    if (do_vert) {
        int nf = m + P.lw; if (nf < b.left) nf = b.left;
        const double pua = unpa(P, m, a.nils ? n : nf);
        const double gnp = c_gnpv;
        gop = c_gopv;
        const bool hu_nv = !isvert(ldir(S.hu));
        g_from_h = hu_nv && (lval(S.hu) + gop > lval(S.gu) + gnp);
        const LRec gs = g_from_h ? S.hu : S.gu;
        gdir = ishori(ldir(gs)) ? D_NEWV : D_VERT;
        gval = lval(gs) + (g_from_h ? gop : gnp);
        gval += pua;
        if (!g_from_h) bits |= T_GEXT;
        mxval = gval; win = 1;
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c_gnpv2;
            gop = P.v2divv1 * gop;
            g2_from_h = hu_nv && (lval(S.hu) + gop > lval(S.g2u) + gnp2);
            const LRec gs2 = g2_from_h ? S.hu : S.g2u;
            g2dir = ishori(ldir(gs2)) ? D_NEWV : D_VERT;
            g2val = lval(gs2) + (g2_from_h ? gop : gnp2);
            g2val += P.u2divu1 * pua;
            if (!g2_from_h) bits |= T_G2EXT;
            if (g2val > mxval) { mxval = g2val; win = 2; }
        }
    } else {
        // first row: G is never touched; mx starts as the (black) G record
        mxval = NEVSEL; win = 1;
    }
    if (do_hori) {
        const double pub = unpb(P, n, m);
        const double gnp = c_gnph;
        gop = c_goph;
        const bool hl_nh = !ishori(ldir(S.hl));
        f_from_h = hl_nh && (lval(S.hl) + gop > lval(S.fl) + gnp);
        const LRec fs = f_from_h ? S.hl : S.fl;
        fdir = isvert(ldir(fs)) ? D_NEWH : D_HORI;
        fval = lval(fs) + (f_from_h ? gop : gnp);
        fval += pub;
        if (!f_from_h) bits |= T_FEXT;
        if (fval >= mxval) { mxval = fval; win = 3; }
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c_gnph2;
            gop = P.v2divv1 * gop;
            f2_from_h = hl_nh && (lval(S.hl) + gop > lval(S.f2l) + gnp2);
            const LRec fs2 = f2_from_h ? S.hl : S.f2l;
            f2dir = isvert(ldir(fs2)) ? D_NEWH : D_HORI;
            f2val = lval(fs2) + (f2_from_h ? gop : gnp2);
            f2val += P.u2divu1 * pub;
            if (!f2_from_h) bits |= T_F2EXT;
            if (f2val >= mxval) { mxval = f2val; win = 4; }
        }
    }
    const bool mx_wins = mxval > hval;                 // diagonal wins ties (fwd2c.h:453)
    if (!mx_wins) win = 0;
    if (!do_vert && win == 1) win = 0;                 // (black G can never win: NEVSEL > hval is false)
    // ---- phase B: list updates, one per lane; the winner's lists are also written to the new H ----
    team_sync();
    const LRec gs = g_from_h ? S.hu : S.gu, gs2 = g2_from_h ? S.hu : S.g2u;
    const LRec fs = f_from_h ? S.hl : S.fl, fs2 = f2_from_h ? S.hl : S.f2l;
    const SList at = gfq_at(P.a, 1, m);
    if (KIND == 2) {
        const SList bt = gfq_at(P.b, 1, n);
        switch (lane) {
        case 0: if (win == 0) p_newdelta(ldla(D.h), 0, at, ldla(S.hd)); break;
        case 1: if (win == 0) p_newdelta(ldlb(D.h, capa), 0, bt, ldlb(S.hd, capa)); break;
        case 2: if (do_vert) p_newdelta(ldla(D.g), win == 1 ? ldla(D.h) : 0, at, ldla(gs)); break;
        case 3: if (do_vert) p_incdelta(ldlb(D.g, capa), win == 1 ? ldlb(D.h, capa) : 0, ldlb(gs, capa)); break;
        case 4: if (do_hori) p_newdelta(ldlb(D.f, capa), win == 3 ? ldlb(D.h, capa) : 0, bt, ldlb(fs, capa)); break;
        case 5: if (do_hori) p_incdelta(ldla(D.f), win == 3 ? ldla(D.h) : 0, ldla(fs)); break;
        case 6: if (NOLL3 && do_vert) { p_newdelta(ldla(D.g2), win == 2 ? ldla(D.h) : 0, at, ldla(gs2));
                                        p_incdelta(ldlb(D.g2, capa), win == 2 ? ldlb(D.h, capa) : 0, ldlb(gs2, capa)); } break;
        default: if (NOLL3 && do_hori) { p_newdelta(ldlb(D.f2, capa), win == 4 ? ldlb(D.h, capa) : 0, bt, ldlb(fs2, capa));
                                         p_incdelta(ldla(D.f2), win == 4 ? ldla(D.h) : 0, ldla(fs2)); } break;
        }
    } else {
        switch (lane) {
        case 0: if (win == 0) p_newdelta(ldla(D.h), 0, at, ldla(S.hd)); break;
        case 2: if (do_vert) p_newdelta(ldla(D.g), win == 1 ? ldla(D.h) : 0, at, ldla(gs)); break;
        case 4: if (do_hori) p_incdelta(ldla(D.f), win == 3 ? ldla(D.h) : 0, ldla(fs)); break;
        case 6: if (NOLL3 && do_vert) p_newdelta(ldla(D.g2), win == 2 ? ldla(D.h) : 0, at, ldla(gs2)); break;
        case 7: if (NOLL3 && do_hori) p_incdelta(ldla(D.f2), win == 4 ? ldla(D.h) : 0, ldla(fs2)); break;
        default: break;
        }
    }
    // scalars of the produced records (lane 0 writes; glb per fwd2c.cc:169,173,177)
    const int glb_g = (KIND == 1 && do_vert) ? lglb(gs) + 1 : 0;
    const int glb_g2 = (KIND == 1 && NOLL3 && do_vert) ? lglb(gs2) + 1 : 0;
    team_sync();
    if (lane == 0) {
        if (do_vert) { lval(D.g) = gval; ldir(D.g) = gdir; lglb(D.g) = glb_g; }
        if (NOLL3 && do_vert) { lval(D.g2) = g2val; ldir(D.g2) = g2dir; lglb(D.g2) = glb_g2; }
        if (do_hori) { lval(D.f) = fval; ldir(D.f) = fdir; lglb(D.f) = 0; }
        if (NOLL3 && do_hori) { lval(D.f2) = f2val; ldir(D.f2) = f2dir; lglb(D.f2) = 0; }
        double hv = hval; int hd = hdir, hg = 0;
        if (win == 1) { hv = gval; hd = gdir; hg = glb_g; }
        else if (win == 2) { hv = g2val; hd = g2dir; hg = glb_g2; }
        else if (win == 3) { hv = fval; hd = fdir; }
        else if (win == 4) { hv = f2val; hd = f2dir; }
        lval(D.h) = hv; ldir(D.h) = hd; lglb(D.h) = hg;
        if (win == 2 || win == 4) bits |= T_SEL2;
        *tr = (uint8_t) (bits | dir2code(hd));
    }
    team_sync();
}

template <int KIND, bool NOLL3>
__device__ void v2_run(const DevProb &P, char *lds)
{
    const DevSide &a = P.a, &b = P.b;
    const int tid = threadIdx.x, lane = tid & (TEAM - 1), team = tid / TEAM;
    V2Geom G;
    G.capa = P.capa; G.capb = (KIND == 2) ? P.capb : 0;
    G.recsz = (16 + 4 * (G.capa + G.capb) + 15) & ~15; G.ndw = G.recsz / 4;
    G.nslot = NOLL3 ? 9 : 6;
    G.R = blockDim.x / TEAM;
    G.lds = lds;
    unsigned *rowH = (unsigned *) P.v2_rowH, *rowG = (unsigned *) P.v2_rowG, *rowG2 = (unsigned *) P.v2_rowG2;
    unsigned *colH = (unsigned *) P.v2_colH;
    const int R = G.R;
    // black record
    if (tid == 0) lrec_black<KIND>(G.extra(EX_BLACK), G.capa);
    // prologue chains: lane 0 of wave 0 (top row) and lane 0 of the last wave (left column)
    if (tid == 0) v2_chain_top<KIND>(P, G, G.extra(EX_H0), G.extra(EX_H1), rowH);
    if (tid == (int) blockDim.x - 64) v2_chain_left<KIND>(P, G, G.row(R - 1, 0), G.row(R - 1, 1), colH);
    __syncthreads();
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const LRec black = G.extra(EX_BLACK);
    for (int m0 = a.left; m0 < a.right; m0 += R) {
        const int m = m0 + team;
        const bool row_ok = m < a.right;
        int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;
        int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
        int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left;          // first row's first column
        int mlast = m0 + R - 1; if (mlast > a.right - 1) mlast = a.right - 1;
        int cend = mlast + P.up + 1; if (cend > b.right) cend = b.right;     // last row's end
        const int nsteps = (cend - cbase) + (mlast - m0) + 1;
        // new rows: F (and F2) start black (reset(f1), reset(f2), fwd2c.h:385-386)
        // (G of the DP's first row is never written and is read as black by the row below, fwd2c.h:401)
        if (lane < G.nslot) lrec_black<KIND>(G.row(team, lane), G.capa);
        if (lane == 0 && G.nslot > TEAM) lrec_black<KIND>(G.row(team, 8), G.capa);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int n = cbase + s - team;
            const bool active = row_ok && n >= nlo && n < nhi;
            if (active) {
                CellSrc S; CellDst D;
                const bool do_vert = m > a.left, do_hori = n > b.left;
                // -- sources -------------------------------------------------------------------
                if (team == 0) {
                    // the row above lives in HBM (previous strip's last row / the top boundary chain)
                    const LRec hs0 = G.extra(EX_H0), hs1 = G.extra(EX_H1), gu = G.extra(EX_GU), g2u = G.extra(EX_G2U);
                    const LRec hcur = (n & 1) ? hs1 : hs0, hnxt = (n & 1) ? hs0 : hs1;
                    if (n == nlo) {                                             // first cell of the row
                        if (n == b.left && m > a.left) rec_g2l(hcur, colH + (size_t) (m - a.left) * G.ndw, G.ndw, lane);
                        else rec_g2l(hcur, rowH + (size_t) n * G.ndw, G.ndw, lane);
                    }
                    const bool up_in = do_vert && (n - (m - 1) <= P.up);       // cell (m-1, n) exists
                    if (up_in || (!do_vert && n + 1 < nhi)) rec_g2l(hnxt, rowH + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                    if (up_in) {
                        rec_g2l(gu, rowG + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                        if (NOLL3) rec_g2l(g2u, rowG2 + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                    }
                    team_sync();
                    S.hd = hcur; S.hu = up_in ? hnxt : black; S.gu = up_in ? gu : black; S.g2u = up_in ? g2u : black;
                } else {
                    S.hd = G.row(team - 1, SLOT_H(n));
                    const bool up_in = (n - (m - 1) <= P.up);
                    S.hu = up_in ? G.row(team - 1, SLOT_H(n + 1)) : black;
                    S.gu = up_in ? G.row(team - 1, SLOT_G(n + 1)) : black;
                    S.g2u = (NOLL3 && up_in) ? G.row(team - 1, SLOT_G2(n + 1)) : black;
                }
                const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
                S.hl = left_in ? G.row(team, SLOT_H(n)) : black;
                S.fl = left_in ? G.row(team, SLOT_F) : black;
                S.f2l = (NOLL3 && left_in) ? G.row(team, SLOT_F2) : black;
                // at the row's first cell F must read black even though the slot is reused in place
                if (!left_in && do_hori) { S.fl = black; S.f2l = black; }
                D.h = G.row(team, SLOT_H(n + 1));
                D.g = G.row(team, SLOT_G(n + 1));
                D.g2 = G.row(team, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
                D.f = G.row(team, SLOT_F);
                D.f2 = G.row(team, NOLL3 ? SLOT_F2 : SLOT_F);
                int mlo, mhi;
                const int d = m + n;
                diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
                uint8_t *tr = P.trace + (size_t) (d - P.d0) * P.tstride + (m - mlo);
                v2_cell<KIND, NOLL3>(P, G, m, n, lane, S, D, do_vert, do_hori, tr);
                // the row below starts at b.left with the left-boundary corner (m+1, b.left) as its
                // diagonal source: park it in this row's H ring where that row will look for it
                if (n == b.left && team + 1 < R && m + 1 < a.right && m + 1 <= m_left_last && (m + 1 + P.lw) <= b.left) {
                    rec_g2l(G.row(team, SLOT_H(b.left)), colH + (size_t) (m + 1 - a.left) * G.ndw, G.ndw, lane);
                }
                // strip boundary: the last row's corners go to HBM for the next strip
                if (team == R - 1 || m == a.right - 1) {
                    rec_l2g(rowH + (size_t) (n + 1) * G.ndw, D.h, G.ndw, lane);
                    rec_l2g(rowG + (size_t) (n + 1) * G.ndw, D.g, G.ndw, lane);
                    if (NOLL3) rec_l2g(rowG2 + (size_t) (n + 1) * G.ndw, D.g2, G.ndw, lane);
                }
                if (m == a.right - 1 && n == b.right - 1 && lane == 0) *P.score = lval(D.h);
            }
            __syncthreads();
        }
    }
}

extern "C" __global__ void __launch_bounds__(G2G_V2_THREADS)
g2g_forward_kernel_v2(const DevProb *probs, const int *idx)
{
    extern __shared__ __attribute__((aligned(16))) char g2g_lds[];
    const DevProb &P = probs[idx[blockIdx.x]];
    if (P.kind == 1) { if (P.noll == 3) v2_run<1, true>(P, g2g_lds); else v2_run<1, false>(P, g2g_lds); }
    else if (P.kind == 2) { if (P.noll == 3) v2_run<2, true>(P, g2g_lds); else v2_run<2, false>(P, g2g_lds); }
}
