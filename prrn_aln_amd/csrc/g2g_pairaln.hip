// g2g_pairaln.hip -- f3, second half: pairwise alignment of two single sequences WITH the path.  Restates Aln2b1::forwardB_ng
// (reference src/fwd2b1.cc:145-279: H, G, F and with -yl3 the long-gap layers G2, F2; similarity = Simmtx lookup, gap costs
// closed form), initB_ng (:64-98), lastB_ng (:100-143) and the record chain of trcbkalignB_ng (:1025-1051), which is what
// alignB_ng (:1347-1353) runs for every DP below MaxVmfSpace (16 M cells, vmf.h:26).
//
// The reference sweeps rows and keeps its layers in arrays indexed by the diagonal r = n - m; a cell only reads slots r - 1,
// r, r + 1, and every comparison is local to the cell.  Here the same arrays are swept in ANTI-DIAGONAL order (the cells of
// one anti-diagonal touch slots of one parity and read the other: independent), one wave per pair, state in LDS -- the
// mapping of g2g_dist.hip plus a direction per slot and ONE TRACE BYTE per cell in HBM.  The reference threads a Vmf record
// index through its records instead; its chain is exactly "the cells of the optimal path where a diagonal run starts"
// (h->dir == NEWD, :264-266), so lane 0 walks the trace bytes back from the end cell and emits those cells.  The host adds
// the two end records and standardises (g2g_stdskl), as globalB_ng does (:1306-1314).
//
// DPs of MaxVmfSpace cells and more go through the reference's linear-space recursion lspB_ng (:1053-1095): centerB_ng
// (:492-782) sweeps the rows below the middle row backward and the rows above it forward with records {val, dir, lwr, upr, lst}
// per diagonal, picks the crossing on the middle row and hands back two narrowed windows; the two parts recurse until they are
// small enough to trace.  Here: a DP is a TASK (ranges of both sequences + a window); g2g_centerb_kernel runs one workgroup per
// (task, phase) over the same diagonal-indexed arrays in HBM, anti-diagonal by anti-diagonal; g2g_centerb_pick_kernel makes the
// crossing; the host walks the recursion level by level (all centers of a level in one launch, all leaves in one launch).
#include <hip/hip_runtime.h>

struct PairTask { int ia, ib, al, ar, bl, br, up, lw; };           // a DP: rows al..ar of sequence ia, columns bl..br of ib, diagonals lw..up
struct PairAlnArgs {
    const uint8_t *pool; const DistSeq *seqs; const PairTask *task; const int *order; int npairs;
    const double *simmtx; int simdim, simrows;
    double bgop, bgep, lgop, lgep, tgapf; int noll, codonk1;
    double *score; int *ends;                                      // per pair: score; {dm, dn, number of records, status}
    int2 *rec; const long long *rec_off;                           // per pair: the traced records (path end first)
    uint8_t *trace; const long long *trace_off;                    // per pair: rows x width bytes
    int *qhead;
    int wmax;                                                      // widest band of the launch
    char *scratch;                                                 // HBM state slices (only when the state is not in LDS)
};
enum { PA_DIAG = 2, PA_NEWD = 3, PA_VERT = 4, PA_HORI = 8, PA_HORL = 11 };       // aln.h:47-52
// trace byte: bits 0-2 where H came from (0 diagonal, 1 G, 2 G2, 3 F, 4 F2), bit 3 the diagonal step started a run (NEWD),
// bits 4-7: G / G2 / F / F2 continued their gap (else: opened from H)
#define PA_T_NEWD 8
#define PA_T_GEXT 16
#define PA_T_G2EXT 32
#define PA_T_FEXT 64
#define PA_T_F2EXT 128
__device__ __forceinline__ bool pa_isdiag(int d) { d &= 15; return d == 2 || d == 3; }
__device__ __forceinline__ bool pa_isvert(int d) { d &= 15; return (d >= 4 && d <= 7) || d == 12; }
__device__ __forceinline__ bool pa_ishori(int d) { d &= 15; return (d >= 8 && d <= 11) || d == 13; }
__device__ __forceinline__ double pa_gappen1(const PairAlnArgs &A) { return 1 > A.codonk1 ? A.lgop + 1 * A.lgep : A.bgop + 1 * A.bgep; }   // GapPenalty(1), aln.h:267
__device__ __forceinline__ double pa_gapext(const PairAlnArgs &A, int i) { return i > A.codonk1 ? A.lgep : A.bgep; }                        // GapExtPen(i), aln.h:272
__device__ __forceinline__ int pa_floor2(int x) { return x >= 0 ? x / 2 : -((-x + 1) / 2); }
__device__ __forceinline__ int pa_ceil2(int x) { return x >= 0 ? (x + 1) / 2 : -((-x) / 2); }

template <class DP, class IP, bool INLDS, bool NOLL3>
__device__ __forceinline__ void pairaln_pair(const PairAlnArgs &A, const int pair, DP dbase, const int wcap, const lf64 *mtx, const int lane)
{
    const PairTask T = A.task[pair];
    const DistSeq sa = A.seqs[T.ia], sb = A.seqs[T.ib];
    const GLB uint8_t *as = glb(A.pool + sa.off), *bs = glb(A.pool + sb.off);
    const int al = T.al, ar = T.ar, bl = T.bl, br = T.br, up = T.up, lw = T.lw;       // (the window: stripe, aln2.cc:156-174, or centerB_ng's)
    const int width = up - lw + 3;
    // five value layers + the direction of the H layer, each `wcap` slots, slot index r - lw + 1
    DP hh = dbase - lw + 1, gg = hh + wcap, g2 = gg + wcap, ff = g2 + wcap, f2 = ff + wcap;
    IP dd = (IP) (dbase + 5 * (size_t) wcap) - lw + 1;
    IP gd = dd + wcap;                                             // (direction of the G layer: the left chain copies VERT records into it)
    GLB uint8_t *trace = glbw(A.trace + A.trace_off[pair]);
    for (int r = lw - 1 + lane; r < lw - 1 + width; r += 64) { hh[r] = NEVSEL; gg[r] = NEVSEL; g2[r] = NEVSEL; ff[r] = NEVSEL; f2[r] = NEVSEL; dd[r] = 0; gd[r] = 0; }
    dist_sync<INLDS>();
    // initB_ng :64-98: two running sums
    const int r0 = bl - al;
    if (lane == 0) {
        hh[r0] = 0; dd[r0] = PA_NEWD;
        const double lt = al ? 1. : A.tgapf;
        int rr = br - al; if (up < rr) rr = up;
        double v = 0;
        for (int i = 1, r = r0 + 1; r <= rr; ++i, ++r) {
            const double gpn = i == 1 ? pa_gappen1(A) : pa_gapext(A, i);
            v = v + gpn * lt;
            hh[r] = v; dd[r] = PA_HORI;
        }
    }
    if (lane == 1) {
        const double lt = bl ? 1. : A.tgapf;
        int rr = bl - ar; if (lw > rr) rr = lw;
        double v = 0;
        for (int i = 1, r = r0 - 1; r >= rr; ++i, --r) {
            const double gpn = i == 1 ? pa_gappen1(A) : pa_gapext(A, i);
            v = v + gpn * lt;
            hh[r] = v; dd[r] = PA_VERT;
            gg[r] = v; gd[r] = PA_VERT;                            // *--g = *h
        }
    }
    dist_sync<INLDS>();
    // forwardB_ng :145-279, anti-diagonal by anti-diagonal
    const int simdim = A.simdim;
    const double bgop = A.bgop, bgep = A.bgep, lgop = A.lgop, lgep = A.lgep;
    for (int d = al + bl; d <= ar + br - 2; ++d) {
        int mlo = pa_ceil2(d - up), mhi = pa_floor2(d - lw);
        if (al > mlo) mlo = al;
        if (d - br + 1 > mlo) mlo = d - br + 1;
        if (ar - 1 < mhi) mhi = ar - 1;
        if (d - bl < mhi) mhi = d - bl;
        for (int m = mlo + lane; m <= mhi; m += 64) {
            const int n = d - m, r = n - m;
            const double hdg = hh[r], hup = hh[r + 1], hlf = hh[r - 1], gup = gg[r + 1], flf = ff[r - 1];
            const bool wasdiag = pa_isdiag(dd[r]);
            int bits = 0;
            double h = hdg + mtx[(int) as[m] * simdim + (int) bs[n]];
            double mxv = h; int win = 0;
            // vertical
            double x = hup + bgop;
            double g;
            if (x >= gup) g = x; else { g = gup; bits |= PA_T_GEXT; }
            g += bgep;
            if (g > mxv) { mxv = g; win = 1; }
            double gq = 0;
            if (NOLL3) {
                const double g2up = g2[r + 1];
                x = hup + lgop;
                if (x >= g2up) gq = x; else { gq = g2up; bits |= PA_T_G2EXT; }
                gq += lgep;
                if (gq > mxv) { mxv = gq; win = 2; }
            }
            // horizontal
            x = hlf + bgop;
            double f;
            if (x >= flf) f = x; else { f = flf; bits |= PA_T_FEXT; }
            f += bgep;
            if (f >= mxv) { mxv = f; win = 3; }
            double fq = 0;
            if (NOLL3) {
                const double f2lf = f2[r - 1];
                x = hlf + lgop;
                if (x >= f2lf) fq = x; else { fq = f2lf; bits |= PA_T_F2EXT; }
                fq += lgep;
                if (fq >= mxv) { mxv = fq; win = 4; }
            }
            int hd = wasdiag ? PA_DIAG : PA_NEWD;
            if (win == 0 && !wasdiag) bits |= PA_T_NEWD;
            if (win == 1 || win == 2) hd = PA_VERT; else if (win == 3) hd = PA_HORI; else if (win == 4) hd = PA_HORL;
            hh[r] = mxv; dd[r] = hd; gg[r] = g; ff[r] = f;
            if (NOLL3) { g2[r] = gq; f2[r] = fq; }
            trace[(size_t) (m - al) * width + (r - lw)] = (uint8_t) (bits | win);
        }
        dist_sync<INLDS>();
    }
    // lastB_ng :100-143 and the walk back, one lane
    if (lane == 0) {
        const int r9 = br - ar;
        const double rt = A.tgapf;
        int dm = 0, dn = 0;
        if (br == sb.len && rt < 1) {
            int rw = up; if (br - al < rw) rw = br - al;
            for (int r = rw - 1; r >= r9; --r) {
                ++dm;
                const double gpn = !pa_isvert(dd[r + 1]) ? pa_gappen1(A) : pa_gapext(A, dm);
                const double gv = hh[r + 1] + gpn * rt;
                hh[r + 1] = gv;
                if (gv > hh[r]) { hh[r] = gv; dd[r] = PA_VERT; } else dm = 0;
            }
        }
        if (ar == sa.len && rt < 1) {
            int rw = lw; if (bl - ar > rw) rw = bl - ar;
            for (int r = rw + 1; r <= r9; ++r) {
                ++dn;
                const double gpn = !pa_ishori(dd[r - 1]) ? pa_gappen1(A) : pa_gapext(A, dn);
                const double fv = hh[r - 1] + gpn * rt;
                hh[r - 1] = fv;
                if (fv > hh[r]) { hh[r] = fv; dd[r] = PA_VERT; } else dn = 0;
            }
        }
        if (dn) dm = 0;
        A.score[pair] = hh[r9];
        // the record chain: the slot the final record came from ends its diagonal at (me, ne); walk the trace bytes back
        GLB int2 *rec = (GLB int2 *) (A.rec + A.rec_off[pair]);
        int nrec = 0;
        const int rend = dn ? r9 - dn : r9 + dm;
        int m = ar - 1, n = m + rend;
        if (n > br - 1) { n = br - 1; m = n - rend; }
        int layer = 0;
        const int cap = (ar - al) + (br - bl) + 2;
        for (int guard = 0; guard < 4 * cap + 8; ++guard) {
            if (m < al || n < bl || n - m < lw || n - m > up) break;               // a boundary record: its chain is the origin
            const int t = trace[(size_t) (m - al) * width + (n - m - lw)];
            if (layer == 0) {
                const int win = t & 7;
                if (win == 0) { if (t & PA_T_NEWD) { if (nrec < cap) { rec[nrec].x = m; rec[nrec].y = n; } ++nrec; } --m; --n; }
                else layer = win;
            } else if (layer == 1) { layer = (t & PA_T_GEXT) ? 1 : 0; --m; }
            else if (layer == 2) { layer = (t & PA_T_G2EXT) ? 2 : 0; --m; }
            else if (layer == 3) { layer = (t & PA_T_FEXT) ? 3 : 0; --n; }
            else { layer = (t & PA_T_F2EXT) ? 4 : 0; --n; }
        }
        GLB int *e = (GLB int *) (A.ends + 4 * (size_t) pair);
        e[0] = dm; e[1] = dn; e[2] = nrec; e[3] = nrec <= cap ? 0 : G2G_ERR_DEVICE;
    }
    dist_sync<INLDS>();
}

template <bool INLDS, bool NOLL3>
__device__ __forceinline__ void pairaln_body(const PairAlnArgs &A, lchar *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    lf64 *mtx = (lf64 *) lds;
    const int nm = A.simdim * A.simrows;
    for (int k = threadIdx.x; k < nm; k += blockDim.x) mtx[k] = A.simmtx[k];
    li32 *pick = (li32 *) (mtx + ((nm + 1) & ~1));
    const int wcap = (A.wmax + 1) & ~1;
    const size_t slot = (size_t) wcap * 48;                        // 5 doubles + 2 ints per band slot
    lchar *state = (lchar *) (pick + 8) + (size_t) wave * slot;
    char *gstate = INLDS ? 0 : A.scratch + ((size_t) blockIdx.x * nwave + wave) * slot;
    __syncthreads();
    for (;;) {
        if (lane == 0) pick[wave] = atomicAdd(A.qhead, 1);
        team_sync();
        const int t = __builtin_amdgcn_readfirstlane(pick[wave]);
        team_sync();
        if (t >= A.npairs) break;
        const int pair = A.order[t];
        if (INLDS) pairaln_pair<lf64 *, li32 *, true, NOLL3>(A, pair, (lf64 *) state, wcap, mtx, lane);
        else pairaln_pair<double *, int *, false, NOLL3>(A, pair, (double *) gstate, wcap, mtx, lane);
    }
}
#define PAIRALN_KERNEL(NAME, INLDS, N3) \
extern "C" __global__ void __launch_bounds__(256) NAME(const PairAlnArgs A) \
{ extern __shared__ __attribute__((aligned(16))) unsigned char pa_lds[]; pairaln_body<INLDS, N3>(A, (lchar *) pa_lds); }
PAIRALN_KERNEL(g2g_pairaln_lds2, true, false)
PAIRALN_KERNEL(g2g_pairaln_lds3, true, true)
PAIRALN_KERNEL(g2g_pairaln_hbm2, false, false)
PAIRALN_KERNEL(g2g_pairaln_hbm3, false, true)

// ---- centerB_ng on the device -------------------------------------------------------------------------------------
// RVDWL (aln.h:96-102) split into a value array and an int4 {dir, lwr, upr, lst} array per layer; layers H, G, G2, F, F2 (the
// reference keeps F / F2 in two registers per row: a slot per diagonal here, so that an anti-diagonal's cells are independent;
// a row's first cell finds the never-written "black" slot, as the reference's reset does).  Per task and phase 5 x width slots.
struct CenterOut { double mxh; int rr0, rr1, kk, status; int fupr, flwr, flst, bupr, blwr, blst, pad0, pad1; };
struct CenterArgs {
    const uint8_t *pool; const DistSeq *seqs; const PairTask *task; int ntask;
    const double *simmtx; int simdim, simrows;
    double bgop, bgep, lgop, lgep, tgapf; int noll, codonk1;
    double *vals; int4 *meta; const long long *soff;               // per task: its first slot; a task owns 10 x width slots
    CenterOut *out;
};
struct CRec { double val; int dir, lwr, upr, lst; };
__device__ __forceinline__ CRec cb_ld(const double *v, const int4 *q, const int i) { CRec r; r.val = v[i]; const int4 t = q[i]; r.dir = t.x; r.lwr = t.y; r.upr = t.z; r.lst = t.w; return r; }
__device__ __forceinline__ void cb_st(double *v, int4 *q, const int i, const CRec &r) { v[i] = r.val; q[i] = make_int4(r.dir, r.lwr, r.upr, r.lst); }
__device__ __forceinline__ double cb_gappen1(const CenterArgs &A) { return 1 > A.codonk1 ? A.lgop + 1 * A.lgep : A.bgop + 1 * A.bgep; }
__device__ __forceinline__ double cb_gapext(const CenterArgs &A, int i) { return i > A.codonk1 ? A.lgep : A.bgep; }

// one phase of centerB_ng: FWD = finitB_ng (:382-435) + the forward sweep (:621-744, without the centre search), else binitB_ng
// (:437-490) + the backward sweep (:520-617)
template <bool FWD, bool NOLL3>
__device__ __forceinline__ void centerb_phase(const CenterArgs &A, const PairTask T, double *v, int4 *q, const lf64 *mtx)
{
    const DistSeq sa = A.seqs[T.ia], sb = A.seqs[T.ib];
    const GLB uint8_t *as = glb(A.pool + sa.off), *bs = glb(A.pool + sb.off);
    const int al = T.al, ar = T.ar, bl = T.bl, br = T.br, up = T.up, lw = T.lw, width = up - lw + 3;
    const int mm = (al + ar + 1) / 2;
    const int tid = threadIdx.x, nthr = blockDim.x;
    double *hv = v - lw + 1, *gv = hv + width, *g2v = gv + width, *fv = g2v + width, *f2v = fv + width;
    int4 *hq = q - lw + 1, *gq = hq + width, *g2q = gq + width, *fq = g2q + width, *f2q = fq + width;
    for (int i = tid; i < 5 * width; i += nthr) { v[i] = NEVSEL; q[i] = make_int4(0, INT_MIN, INT_MAX, 0); }      // black_vdwl, aln.h:186
    __syncthreads();
    if (tid == 0) {
        const double gp1 = cb_gappen1(A);
        if (FWD) {
            const int r0 = bl - al;
            double tg = al ? 1. : A.tgapf;
            CRec o; o.val = 0; o.dir = 0; o.lwr = o.upr = o.lst = r0;
            cb_st(hv, hq, r0, o);
            int rr = br - al; if (up < rr) rr = up;
            double pv = 0;
            for (int i = 1, r = r0 + 1; r <= rr; ++i, ++r) {
                const double gpn = i == 1 ? gp1 : cb_gapext(A, i);
                CRec h; h.dir = PA_HORI; h.lwr = h.lst = r0; h.upr = r;
                h.val = pv + gpn * tg; pv = h.val;
                cb_st(hv, hq, r, h);
            }
            tg = bl ? 1. : A.tgapf;
            rr = bl - mm; if (lw > rr) rr = lw;
            CRec h = o, g; g.val = NEVSEL; g.dir = 0; g.lwr = INT_MIN; g.upr = INT_MAX; g.lst = 0;
            for (int i = 1, r = r0 - 1; r >= rr; ++i, --r) {
                if (i == 1) { h.val += gp1 * tg; h.dir = PA_VERT; g = h; }
                else { h.val += cb_gapext(A, i) * tg; g.val += A.bgep * tg; }
                h.lwr = g.lwr = r;
                cb_st(hv, hq, r, h); cb_st(gv, gq, r, g);
            }
        } else {
            const int r9 = br - ar;
            double tg = ar < sa.len ? 1. : A.tgapf;
            CRec o; o.val = 0; o.dir = 0; o.lwr = o.upr = o.lst = r9;
            cb_st(hv, hq, r9, o);
            int rr = bl - ar; if (lw > rr) rr = lw;
            double pv = 0;
            for (int i = 1, r = r9 - 1; r >= rr; ++i, --r) {
                const double gpn = i == 1 ? gp1 : cb_gapext(A, i);
                CRec h; h.dir = PA_HORI; h.upr = h.lst = r9; h.lwr = r;
                h.val = pv + gpn * tg; pv = h.val;
                cb_st(hv, hq, r, h);
            }
            tg = br < sb.len ? 1. : A.tgapf;
            rr = br - mm; if (up < rr) rr = up;
            CRec h = o, g; g.val = NEVSEL; g.dir = 0; g.lwr = INT_MIN; g.upr = INT_MAX; g.lst = 0;
            for (int i = 1, r = r9 + 1; r <= rr; ++i, ++r) {
                if (i == 1) { h.val += gp1 * tg; h.dir = PA_VERT; g = h; }
                else { h.val += cb_gapext(A, i) * tg; g.val += A.bgep * tg; }
                h.upr = g.upr = r;
                cb_st(hv, hq, r, h); cb_st(gv, gq, r, g);
            }
        }
    }
    __syncthreads();
    const int simdim = A.simdim;
    const double bgop = A.bgop, bgep = A.bgep, lgop = A.lgop, lgep = A.lgep;
    const int row_lo = FWD ? al : mm, row_hi = FWD ? mm - 1 : ar - 1;
    const int d_first = FWD ? al + bl : (ar - 1) + (br - 1), d_last = FWD ? (mm - 1) + (br - 1) : mm + bl;
    const int S = FWD ? 1 : -1;                                    // where the previous row's cells sit: slot r + S (same column), the previous column's: r - S
    for (int d = d_first; FWD ? d <= d_last : d >= d_last; d += S) {
        int mlo = pa_ceil2(d - up), mhi = pa_floor2(d - lw);
        if (row_lo > mlo) mlo = row_lo;
        if (d - br + 1 > mlo) mlo = d - br + 1;
        if (row_hi < mhi) mhi = row_hi;
        if (d - bl < mhi) mhi = d - bl;
        for (int m = mlo + tid; m <= mhi; m += nthr) {
            const int n = d - m, r = n - m;
            CRec H = cb_ld(hv, hq, r);
            const CRec Hu = cb_ld(hv, hq, r + S), Gu = cb_ld(gv, gq, r + S), Hl = cb_ld(hv, hq, r - S);
            CRec F = cb_ld(fv, fq, r - S);
            H.val += mtx[(int) as[m] * simdim + (int) bs[n]];
            H.dir = pa_isdiag(H.dir) ? PA_DIAG : PA_NEWD;
            int mx = 0; double mxv = H.val;
            double x = Hu.val + bgop;                              // vertical
            CRec G;
            if (x >= Gu.val) { G = Hu; G.val = x; G.dir = PA_VERT; } else G = Gu;
            G.val += bgep;
            if (G.val >= mxv) { mx = 1; mxv = G.val; }
            CRec G2, F2;
            if (NOLL3) {
                const CRec G2u = cb_ld(g2v, g2q, r + S);
                x = Hu.val + lgop;
                if (x >= G2u.val) { G2 = Hu; G2.val = x; G2.dir = PA_VERT; } else G2 = G2u;
                G2.val += lgep;
                if (G2.val >= mxv) { mx = 2; mxv = G2.val; }
            }
            x = Hl.val + bgop;                                     // horizontal
            if (x >= F.val) { F = Hl; F.val = x; F.dir = PA_HORI; }
            F.val += bgep;
            if (F.val >= mxv) { mx = 3; mxv = F.val; }
            if (NOLL3) {
                F2 = cb_ld(f2v, f2q, r - S);
                x = Hl.val + lgop;
                if (x >= F2.val) { F2 = Hl; F2.val = x; F2.dir = PA_HORI; }
                F2.val += lgep;
                if (F2.val >= mxv) { mx = 4; mxv = F2.val; }
            }
            // if (mx->dir == NEWD) mx->lst = r;  if (h != mx) {*h = *mx; widen}   (:592-602, :693-701)
            if (mx == 0) { if (H.dir == PA_NEWD) H.lst = r; }
            else {
                if (mx == 1) { if (G.dir == PA_NEWD) G.lst = r; H = G; }
                else if (mx == 3) { if (F.dir == PA_NEWD) F.lst = r; H = F; }
                else if (NOLL3 && mx == 2) { if (G2.dir == PA_NEWD) G2.lst = r; H = G2; }
                else if (NOLL3) { if (F2.dir == PA_NEWD) F2.lst = r; H = F2; }
                if (H.upr < r) H.upr = r;
                if (H.lwr > r) H.lwr = r;
            }
            cb_st(hv, hq, r, H); cb_st(gv, gq, r, G); cb_st(fv, fq, r, F);
            if (NOLL3) { cb_st(g2v, g2q, r, G2); cb_st(f2v, f2q, r, F2); }
        }
        __syncthreads();
    }
}

template <bool NOLL3>
__device__ __forceinline__ void centerb_body(const CenterArgs &A, lchar *lds)
{
    lf64 *mtx = (lf64 *) lds;
    const int nm = A.simdim * A.simrows;
    for (int k = threadIdx.x; k < nm; k += blockDim.x) mtx[k] = A.simmtx[k];
    __syncthreads();
    const int t = blockIdx.x >> 1, phase = blockIdx.x & 1;        // phase 0: forward (arrays 0..4), 1: backward (5..9)
    if (t >= A.ntask) return;
    const PairTask T = A.task[t];
    const int width = T.up - T.lw + 3;
    double *v = A.vals + A.soff[t] + (size_t) phase * 5 * width;
    int4 *q = A.meta + A.soff[t] + (size_t) phase * 5 * width;
    if (phase == 0) centerb_phase<true, NOLL3>(A, T, v, q, mtx);
    else centerb_phase<false, NOLL3>(A, T, v, q, mtx);
}
extern "C" __global__ void __launch_bounds__(256) g2g_centerb_kernel2(const CenterArgs A)
{ extern __shared__ __attribute__((aligned(16))) unsigned char cb_lds[]; centerb_body<false>(A, (lchar *) cb_lds); }
extern "C" __global__ void __launch_bounds__(256) g2g_centerb_kernel3(const CenterArgs A)
{ extern __shared__ __attribute__((aligned(16))) unsigned char cb_lds[]; centerb_body<true>(A, (lchar *) cb_lds); }

// "Find Center" (:716-741) over the forward sweep's last row mm - 1 and the backward sweep's last row mm, then what centerB_ng
// reads off the two records it picked (:746-773).  The candidates of all diagonals in parallel, the reference's fuzzy running
// maximum (gt / ge, cmn.h:61-62) by one thread in the row's order.
__device__ __forceinline__ bool cb_gt(double a, double b) { const double ab = fabs(b); return a > b + 1.e-7 * (1. > ab ? 1. : ab); }
__device__ __forceinline__ bool cb_ge(double a, double b) { const double ab = fabs(b); return a >= b - 1.e-7 * (1. > ab ? 1. : ab); }
extern "C" __global__ void __launch_bounds__(256) g2g_centerb_pick_kernel(const CenterArgs A)
{
    const int t = blockIdx.x;
    const PairTask T = A.task[t];
    const int al = T.al, ar = T.ar, bl = T.bl, br = T.br, up = T.up, lw = T.lw, width = up - lw + 3;
    const int mm = (al + ar + 1) / 2, mm1 = mm - 1;
    double *v = A.vals + A.soff[t]; int4 *q = A.meta + A.soff[t];
    double *hv = v - lw + 1, *gv = hv + width, *g2v = gv + width, *xs = g2v + width;          // (the forward F layer is dead: candidates go there)
    int4 *hq = q - lw + 1, *gq = hq + width, *g2q = gq + width, *ks = g2q + width;
    double *bhv = hv + 5 * width, *bgv = bhv + width, *bg2v = bgv + width;
    int4 *bhq = hq + 5 * width, *bgq = bhq + width, *bg2q = bgq + width;
    const double diffu = A.lgep - A.bgep;
    const int n0 = max(mm1 + lw, bl), n9 = min(mm1 + up + 1, br);
    for (int n = n0 + (int) threadIdx.x; n < n9; n += blockDim.x) {
        const int r = n - mm1;
        const CRec h = cb_ld(hv, hq, r), g = cb_ld(gv, gq, r), hb0 = cb_ld(bhv, bhq, r), hb1 = cb_ld(bgv, bgq, r);
        double x = h.val + hb0.val;
        int k1 = pa_isvert(h.dir) ? 1 : 0, k2 = pa_isvert(hb0.dir) ? 1 : 0;
        double y = g.val + hb1.val - A.bgop;
        const int l = g.lst - hb1.lst - A.codonk1;
        if (l > 0) y += diffu * l;
        if (y >= x) { x = y; k1 = k2 = 1; }
        if (A.noll == 3) {
            const CRec g2 = cb_ld(g2v, g2q, r), hb2 = cb_ld(bg2v, bg2q, r);
            y = g.val + hb2.val - A.bgop + diffu * (g.lst - r);
            if (y > x) { x = y; k1 = 1; k2 = 2; }
            y = g2.val + hb1.val - A.bgop + diffu * (r - hb1.lst);
            if (y > x) { x = y; k1 = 2; k2 = 1; }
            y = g2.val + hb2.val - A.lgop;
            if (y > x) { x = y; k1 = k2 = 2; }
        }
        xs[r] = x; ks[r].x = k1 + 4 * k2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double mxh = NEVSEL; int rr0 = 0, rr1 = 0, kk = 0;
        for (int n = n0; n < n9; ++n) {
            const int r = n - mm1;
            const double x = xs[r];
            if (cb_gt(x, mxh)) { mxh = x; rr0 = rr1 = r; kk = ks[r].x; }
            else if (cb_ge(x, mxh)) rr1 = r;
        }
        CenterOut o;
        o.mxh = mxh; o.rr0 = rr0; o.rr1 = rr1; o.kk = kk; o.status = n0 < n9 ? 0 : G2G_ERR_DEVICE; o.pad0 = o.pad1 = 0;
        o.fupr = o.flwr = o.flst = o.bupr = o.blwr = o.blst = 0;
        if (n0 < n9) {
            const int k1 = kk % 4, k2 = kk / 4;
            const int4 f = (hq + k1 * width)[rr0], b = (bhq + k2 * width)[rr1];
            o.flwr = f.y; o.fupr = f.z; o.flst = f.w; o.blwr = b.y; o.bupr = b.z; o.blst = b.w;
        }
        A.out[t] = o;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
extern "C" g2g_skl *g2g_stdskl(const g2g_skl *in, int num, int *nout);

struct PaOut { double score; int status; std::vector<g2g_skl> recs; };
struct PaDev {                                                     // what every launch of one call shares: sequences + matrix on the device
    char *dev; size_t o_seq, o_mtx; int ncu;
    double bgop, bgep, lgop, lgep, tgapf; int noll, codonk1;
};
static void pa_consts(const g2g_params *prm, PaDev &P)
{   // PwdB::PwdB for two single sequences (src/aln2.cc:80-120): float products of the ALPRM members
    const float f_scale = (float) prm->scale, f_u = (float) prm->u, f_v = (float) prm->v, f_u1 = (float) prm->u1;
    const float vab = f_scale * 1 * 1;
    P.bgop = (double) (-f_v * vab); P.bgep = (double) (-f_u * vab); P.lgep = (double) (-f_u1 * vab);
    P.lgop = P.bgop - (P.lgep - P.bgep) * prm->k1;
    P.noll = prm->ls < 2 ? 2 : prm->ls > 3 ? 3 : prm->ls;
    P.codonk1 = prm->ls == 3 ? (prm->molc == 1 ? 1 : 3) * prm->k1 : INT_MAX / 4 * 3;
    P.tgapf = (double) (float) prm->tgapf;
}

// trcbkalignB_ng of every task: the traced records of a task as the reference writes them (chain end first, then the origin)
static int pairaln_chunk(g2g_ctx *ctx, const g2g_params *prm, const PaDev &P, const std::vector<PairTask> &tasks,
                         const std::vector<int> &run, std::vector<PaOut> &out)
{
    const int nrun = (int) run.size();
    int wmax = 0;
    std::vector<long long> toff((size_t) nrun + 1, 0), roff((size_t) nrun + 1, 0);
    std::vector<std::pair<long long, int> > cost;
    std::vector<PairTask> lt((size_t) nrun);
    for (int k = 0; k < nrun; ++k) {
        const PairTask &T = tasks[run[k]];
        lt[k] = T;
        const int w = T.up - T.lw + 3;
        wmax = std::max(wmax, w);
        toff[k + 1] = toff[k] + (((long long) (T.ar - T.al) * w + 255) & ~255LL);
        roff[k + 1] = roff[k] + (T.ar - T.al) + (T.br - T.bl) + 4;
        cost.push_back(std::make_pair(-(long long) w * ((T.ar - T.al) + (T.br - T.bl)), k));
    }
    std::sort(cost.begin(), cost.end());
    std::vector<int> order;
    for (int k = 0; k < nrun; ++k) order.push_back(cost[k].second);
    auto al256 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    const size_t nm = (size_t) prm->simdim * prm->simrows;
    const size_t o_task = 0, o_ord = al256(o_task + sizeof(PairTask) * (size_t) nrun), o_to = al256(o_ord + 4 * (size_t) nrun),
                 o_ro = al256(o_to + 8 * ((size_t) nrun + 1)), o_in = al256(o_ro + 8 * ((size_t) nrun + 1)),
                 o_scr = o_in, o_end = al256(o_scr + 8 * (size_t) nrun), o_q = al256(o_end + 16 * (size_t) nrun),
                 o_rec = al256(o_q + 256), o_tr = al256(o_rec + 8 * (size_t) roff[nrun]), total = o_tr + (size_t) toff[nrun] + 256;
    std::vector<char> img(o_in, 0);
    memcpy(img.data() + o_task, lt.data(), sizeof(PairTask) * (size_t) nrun);
    memcpy(img.data() + o_ord, order.data(), 4 * (size_t) nrun);
    memcpy(img.data() + o_to, toff.data(), 8 * ((size_t) nrun + 1));
    memcpy(img.data() + o_ro, roff.data(), 8 * ((size_t) nrun + 1));
    char *dev = 0;
    hipError_t e = hipMalloc((void **) &dev, total);
    if (e != hipSuccess) { g2g_set_error("g2g_alignb_ng_batch: hipMalloc: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_NOMEM; }
    e = hipMemcpyAsync(dev, img.data(), o_in, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dev + o_in, 0, o_rec - o_in, ctx->stream);
    PairAlnArgs A;
    A.pool = (const uint8_t *) P.dev; A.seqs = (const DistSeq *) (P.dev + P.o_seq); A.task = (const PairTask *) (dev + o_task);
    A.order = (const int *) (dev + o_ord); A.npairs = nrun;
    A.simmtx = (const double *) (P.dev + P.o_mtx); A.simdim = prm->simdim; A.simrows = prm->simrows;
    A.bgop = P.bgop; A.bgep = P.bgep; A.lgop = P.lgop; A.lgep = P.lgep; A.noll = P.noll; A.codonk1 = P.codonk1; A.tgapf = P.tgapf;
    A.score = (double *) (dev + o_scr); A.ends = (int *) (dev + o_end); A.qhead = (int *) (dev + o_q);
    A.rec = (int2 *) (dev + o_rec); A.rec_off = (const long long *) (dev + o_ro);
    A.trace = (uint8_t *) (dev + o_tr); A.trace_off = (const long long *) (dev + o_to);
    A.wmax = wmax; A.scratch = 0;
    const int ncu = P.ncu;
    const size_t fixed = 8 * ((nm + 1) & ~(size_t) 1) + 32;
    const size_t per_wave = (size_t) ((wmax + 1) & ~1) * 48;
    int nwave = 4;
    while (nwave > 1 && fixed + nwave * per_wave > 64 * 1024) --nwave;
    const bool inlds = fixed + nwave * per_wave <= 160 * 1024 && !g2g_opt(ctx, "DIST_HBM");
    char *scratch = 0;
    int grid; size_t lds;
    typedef void (*pk_t)(const PairAlnArgs);
    const pk_t kern = inlds ? (A.noll == 3 ? g2g_pairaln_lds3 : g2g_pairaln_lds2) : (A.noll == 3 ? g2g_pairaln_hbm3 : g2g_pairaln_hbm2);
    if (inlds) {
        lds = fixed + nwave * per_wave;
        const int wg_per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
        grid = std::min((nrun + nwave - 1) / nwave, ncu * wg_per_cu);
        if (e == hipSuccess && lds > 64 * 1024) e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    } else {
        nwave = 4; lds = fixed;
        grid = std::min((nrun + nwave - 1) / nwave, ncu * 4);
        if (e == hipSuccess) e = hipMalloc((void **) &scratch, per_wave * (size_t) grid * nwave);
        A.scratch = scratch;
    }
    if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] alignB_ng: %d traced DPs, widest band %d, Noll %d, state in %s, %d waves per workgroup, grid %d, lds %zu, trace %.3g bytes\n", nrun, wmax, A.noll, inlds ? "LDS" : "HBM", nwave, grid, lds, (double) toff[nrun]); fflush(stderr); }
    if (e == hipSuccess) { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nwave), lds, ctx->stream, A); e = hipGetLastError(); }
    std::vector<double> hscore((size_t) nrun);
    std::vector<int> hend(4 * (size_t) nrun);
    std::vector<int2> hrec((size_t) roff[nrun] + 1);
    if (e == hipSuccess) e = hipMemcpyAsync(hscore.data(), dev + o_scr, 8 * (size_t) nrun, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hend.data(), dev + o_end, 16 * (size_t) nrun, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && roff[nrun]) e = hipMemcpyAsync(hrec.data(), dev + o_rec, 8 * (size_t) roff[nrun], hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dev);
    if (scratch) hipFree(scratch);
    if (e != hipSuccess) { g2g_set_error("g2g_alignb_ng_batch: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_DEVICE; }
    for (int k = 0; k < nrun; ++k) {
        const PairTask &T = tasks[run[k]];
        PaOut &o = out[run[k]];
        const int dm = hend[4 * k], dn = hend[4 * k + 1], nrec = hend[4 * k + 2];
        o.score = hscore[k];
        if (hend[4 * k + 3] != 0) { o.status = G2G_ERR_DEVICE; continue; }
        // the chain as Vmf::traceback returns it (end first), then the origin (trcbkalignB_ng :1037-1044)
        g2g_skl s;
        s.m = T.ar; s.n = T.br; o.recs.push_back(s);
        if (dm || dn) { s.m = T.ar - dm; s.n = T.br - dn; o.recs.push_back(s); }
        for (int i = 0; i < nrec; ++i) { s.m = hrec[(size_t) roff[k] + i].x; s.n = hrec[(size_t) roff[k] + i].y; o.recs.push_back(s); }
        s.m = T.al; s.n = T.bl; o.recs.push_back(s);
        o.status = G2G_OK;
    }
    return G2G_OK;
}
static int pairaln_tasks(g2g_ctx *ctx, const g2g_params *prm, const PaDev &P, const std::vector<PairTask> &tasks, std::vector<PaOut> &out)
{
    // chunks bounded by the trace bytes (one byte per in-band cell): 8 GB at a time
    size_t budget = (size_t) 8 << 30;
    if (const char *e = g2g_opt(ctx, "ARENA_LIMIT_GB")) { const double g = atof(e); if (g > 0) budget = (size_t) (g * (double) ((size_t) 1 << 30)); }
    out.assign(tasks.size(), PaOut());
    std::vector<int> run;
    size_t acc = 0;
    for (size_t t = 0; t < tasks.size(); ++t) {
        const size_t need = (size_t) (tasks[t].ar - tasks[t].al) * (tasks[t].up - tasks[t].lw + 3) + 4096;
        if (!run.empty() && acc + need > budget) {
            const int rc = pairaln_chunk(ctx, prm, P, tasks, run, out);
            if (rc != G2G_OK) return rc;
            run.clear(); acc = 0;
        }
        run.push_back((int) t); acc += need;
    }
    if (!run.empty()) return pairaln_chunk(ctx, prm, P, tasks, run, out);
    return G2G_OK;
}

// centerB_ng of every task (chunks bounded by the state: 240 bytes per diagonal of the window)
static int centerb_tasks(g2g_ctx *ctx, const g2g_params *prm, const PaDev &P, const std::vector<PairTask> &tasks, std::vector<CenterOut> &out)
{
    out.assign(tasks.size(), CenterOut());
    size_t budget = (size_t) 8 << 30;
    if (const char *e = g2g_opt(ctx, "ARENA_LIMIT_GB")) { const double g = atof(e); if (g > 0) budget = (size_t) (g * (double) ((size_t) 1 << 30)); }
    const size_t nm = (size_t) prm->simdim * prm->simrows;
    int threads = 256;
    if (const char *e = g2g_opt(ctx, "CENTER_THREADS")) { const int t = atoi(e); if (t >= 64 && t <= 256 && t % 64 == 0) threads = t; }
    size_t t0 = 0;
    while (t0 < tasks.size()) {
        size_t t1 = t0, slots = 0;
        std::vector<long long> soff;
        while (t1 < tasks.size()) {
            const size_t w = (size_t) (tasks[t1].up - tasks[t1].lw + 3);
            if (t1 > t0 && (slots + 10 * w) * 24 > budget) break;
            soff.push_back((long long) slots);
            slots += 10 * w; ++t1;
        }
        const int nt = (int) (t1 - t0);
        auto al256 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
        const size_t o_task = 0, o_soff = al256(sizeof(PairTask) * (size_t) nt), o_out = al256(o_soff + 8 * (size_t) nt),
                     o_val = al256(o_out + sizeof(CenterOut) * (size_t) nt), o_meta = al256(o_val + 8 * slots), total = o_meta + 16 * slots + 256;
        std::vector<char> img(o_out, 0);
        memcpy(img.data() + o_task, tasks.data() + t0, sizeof(PairTask) * (size_t) nt);
        memcpy(img.data() + o_soff, soff.data(), 8 * (size_t) nt);
        char *dev = 0;
        hipError_t e = hipMalloc((void **) &dev, total);
        if (e != hipSuccess) { g2g_set_error("g2g_alignb_ng_batch: hipMalloc (centerB_ng): %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_NOMEM; }
        e = hipMemcpyAsync(dev, img.data(), o_out, hipMemcpyHostToDevice, ctx->stream);
        CenterArgs A;
        A.pool = (const uint8_t *) P.dev; A.seqs = (const DistSeq *) (P.dev + P.o_seq); A.task = (const PairTask *) (dev + o_task); A.ntask = nt;
        A.simmtx = (const double *) (P.dev + P.o_mtx); A.simdim = prm->simdim; A.simrows = prm->simrows;
        A.bgop = P.bgop; A.bgep = P.bgep; A.lgop = P.lgop; A.lgep = P.lgep; A.noll = P.noll; A.codonk1 = P.codonk1; A.tgapf = P.tgapf;
        A.vals = (double *) (dev + o_val); A.meta = (int4 *) (dev + o_meta); A.soff = (const long long *) (dev + o_soff);
        A.out = (CenterOut *) (dev + o_out);
        const size_t lds = 8 * ((nm + 1) & ~(size_t) 1) + 32;
        if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] alignB_ng: %d centerB_ng tasks, %zu diagonals of state, %d threads per phase\n", nt, slots / 10, threads); fflush(stderr); }
        if (e == hipSuccess) {
            if (P.noll == 3) hipLaunchKernelGGL(g2g_centerb_kernel3, dim3(2 * nt), dim3(threads), lds, ctx->stream, A);
            else hipLaunchKernelGGL(g2g_centerb_kernel2, dim3(2 * nt), dim3(threads), lds, ctx->stream, A);
            e = hipGetLastError();
        }
        if (e == hipSuccess) { hipLaunchKernelGGL(g2g_centerb_pick_kernel, dim3(nt), dim3(256), 0, ctx->stream, A); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(out.data() + t0, dev + o_out, sizeof(CenterOut) * (size_t) nt, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(dev);
        if (e != hipSuccess) { g2g_set_error("g2g_alignb_ng_batch: centerB_ng: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_DEVICE; }
        t0 = t1;
    }
    return G2G_OK;
}

// lspB_ng's volume test (:1062-1069), int arithmetic as the reference has it (sides below 46341)
static bool pa_traced_whole(const PairTask &T, long long maxvmf)
{
    const int m = T.ar - T.al, n = T.br - T.bl, k = T.lw - T.bl + T.ar, q = T.br - T.al - T.up;
    const long long cvol = (long long) m * n - ((long long) k * k + (long long) q * q) / 2;
    return cvol < maxvmf || m == 1 || n <= 1;
}

extern "C" int g2g_alignb_ng_batch(g2g_ctx *ctx, const g2g_params *prm, int nseq, const g2g_dseq *seqs, int npairs,
                                   const int32_t *ia, const int32_t *ib, double *scr, g2g_skl **skl, int *nskl, int32_t *status)
{
    if (!ctx || !prm || nseq < 0 || npairs < 0 || (nseq && !seqs) || (npairs && (!ia || !ib || !scr || !skl || !nskl || !status))) return G2G_ERR_ARG;
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    if (!prm->simmtx || prm->simdim <= 0 || prm->simrows <= 0) { g2g_set_error("%s", "g2g_alignb_ng_batch: no similarity matrix"); return G2G_ERR_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<DistSeq> hs((size_t) nseq);
    std::vector<char> bad((size_t) nseq, 0);
    size_t pool = 0;
    for (int k = 0; k < nseq; ++k) {
        const g2g_dseq &s = seqs[k];
        hs[k].off = (long long) pool; hs[k].len = 0; hs[k].left = s.left; hs[k].right = s.right;
        if (!s.res || s.len <= 0 || s.left < 0 || s.right <= s.left || s.right > s.len) { bad[k] = 1; continue; }
        for (int i = 0; i < s.len; ++i) if (s.res[i] >= prm->simrows || s.res[i] >= prm->simdim) { bad[k] = 1; break; }
        if (bad[k]) continue;
        hs[k].len = s.len;
        pool += ((size_t) s.len + 15) & ~(size_t) 15;
    }
    long long maxvmf = 16LL * 1024 * 1024;                         // MaxVmfSpace = DefMaxVMF (vmf.h:26); the reference's setVmfSpace <-> this option
    if (const char *e = g2g_opt(ctx, "MAX_VMF_SPACE")) { const long long v = atoll(e); if (v > 0) maxvmf = v; }
    // level 0 of the recursion: every pair with its stripe window (aln2.cc:156-174)
    struct Node { int pair; PairTask T; };
    std::vector<Node> level;
    std::vector<std::vector<g2g_skl> > recs((size_t) npairs);
    for (int p = 0; p < npairs; ++p) {
        status[p] = G2G_ERR_ARG; scr[p] = 0; skl[p] = 0; nskl[p] = 0;
        if (ia[p] < 0 || ia[p] >= nseq || ib[p] < 0 || ib[p] >= nseq || bad[ia[p]] || bad[ib[p]]) continue;
        const g2g_dseq &a = seqs[ia[p]], &b = seqs[ib[p]];
        const long long mm = a.right - a.left, nn = b.right - b.left;
        int sh = prm->sh;
        if (sh < 0) sh = -sh * (int) std::min(mm, nn) / 100;
        int up = b.right - a.right, lw = b.left - a.left;
        if (up < lw) std::swap(up, lw);
        up += sh; lw -= sh;
        up = std::min(up, b.right - a.left); lw = std::max(lw, b.left - a.right);
        if (up - lw + 3 < 3) continue;
        // a band of one diagonal at the top (diagonalB_ng, fwd2b1.cc:1061: its score is a plain sum) is not on this path; sides of
        // 46341 and more overflow the reference's int volume (:1066)
        if (up == lw || mm > 46340 || nn > 46340) { status[p] = G2G_ERR_MODE; continue; }
        Node nd; nd.pair = p;
        nd.T.ia = ia[p]; nd.T.ib = ib[p]; nd.T.al = a.left; nd.T.ar = a.right; nd.T.bl = b.left; nd.T.br = b.right; nd.T.up = up; nd.T.lw = lw;
        level.push_back(nd);
        status[p] = G2G_OK;
    }
    if (level.empty()) return G2G_OK;
    // sequences + matrix: uploaded once for all levels
    PaDev P;
    pa_consts(prm, P);
    {
        auto al256 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
        const size_t nm = (size_t) prm->simdim * prm->simrows;
        P.o_seq = al256(pool); P.o_mtx = al256(P.o_seq + sizeof(DistSeq) * nseq);
        const size_t total = al256(P.o_mtx + 8 * nm);
        std::vector<char> img(total, 0);
        for (int k = 0; k < nseq; ++k) if (hs[k].len > 0 && seqs[k].res) memcpy(img.data() + hs[k].off, seqs[k].res, (size_t) hs[k].len);
        memcpy(img.data() + P.o_seq, hs.data(), sizeof(DistSeq) * nseq);
        memcpy(img.data() + P.o_mtx, prm->simmtx, 8 * nm);
        P.dev = 0;
        hipError_t e = hipMalloc((void **) &P.dev, total);
        if (e == hipSuccess) e = hipMemcpyAsync(P.dev, img.data(), total, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        P.ncu = ctx->ncu > 0 ? ctx->ncu : 256;
        if (e != hipSuccess) { if (P.dev) hipFree(P.dev); g2g_set_error("g2g_alignb_ng_batch: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_NOMEM; }
    }
    int rc = G2G_OK;
    for (int depth = 0; !level.empty() && rc == G2G_OK; ++depth) {
        // lspB_ng :1061-1069: one diagonal -> its two end records (diagonalB_ng :1015-1021); small enough -> traced; else centre
        std::vector<PairTask> leaves, centers;
        std::vector<int> leaf_pair, center_pair;
        std::vector<char> leaf_top, center_top;
        for (const Node &nd : level) {
            if (status[nd.pair] != G2G_OK) continue;
            const PairTask &T = nd.T;
            if (T.up == T.lw) {
                g2g_skl s;
                s.m = T.al; s.n = T.bl; recs[nd.pair].push_back(s);
                s.m = T.ar; s.n = T.ar + (T.bl - T.al); recs[nd.pair].push_back(s);
            } else if (pa_traced_whole(T, maxvmf)) { leaves.push_back(T); leaf_pair.push_back(nd.pair); leaf_top.push_back(depth == 0); }
            else { centers.push_back(T); center_pair.push_back(nd.pair); center_top.push_back(depth == 0); }
        }
        level.clear();
        if (!leaves.empty()) {
            std::vector<PaOut> lo;
            rc = pairaln_tasks(ctx, prm, P, leaves, lo);
            if (rc != G2G_OK) break;
            for (size_t k = 0; k < leaves.size(); ++k) {
                const int p = leaf_pair[k];
                if (lo[k].status != G2G_OK) { status[p] = lo[k].status; continue; }
                recs[p].insert(recs[p].end(), lo[k].recs.begin(), lo[k].recs.end());
                if (leaf_top[k]) scr[p] = lo[k].score;               // (the parts' own values are dropped, :1078, :1090)
            }
        }
        if (!centers.empty()) {
            std::vector<CenterOut> co;
            rc = centerb_tasks(ctx, prm, P, centers, co);
            if (rc != G2G_OK) break;
            for (size_t k = 0; k < centers.size(); ++k) {
                const int p = center_pair[k];
                const PairTask &T = centers[k];
                const CenterOut &o = co[k];
                if (o.status != 0) { status[p] = G2G_ERR_DEVICE; continue; }
                if (center_top[k]) scr[p] = o.mxh;
                // centerB_ng :746-773
                const int mm = (T.al + T.ar + 1) / 2, k1 = o.kk % 4, k2 = o.kk / 4;
                int ml, mr, nl, nr;
                g2g_skl w;
                nl = o.rr0 + mm;
                if (k1) { w.m = ml = nl - o.flst; w.n = nl; recs[p].push_back(w); } else w.m = ml = mm;
                w.n = nr = o.rr1 + mm;
                if (k2) { recs[p].push_back(w); w.m = mr = nr - o.blst; recs[p].push_back(w); } else mr = mm;
                // lspB_ng :1072-1090: the two parts with their windows
                Node L; L.pair = p; L.T = T;
                L.T.up = o.fupr; L.T.lw = o.flwr;
                { const int r = L.T.bl - L.T.al; if (r < L.T.lw) L.T.bl = L.T.al + L.T.lw; if (r > L.T.up) L.T.al = L.T.bl - L.T.up; }
                L.T.ar = ml; L.T.br = nl;
                Node R; R.pair = p; R.T = T;
                R.T.up = o.bupr; R.T.lw = o.blwr;
                { const int r = R.T.br - R.T.ar; if (r < R.T.lw) R.T.ar = R.T.br - R.T.lw; if (r > R.T.up) R.T.br = R.T.ar + R.T.up; }
                R.T.al = mr; R.T.bl = nr;
                // (a part the reference would run into the ground -- an empty range, a window that does not hold its corners -- fails the pair)
                auto sane = [&](const PairTask &Q) {
                    return Q.al >= 0 && Q.bl >= 0 && Q.ar >= Q.al && Q.br >= Q.bl && Q.up >= Q.lw && Q.bl - Q.al >= Q.lw && Q.bl - Q.al <= Q.up &&
                           Q.br - Q.ar >= Q.lw && Q.br - Q.ar <= Q.up && Q.ar <= seqs[Q.ia].len && Q.br <= seqs[Q.ib].len &&
                           (Q.ar - Q.al) + (Q.br - Q.bl) < (T.ar - T.al) + (T.br - T.bl);
                };
                if (!sane(L.T) || !sane(R.T)) { status[p] = G2G_ERR_MODE; g2g_set_error("%s", "g2g_alignb_ng_batch: centerB_ng left a part the recursion cannot take"); continue; }
                level.push_back(L); level.push_back(R);
            }
        }
    }
    hipFree(P.dev);
    if (rc != G2G_OK) return rc;
    for (int p = 0; p < npairs; ++p) {
        if (status[p] != G2G_OK) { scr[p] = 0; continue; }
        int ns = 0;
        skl[p] = g2g_stdskl(recs[p].data(), (int) recs[p].size(), &ns);
        nskl[p] = ns;
        if (!skl[p]) status[p] = G2G_ERR_NOMEM;
    }
    return G2G_OK;
}
