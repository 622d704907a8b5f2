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
#include <hip/hip_runtime.h>

struct PairAlnArgs {
    const uint8_t *pool; const DistSeq *seqs; const int *ia, *ib, *order; int npairs;
    const double *simmtx; int simdim, simrows;
    double bgop, bgep, lgop, lgep, tgapf; int noll, codonk1, sh;
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
    const DistSeq sa = A.seqs[A.ia[pair]], sb = A.seqs[A.ib[pair]];
    const GLB uint8_t *as = glb(A.pool + sa.off), *bs = glb(A.pool + sb.off);
    const int al = sa.left, ar = sa.right, bl = sb.left, br = sb.right;
    int sh = A.sh;                                                 // stripe, aln2.cc:156-174
    if (sh < 0) { const int shorter = ar - al < br - bl ? ar - al : br - bl; sh = -sh * shorter / 100; }
    int up = br - ar, lw = bl - al;
    if (up < lw) { const int t = up; up = lw; lw = t; }
    up += sh; lw -= sh;
    if (br - al < up) up = br - al;
    if (bl - ar > lw) lw = bl - ar;
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

// ---- host side ----------------------------------------------------------------------------------------------------
extern "C" g2g_skl *g2g_stdskl(const g2g_skl *in, int num, int *nout);

static int pairaln_chunk(g2g_ctx *ctx, const g2g_params *prm, int nseq, const g2g_dseq *seqs, const std::vector<DistSeq> &hs,
                         const std::vector<int> &run, const int32_t *ia, const int32_t *ib, double *scr, g2g_skl **skl, int *nskl, int32_t *status)
{
    const int nrun = (int) run.size();
    int wmax = 0;
    std::vector<long long> toff((size_t) nrun + 1, 0), roff((size_t) nrun + 1, 0);
    std::vector<std::pair<long long, int> > cost;
    for (int k = 0; k < nrun; ++k) {
        const g2g_dseq &a = seqs[ia[run[k]]], &b = seqs[ib[run[k]]];
        const int w = dist_width(a, b, prm->sh);
        wmax = std::max(wmax, w);
        toff[k + 1] = toff[k] + (((long long) (a.right - a.left) * w + 255) & ~255LL);
        roff[k + 1] = roff[k] + (a.right - a.left) + (b.right - b.left) + 4;
        cost.push_back(std::make_pair(-(long long) w * (a.len + b.len), k));
    }
    std::sort(cost.begin(), cost.end());
    std::vector<int> order;
    for (int k = 0; k < nrun; ++k) order.push_back(cost[k].second);
    std::vector<int> lia((size_t) nrun), lib((size_t) nrun);
    for (int k = 0; k < nrun; ++k) { lia[k] = ia[run[k]]; lib[k] = ib[run[k]]; }
    size_t pool = 0;
    for (int k = 0; k < nseq; ++k) pool = std::max(pool, (size_t) hs[k].off + (((size_t) (hs[k].len > 0 ? hs[k].len : 0) + 15) & ~(size_t) 15));
    auto al256 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    const size_t nm = (size_t) prm->simdim * prm->simrows;
    const size_t o_seq = al256(pool), o_ia = al256(o_seq + sizeof(DistSeq) * nseq), o_ib = al256(o_ia + 4 * (size_t) nrun),
                 o_ord = al256(o_ib + 4 * (size_t) nrun), o_mtx = al256(o_ord + 4 * (size_t) nrun), o_to = al256(o_mtx + 8 * nm),
                 o_ro = al256(o_to + 8 * ((size_t) nrun + 1)), o_in = al256(o_ro + 8 * ((size_t) nrun + 1)),
                 o_scr = o_in, o_end = al256(o_scr + 8 * (size_t) nrun), o_q = al256(o_end + 16 * (size_t) nrun),
                 o_rec = al256(o_q + 256), o_tr = al256(o_rec + 8 * (size_t) roff[nrun]), total = o_tr + (size_t) toff[nrun] + 256;
    std::vector<char> img(o_in, 0);
    for (int k = 0; k < nseq; ++k) if (hs[k].len > 0 && seqs[k].res) memcpy(img.data() + hs[k].off, seqs[k].res, (size_t) hs[k].len);
    memcpy(img.data() + o_seq, hs.data(), sizeof(DistSeq) * nseq);
    memcpy(img.data() + o_ia, lia.data(), 4 * (size_t) nrun);
    memcpy(img.data() + o_ib, lib.data(), 4 * (size_t) nrun);
    memcpy(img.data() + o_ord, order.data(), 4 * (size_t) nrun);
    memcpy(img.data() + o_mtx, prm->simmtx, 8 * nm);
    memcpy(img.data() + o_to, toff.data(), 8 * ((size_t) nrun + 1));
    memcpy(img.data() + o_ro, roff.data(), 8 * ((size_t) nrun + 1));
    char *dev = 0;
    hipError_t e = hipMalloc((void **) &dev, total);
    if (e != hipSuccess) { g2g_set_error("g2g_alignb_ng_batch: hipMalloc: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_NOMEM; }
    e = hipMemcpyAsync(dev, img.data(), o_in, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dev + o_in, 0, o_rec - o_in, ctx->stream);
    PairAlnArgs A;
    A.pool = (const uint8_t *) dev; A.seqs = (const DistSeq *) (dev + o_seq); A.ia = (const int *) (dev + o_ia); A.ib = (const int *) (dev + o_ib);
    A.order = (const int *) (dev + o_ord); A.npairs = nrun;
    A.simmtx = (const double *) (dev + o_mtx); A.simdim = prm->simdim; A.simrows = prm->simrows;
    {   // PwdB::PwdB for two single sequences (src/aln2.cc:80-120): float products of the ALPRM members
        const float f_scale = (float) prm->scale, f_u = (float) prm->u, f_v = (float) prm->v, f_u1 = (float) prm->u1;
        const float vab = f_scale * 1 * 1;
        A.bgop = (double) (-f_v * vab); A.bgep = (double) (-f_u * vab); A.lgep = (double) (-f_u1 * vab);
        A.lgop = A.bgop - (A.lgep - A.bgep) * prm->k1;
        A.noll = prm->ls < 2 ? 2 : prm->ls > 3 ? 3 : prm->ls;
        A.codonk1 = prm->ls == 3 ? (prm->molc == 1 ? 1 : 3) * prm->k1 : INT_MAX / 4 * 3;
    }
    A.tgapf = (double) (float) prm->tgapf; A.sh = prm->sh;
    A.score = (double *) (dev + o_scr); A.ends = (int *) (dev + o_end); A.qhead = (int *) (dev + o_q);
    A.rec = (int2 *) (dev + o_rec); A.rec_off = (const long long *) (dev + o_ro);
    A.trace = (uint8_t *) (dev + o_tr); A.trace_off = (const long long *) (dev + o_to);
    A.wmax = wmax; A.scratch = 0;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, ctx->device);
    const int ncu = e == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
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
    if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] alignB_ng: %d pairs, widest band %d, Noll %d, state in %s, %d waves per workgroup, grid %d, lds %zu, trace %.3g bytes\n", nrun, wmax, A.noll, inlds ? "LDS" : "HBM", nwave, grid, lds, (double) toff[nrun]); fflush(stderr); }
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
        const int p = run[k];
        const g2g_dseq &a = seqs[ia[p]], &b = seqs[ib[p]];
        const int dm = hend[4 * k], dn = hend[4 * k + 1], nrec = hend[4 * k + 2];
        if (hend[4 * k + 3] != 0) { status[p] = G2G_ERR_DEVICE; continue; }
        // the chain as Vmf::traceback returns it (end first), then the origin (trcbkalignB_ng :1037-1044)
        std::vector<g2g_skl> raw;
        g2g_skl s;
        s.m = a.right; s.n = b.right; raw.push_back(s);
        if (dm || dn) { s.m = a.right - dm; s.n = b.right - dn; raw.push_back(s); }
        for (int i = 0; i < nrec; ++i) { s.m = hrec[(size_t) roff[k] + i].x; s.n = hrec[(size_t) roff[k] + i].y; raw.push_back(s); }
        s.m = a.left; s.n = b.left; raw.push_back(s);
        int ns = 0;
        skl[p] = g2g_stdskl(raw.data(), (int) raw.size(), &ns);
        nskl[p] = ns;
        scr[p] = hscore[k];
        status[p] = skl[p] ? G2G_OK : G2G_ERR_NOMEM;
    }
    return G2G_OK;
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
    // chunks bounded by the trace bytes (one byte per in-band cell): 8 GB at a time
    size_t budget = (size_t) 8 << 30;
    if (const char *e = g2g_opt(ctx, "ARENA_LIMIT_GB")) { const double g = atof(e); if (g > 0) budget = (size_t) (g * (double) ((size_t) 1 << 30)); }
    std::vector<int> run;
    size_t acc = 0;
    for (int p = 0; p < npairs; ++p) {
        status[p] = G2G_ERR_ARG; scr[p] = 0; skl[p] = 0; nskl[p] = 0;
        if (ia[p] < 0 || ia[p] >= nseq || ib[p] < 0 || ib[p] >= nseq || bad[ia[p]] || bad[ib[p]]) continue;
        const g2g_dseq &a = seqs[ia[p]], &b = seqs[ib[p]];
        const int w = dist_width(a, b, prm->sh);
        if (w < 3) continue;
        {   // what alignB_ng does NOT trace in one piece is not on this path: a band of one diagonal (diagonalB_ng, fwd2b1.cc:1061)
            // and DPs of MaxVmfSpace cells or more (the linear-space recursion, :1062-1069)
            const long long mm = a.right - a.left, nn = b.right - b.left;
            int sh = prm->sh;
            if (sh < 0) sh = -sh * (int) std::min(mm, nn) / 100;
            int up = b.right - a.right, lw = b.left - a.left;
            if (up < lw) std::swap(up, lw);
            up += sh; lw -= sh;
            up = std::min(up, b.right - a.left); lw = std::max(lw, b.left - a.right);
            const long long kk = lw - b.left + a.right, qq = b.right - a.left - up;
            const long long cvol = mm * nn - (kk * kk + qq * qq) / 2;
            if (up == lw || !(cvol < 16LL * 1024 * 1024 || mm == 1 || nn <= 1)) { status[p] = G2G_ERR_MODE; continue; }
        }
        const size_t need = (size_t) (a.right - a.left) * w + 4096;
        if (!run.empty() && acc + need > budget) {
            const int rc = pairaln_chunk(ctx, prm, nseq, seqs, hs, run, ia, ib, scr, skl, nskl, status);
            if (rc != G2G_OK) return rc;
            run.clear(); acc = 0;
        }
        run.push_back(p); acc += need;
    }
    if (!run.empty()) return pairaln_chunk(ctx, prm, nseq, seqs, hs, run, ia, ib, scr, skl, nskl, status);
    return G2G_OK;
}
