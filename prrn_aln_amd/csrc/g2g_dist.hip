// g2g_dist.hip -- f3, the guide-tree stage: score-only pairwise DPs between single sequences, a whole distance matrix per
// launch.  Restates Fwd2d (reference src/fwd2d1.cc:58-158) behind alnScoreD's global branch (:324-338): three arrays
// hh / ff / gg indexed by the diagonal r = n - m, swept in ANTI-DIAGONAL order exactly as the reference does (the cells of
// one anti-diagonal have r of one parity and read only the other parity: they are independent), then the terminal-gap walk
// lastD.  Arithmetic: IEEE doubles, the reference's operation order, no contraction -- scores are bit-equal.
//
// Mapping: ONE WAVE PER PAIR.  A pair's state is 3 x width doubles (width = band + 3, at most len_a + len_b + 3): it lives
// in LDS (the wave's slice of the workgroup's allocation), so a cell is six LDS reads and three LDS writes and nothing
// else touches memory but the two residue bytes and the 8-byte matrix entry (matrix in LDS too).  The lanes take the
// cells of an anti-diagonal 64 at a time; between anti-diagonals the wave only needs its own LDS traffic ordered (no
// barrier).  Waves pull pairs from a queue sorted by cost (longest first), so ragged families balance themselves.  Pairs
// whose state does not fit LDS (sequences of tens of thousands of residues) keep it in an HBM scratch slice per wave.
#include <hip/hip_runtime.h>

struct DistSeq { long long off; int len, left, right; };          // residues at pool + off
struct DistArgs {
    const uint8_t *pool; const DistSeq *seqs; const int *ia, *ib, *order; int npairs;
    const double *simmtx; int simdim, simrows;
    double uu, vv, tgapf; int sh;
    double *score; int *status; int *qhead;
    int wmax;                                                      // doubles of state per wave slot = 3 * (largest width)
    double *scratch;                                               // HBM state slices (only when the state is not in LDS)
};
#define DIST_NEG_INT ((double) (INT_MIN / 8 * 7))                  // cmn.h:97
__device__ __forceinline__ double dist_max(double x, double y) { return x < y ? y : x; }   // std::max

// between anti-diagonals the lanes of the wave read what other lanes wrote: in LDS that is a compiler matter (one wave, LDS
// executes its instructions in order); for the HBM home the stores must have reached the coherent level first
template <bool INLDS>
__device__ __forceinline__ void dist_sync()
{
    if (INLDS) team_sync();
    else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __builtin_amdgcn_wave_barrier(); }
}
// ---- the boundary: a terminal gap of k residues in front of the other sequence --------------------------------------------
// The reference prices it by accumulation: start at -v * lt, add -u * lt once per residue (Fwd2d::Fwd2d, fwd2d1.cc:58-93; lt =
// tgapf where the gap hangs over the sequence's true end, 1 inside a larger sequence), so slot k holds the k-fold ROUNDED sum.
// When no partial sum can round -- both terms are multiples of some 2^q and the largest sum stays below 2^(q+52) -- the k-th
// value is exactly start + k * step and every lane writes its own slots (the ordinary case: u, v small dyadic numbers, tgapf
// 1, 0.5 or 0); otherwise one lane accumulates in order, which is the only way to reproduce the rounding.
__device__ __forceinline__ int dist_low_bit(const double x)              // binary weight of the lowest set bit of x (large when x == 0)
{
    const long long bits = __double_as_longlong(x);
    const int ex = (int) ((bits >> 52) & 0x7FF);
    const long long frac = bits & ((1LL << 52) - 1);
    if (ex == 0) return frac ? -100000 : 100000;                          // subnormal: never "exact"; zero: no constraint
    return ex - 1075 + __ffsll((frac | (1LL << 52))) - 1;
}
__device__ __forceinline__ bool dist_sums_exact(const double start, const double step, const int count)
{
    const int q = min(dist_low_bit(start), dist_low_bit(step));
    if (q <= -100000) return false;
    if (q >= 100000) return true;                                         // both zero
    const double reach = fabs(start) + (double) (count + 1) * fabs(step);
    return reach < ldexp(1.0, q + 52);
}
template <class PTR>
__device__ __forceinline__ void dist_edge(PTR hh, const int r0, const int count, const int dir, const double lt, const double uu,
                                          const double vv, const int lane)
{
    const double start = -vv * lt, step = -uu * lt;
    if (dist_sums_exact(start, step, count)) {
        for (int k = 1 + lane; k <= count; k += 64) hh[r0 + dir * k] = start + (double) k * step;
    } else if (lane == 0) {
        double acc = start;
        for (int k = 1; k <= count; ++k) { acc += step; hh[r0 + dir * k] = acc; }
    }
}

// ---- the end: a terminal gap BEHIND one of the sequences, discounted by tgapf (Fwd2d::lastD, fwd2d1.cc:100-134) --------------
// Along the last column (and, mirrored, the last row) the best score may end in a terminal gap: walking towards the end corner,
// the candidate carried so far pays rt * (v + u) for the first residue of such a gap and rt * u for each further one, and a
// cell replaces it when its own score is at least as good (which also ends the gap: the next residue opens a new one).  The
// walk is a chain of rounded additions with a data-dependent restart, so it cannot be reassociated; what CAN be wave-wide is
// the memory side: 64 cells are loaded with one access, the chain runs through them in registers (readlane of the cell, the
// carried pair (value, in-gap) is wave-uniform), nothing is written back -- the reference's stores into the neighbour slot are
// only ever read by the next step of the same walk.  Returns the carried value after the last cell of [from, to] (to = the
// cell next to the end corner), starting from the value at `origin`.
struct DistCarry { double val; bool in_gap; };
template <class PTR>
__device__ __forceinline__ DistCarry dist_tail_walk(PTR hh, const int origin, const int ncell, const int dir, const double first, const double more, const int lane)
{
    DistCarry c;
    c.val = hh[origin]; c.in_gap = false;
    for (int base = 0; base < ncell; base += 64) {
        const int cnt = ncell - base < 64 ? ncell - base : 64;
        const double x = lane < cnt ? hh[origin + dir * (base + lane + 1)] : 0.;
        const int xlo = __double2loint(x), xhi = __double2hiint(x);
        for (int j = 0; j < cnt; ++j) {
            const double cell = __hiloint2double(__builtin_amdgcn_readlane(xhi, j), __builtin_amdgcn_readlane(xlo, j));
            const double ext = c.val + (c.in_gap ? more : first);
            c.in_gap = cell < ext;                                        // (a tie goes to the cell: the gap ends)
            c.val = c.in_gap ? ext : cell;
        }
    }
    return c;
}

template <class PTR, bool INLDS>
__device__ __forceinline__ void dist_pair(const DistArgs &A, const int pair, PTR base, const lf64 *mtx, const int lane)
{
    const DistSeq sa = A.seqs[A.ia[pair]], sb = A.seqs[A.ib[pair]];
    const GLB uint8_t *as = glb(A.pool + sa.off), *bs = glb(A.pool + sb.off);
    const int al = sa.left, ar = sa.right, bl = sb.left, br = sb.right;
    // stripe, aln2.cc:156-174
    int sh = A.sh;
    if (sh < 0) { const int shorter = ar - al < br - bl ? ar - al : br - bl; sh = -sh * shorter / 100; }
    int up = br - ar, lw = bl - al;
    if (up < lw) { const int t = up; up = lw; lw = t; }
    up += sh; lw -= sh;
    if (br - al < up) up = br - al;
    if (bl - ar > lw) lw = bl - ar;
    const int width = up - lw + 3;
    PTR hh = base - lw + 1, ff = hh + width, gg = ff + width;
    const double uu = A.uu, vv = A.vv;
    // the state before the first anti-diagonal: no gap record anywhere, the start diagonal at 0, leading terminal gaps on either
    // side of it, "never" just outside the band
    const int r0 = bl - al;
    for (int r = lw - 1 + lane; r < lw - 1 + width; r += 64) { ff[r] = NEVSEL; gg[r] = NEVSEL; }
    if (lane == 0) { hh[r0] = 0; hh[up + 1] = DIST_NEG_INT; hh[lw - 1] = DIST_NEG_INT; }
    dist_edge(hh, r0, up - r0, +1, al ? 1. : A.tgapf, uu, vv, lane);
    dist_edge(hh, r0, r0 - lw, -1, bl ? 1. : A.tgapf, uu, vv, lane);
    dist_sync<INLDS>();
    // forwardD :136-158
    const int simdim = A.simdim;
    for (int d = al + bl; d < ar + br - 1; ++d) {
        int n = (d + lw + 1) / 2, m9 = (d - up + 1) / 2;          // (C division, truncating towards zero, as in the reference)
        if (d - ar + 1 > n) n = d - ar + 1;
        if (bl > n) n = bl;
        const int m = d - n;
        if (d - br + 1 > m9) m9 = d - br + 1;
        if (al > m9) m9 = al;
        const int n9 = d - m9 + 1;
        const int rs = n - m, r9 = n9 - m9;
        const int cnt = r9 > rs ? (r9 - rs + 1) >> 1 : 0;
        for (int k = lane; k < cnt; k += 64) {
            const int r = rs + 2 * k, mm = m - k, nn = n + k;
            if (r < lw || r > up) continue;                        // (never: the reference would be outside its arrays)
            const double hl = hh[r - 1], hr = hh[r + 1], fl = ff[r - 1], gr = gg[r + 1], h0 = hh[r];
            const double f = dist_max(hl - vv, fl) - uu;
            const double g = dist_max(hr - vv, gr) - uu;
            double h = h0 + mtx[(int) as[mm] * simdim + (int) bs[nn]];
            h = dist_max(dist_max(h, f), g);
            ff[r] = f; gg[r] = g; hh[r] = h;
        }
        dist_sync<INLDS>();
    }
    // the end corner, with discounted terminal gaps behind either sequence when tgapf < 1
    {
        const int rend = br - ar;
        const double rt = A.tgapf;
        const double first = (vv + uu) * rt, more = uu * rt;
        double best = hh[rend];
        if (br == sb.len && rt < 1) {                              // b is used up: the rest of a hangs over (the last column, towards larger m)
            int from = up + 1;
            if (br - al < from) from = br - al;
            if (from > rend) {
                const DistCarry c = dist_tail_walk(hh, from, from - 1 - rend, -1, first, more, lane);
                const double ext = c.val + (c.in_gap ? more : first);
                if (best < ext) best = ext;
            }
        }
        if (ar == sa.len && rt < 1) {                              // a is used up: the rest of b hangs over (the last row)
            int from = lw;
            if (bl - ar + 1 > from) from = bl - ar + 1;
            if (from < rend) {
                const DistCarry c = dist_tail_walk(hh, from, rend - 1 - from, +1, first, more, lane);
                const double ext = c.val + (c.in_gap ? more : first);
                if (best < ext) best = ext;
            }
        }
        if (lane == 0) { A.score[pair] = best; A.status[pair] = 0; }
    }
    dist_sync<INLDS>();
}

template <bool INLDS>
__device__ __forceinline__ void dist_body(const DistArgs &A, lchar *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    lf64 *mtx = (lf64 *) lds;
    const int nm = A.simdim * A.simrows;
    for (int k = threadIdx.x; k < nm; k += blockDim.x) mtx[k] = A.simmtx[k];
    li32 *pick = (li32 *) (mtx + ((nm + 1) & ~1));                // one queue ticket per wave
    lf64 *state = (lf64 *) (pick + 8) + (size_t) wave * A.wmax;
    double *gstate = INLDS ? 0 : A.scratch + ((size_t) blockIdx.x * nwave + wave) * A.wmax;
    __syncthreads();
    for (;;) {
        if (lane == 0) pick[wave] = atomicAdd(A.qhead, 1);
        team_sync();
        const int t = __builtin_amdgcn_readfirstlane(pick[wave]);
        team_sync();
        if (t >= A.npairs) break;
        const int pair = A.order[t];
        if (INLDS) dist_pair<lf64 *, true>(A, pair, state, mtx, lane);
        else dist_pair<double *, false>(A, pair, gstate, mtx, lane);
    }
}
extern "C" __global__ void __launch_bounds__(256) g2g_dist_lds_kernel(const DistArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dist_lds[];
    dist_body<true>(A, (lchar *) dist_lds);
}
extern "C" __global__ void __launch_bounds__(256) g2g_dist_hbm_kernel(const DistArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dist_lds[];
    dist_body<false>(A, (lchar *) dist_lds);
}

// ---- host side ----------------------------------------------------------------------------------------------------
static int dist_width(const g2g_dseq &a, const g2g_dseq &b, int sh)
{
    if (sh < 0) { const int shorter = std::min(a.right - a.left, b.right - b.left); sh = -sh * shorter / 100; }
    int up = b.right - a.right, lw = b.left - a.left;
    if (up < lw) std::swap(up, lw);
    up += sh; lw -= sh;
    up = std::min(up, b.right - a.left);
    lw = std::max(lw, b.left - a.right);
    return up - lw + 3;
}

extern "C" int g2g_alnscored_batch(g2g_ctx *ctx, const g2g_params *prm, int nseq, const g2g_dseq *seqs, int npairs,
                                   const int32_t *ia, const int32_t *ib, double *score, int32_t *status)
{
    if (!ctx || !prm || nseq < 0 || npairs < 0 || (nseq && !seqs) || (npairs && (!ia || !ib || !score || !status))) return G2G_ERR_ARG;
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    if (!prm->simmtx || prm->simdim <= 0 || prm->simrows <= 0) { g2g_set_error("%s", "g2g_alnscored_batch: no similarity matrix"); return G2G_ERR_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    if (npairs == 0) return G2G_OK;
    // sequences: one pool of residue codes; a sequence the reference could not take (empty range, left != 0: Fwd2d indexes
    // at(left) with the absolute m, fwd2d1.cc:63-64,147; a code outside the matrix) fails its pairs, not the batch
    std::vector<DistSeq> hs((size_t) nseq);
    std::vector<char> bad((size_t) nseq, 0);
    size_t pool = 0;
    for (int k = 0; k < nseq; ++k) {
        const g2g_dseq &s = seqs[k];
        hs[k].off = (long long) pool; hs[k].len = s.len; hs[k].left = s.left; hs[k].right = s.right;
        if (!s.res || s.len <= 0 || s.left != 0 || s.right <= s.left || s.right > s.len) { bad[k] = 1; continue; }
        for (int i = 0; i < s.len; ++i) if (s.res[i] >= prm->simrows || s.res[i] >= prm->simdim) { bad[k] = 1; break; }
        pool += ((size_t) s.len + 15) & ~(size_t) 15;
    }
    std::vector<int> order;
    std::vector<std::pair<long long, int> > cost;
    int wmax = 0;
    for (int p = 0; p < npairs; ++p) {
        status[p] = G2G_ERR_ARG; score[p] = 0;
        if (ia[p] < 0 || ia[p] >= nseq || ib[p] < 0 || ib[p] >= nseq || bad[ia[p]] || bad[ib[p]]) continue;
        const int w = dist_width(seqs[ia[p]], seqs[ib[p]], prm->sh);
        if (w < 3) continue;
        wmax = std::max(wmax, w);
        cost.push_back(std::make_pair(-(long long) w * (seqs[ia[p]].len + seqs[ib[p]].len), p));
    }
    if (cost.empty()) return G2G_OK;
    std::sort(cost.begin(), cost.end());
    for (size_t k = 0; k < cost.size(); ++k) order.push_back(cost[k].second);
    const int nrun = (int) order.size();
    // device image: pool | seqs | ia | ib | order | simmtx | score | status | qhead
    auto al16 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    const size_t o_seq = al16(pool), o_ia = al16(o_seq + sizeof(DistSeq) * nseq), o_ib = al16(o_ia + 4 * (size_t) npairs),
                 o_ord = al16(o_ib + 4 * (size_t) npairs), o_mtx = al16(o_ord + 4 * (size_t) nrun),
                 o_scr = al16(o_mtx + 8 * (size_t) prm->simdim * prm->simrows), o_st = al16(o_scr + 8 * (size_t) npairs),
                 o_q = al16(o_st + 4 * (size_t) npairs), total = o_q + 256;
    std::vector<char> img(o_scr, 0);
    for (int k = 0; k < nseq; ++k) if (!bad[k]) memcpy(img.data() + hs[k].off, seqs[k].res, (size_t) seqs[k].len);
    memcpy(img.data() + o_seq, hs.data(), sizeof(DistSeq) * nseq);
    memcpy(img.data() + o_ia, ia, 4 * (size_t) npairs);
    memcpy(img.data() + o_ib, ib, 4 * (size_t) npairs);
    memcpy(img.data() + o_ord, order.data(), 4 * (size_t) nrun);
    memcpy(img.data() + o_mtx, prm->simmtx, 8 * (size_t) prm->simdim * prm->simrows);
    char *dev = 0;
    HIPCHK(hipMalloc((void **) &dev, total));
    hipError_t e = hipMemcpyAsync(dev, img.data(), o_scr, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dev + o_scr, 0, total - o_scr, ctx->stream);
    DistArgs A;
    A.pool = (const uint8_t *) dev; A.seqs = (const DistSeq *) (dev + o_seq); A.ia = (const int *) (dev + o_ia); A.ib = (const int *) (dev + o_ib);
    A.order = (const int *) (dev + o_ord); A.npairs = nrun;
    A.simmtx = (const double *) (dev + o_mtx); A.simdim = prm->simdim; A.simrows = prm->simrows;
    A.uu = (double) ((float) prm->u * (float) prm->scale); A.vv = (double) ((float) prm->v * (float) prm->scale);   // fwd2d1.cc:65-66 (float members of ALPRM)
    A.tgapf = (double) (float) prm->tgapf; A.sh = prm->sh;
    A.score = (double *) (dev + o_scr); A.status = (int *) (dev + o_st); A.qhead = (int *) (dev + o_q);
    A.wmax = (3 * wmax + 1) & ~1; A.scratch = 0;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, ctx->device);
    const int ncu = e == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t fixed = 8 * (size_t) ((prm->simdim * prm->simrows + 1) & ~1) + 32;
    const size_t per_wave = 8 * (size_t) A.wmax;
    int nwave = 4;
    while (nwave > 1 && fixed + nwave * per_wave > 64 * 1024) --nwave;        // (several workgroups per CU rather than one fat one)
    const bool inlds = fixed + nwave * per_wave <= 160 * 1024 && !g2g_opt(ctx, "DIST_HBM");     // (DIST_HBM: test switch)
    double *scratch = 0;
    int grid;
    size_t lds;
    if (inlds) {
        lds = fixed + nwave * per_wave;
        const int wg_per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
        grid = std::min((nrun + nwave - 1) / nwave, ncu * wg_per_cu);
        if (e == hipSuccess && lds > 64 * 1024) e = hipFuncSetAttribute((const void *) g2g_dist_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    } else {
        nwave = 4; lds = fixed;
        grid = std::min((nrun + nwave - 1) / nwave, ncu * 4);
        if (e == hipSuccess) e = hipMalloc((void **) &scratch, sizeof(double) * (size_t) grid * nwave * A.wmax);
        A.scratch = scratch;
    }
    if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] alnScoreD: %d pairs of %d sequences, widest band %d, state in %s, %d waves per workgroup, grid %d, lds %zu\n", nrun, nseq, wmax, inlds ? "LDS" : "HBM", nwave, grid, lds); fflush(stderr); }
    if (e == hipSuccess) {
        if (inlds) hipLaunchKernelGGL(g2g_dist_lds_kernel, dim3(grid), dim3(64 * nwave), lds, ctx->stream, A);
        else hipLaunchKernelGGL(g2g_dist_hbm_kernel, dim3(grid), dim3(64 * nwave), lds, ctx->stream, A);
        e = hipGetLastError();
    }
    std::vector<double> hscore((size_t) npairs);
    std::vector<int> hstat((size_t) npairs);
    if (e == hipSuccess) e = hipMemcpyAsync(hscore.data(), dev + o_scr, 8 * (size_t) npairs, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hstat.data(), dev + o_st, 4 * (size_t) npairs, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dev);
    if (scratch) hipFree(scratch);
    if (e != hipSuccess) { g2g_set_error("g2g_alnscored_batch: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_DEVICE; }
    for (int k = 0; k < nrun; ++k) { const int p = order[k]; score[p] = hscore[p]; status[p] = G2G_OK; }
    return G2G_OK;
}
