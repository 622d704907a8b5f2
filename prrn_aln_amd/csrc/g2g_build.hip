// g2g_build.hip -- SURVEY.md section 8 rows a8 / a9 ON THE DEVICE: what the reference builds per group before a DP can run --
// column thickness (mSeq::mkthick, reference src/mseq.cc:149-354), frequency and profile vectors (mSeq::convseq / aas2cvec /
// nuc2cvec / profile_p / profile_n / profile, src/mseq.cc:392-587) and the three views of the static gap profile (Gfq::Gfq /
// seq2gfq, src/gfreq.cc:134-312) -- for a whole batch of groups at once.  Included by g2g_engine.hip (shares the context's stream,
// staging buffer and device-memory pool).  The host formulation of the same arrays (g2g_host.cpp: mkthick, convseq, GfqBuilder)
// is pinned on the reference's dumps; this one is pinned on the host's (tests/test_gpu_builders.py: every array bit for bit).
//
// Mapping.  Everything that is a sum over the members of ONE column is independent of every other column: one THREAD per column
// adds its members in member order (the order is the contract: sums are floating point).  That covers the thickness rows, the
// frequency vectors, the profile vectors (a 22 x 22 matrix-vector product per column, terms in the reference's order) and three
// of the gap profile's per-column sums (residues that follow a residue, residues with a possible gap behind them, gap runs that
// open here).  What is sequential along the columns is the bookkeeping of the gap CLASSES (the members whose current gap run
// opened at the same column; g2g_host.cpp describes the model): a class's running weight is added once and then reduced member
// by member, in (column, member) order, as its members leave.  One WAVE per group walks the columns: the lanes look at 64
// members at a time, only the members that LEAVE a gap run at this column (a handful) are visited one by one, lane k holds
// class k, and the column's three lists are packed with ballots.  A first pass counts (pool sizes, hetero), the host lays the
// pools out, a second pass fills them.
//
// Not taken (the caller builds those groups on the host): groups with nil codes -- terminal gaps discounted (tgapf < 1) or free
// ends --, whose zones and gap densities add per-member state to every formula above; more than 64 classes alive at a column.
#include <hip/hip_runtime.h>
#include "g2g_group.h"

namespace g2gb {

enum { B_NIL = 0, B_GAP = 1, B_ASN = 5, B_ASP = 6, B_GLN = 8, B_GLU = 9, B_ASX = 23, B_GLX = 24 };
__device__ const int b_nbits[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};                    // src/mseq.h:42-43
__device__ const int b_amblist[] = {2,3,2,3,5,2,5,3,5,2,3,5,9,2,9,3,9,2,3,9,5,9,2,5,9,3,5,9,2,3,5,9};   // :45-46
__device__ const int b_ambaddr[] = {0, 0, 0, 1, 2, 4, 5, 7, 9, 12, 13, 15, 17, 20, 22, 25, 28};          // :49-50
__device__ const int b_decompact[6] = {0, 1, 2, 3, 5, 9};                                             // :41

struct BGroup {
    int many, len, molc, need, max_code;
    int own_weights;                 // has_weight && many > 1: what mkthick's weight_of() reads (else 1.0)
    int has_weight;                  // what the vectors and the gap profile read (else 1.0)
    int dels;
    int felm, nelm_vec, nelm_out, simdim, simrows;
    double total, lead, trail;       // sumwt; terminal-gap factors of the two ends (tgapf here: no free ends on this path)
    const uint8_t *seq;              // (len + 2) * many, position -1 first
    const double *weight;            // many (unused when !has_weight)
    double *thk;                     // (len + 2) * 3
    double *colsum;                  // len * 3: plain, post, opening weights
    int *coln;                       // len * 3: their member counts
    double *pseq;                    // (len + 2) * nelm_out
    int *off[3]; int *glen[3]; double *freq[3];
    int *counts;                     // [8]: pool sizes of s, t, r; most entries in a t view; longest list incl. terminator; overflow
};

// ---- per-column sums: thickness rows of a group with gaps, and the column-local sums of its gap profile ---------------------
__global__ __launch_bounds__(256) void g2g_build_cols_kernel(const BGroup *G, const int2 *blocks)
{
    const int2 bk = blocks[blockIdx.x];
    const BGroup g = G[bk.x];
    const int col = bk.y + (int) threadIdx.x;
    if (col >= g.len) return;
    const int many = g.many;
    const uint8_t *cur = g.seq + (size_t) (col + 1) * many, *prv = g.seq + (size_t) col * many;
    double gap = 0, rest = 0, plain = 0, post = 0, opening = 0;
    int n_plain = 0, n_post = 0, n_open = 0;
    for (int i = 0; i < many; ++i) {
        const bool is_gap = cur[i] == B_GAP;
        const bool in_run = col > 0 && prv[i] == B_GAP;          // (the walk starts at column 0 with every run length 0)
        const double wt = g.own_weights ? g.weight[i] : 1.0, wg = g.has_weight ? g.weight[i] : 1.0;
        if (is_gap) {
            gap += wt;
            if (!in_run) { opening += wg * 1.0; ++n_open; }      // (x gap density, 1 for a gap that counts)
        } else {
            rest += wt;
            post += wg * 1.0; ++n_post;                           // (x density of the gap that may follow)
            if (!in_run) { plain += wg; ++n_plain; }
        }
    }
    double *t = g.thk + (size_t) (col + 1) * 3;
    t[0] = rest; t[1] = gap; t[2] = g.total;
    if (col == 0) { double *b = g.thk; b[0] = 0; b[1] = g.total * g.lead; b[2] = g.total * g.lead; }
    if (col == g.len - 1) { double *e = g.thk + (size_t) (g.len + 1) * 3; e[0] = 0; e[1] = g.total * g.trail; e[2] = 0; }
    double *cs = g.colsum + (size_t) col * 3;
    cs[0] = plain; cs[1] = post; cs[2] = opening;
    int *cn = g.coln + (size_t) col * 3;
    cn[0] = n_plain; cn[1] = n_post; cn[2] = n_open;
}

// ---- frequency vectors (VECTOR) and profile vectors (VECPRO): one thread per position -1 .. len ------------------------------
#define BV_THREADS 64
__global__ __launch_bounds__(BV_THREADS) void g2g_build_vec_kernel(const BGroup *G, const int2 *blocks, const double *sm)
{
    extern __shared__ double bv_acc[];                           // [nelm_vec][BV_THREADS]
    const int2 bk = blocks[blockIdx.x];
    const BGroup g = G[bk.x];
    const int p = bk.y + (int) threadIdx.x;                      // 0 .. len + 1  <->  position p - 1
    if (p >= g.len + 2) return;
    const int many = g.many, nelm = g.nelm_vec, felm = g.felm, tid = (int) threadIdx.x;
    const uint8_t *cur = g.seq + (size_t) p * many;
#define ACC(k) bv_acc[(k) * BV_THREADS + tid]
    for (int k = 0; k < nelm; ++k) ACC(k) = 0.;
    double e = 0;
    if (g.molc == 1) {                                            // aas2cvec, src/mseq.cc:455-476
        for (int i = 0; i < many; ++i) {
            const int k = cur[i];
            const double wt = g.has_weight ? g.weight[i] : 1.0;
            if (k == B_ASX) { ACC(B_ASN) += wt / 2; ACC(B_ASP) += wt / 2; }
            else if (k == B_GLX) { ACC(B_GLN) += wt / 2; ACC(B_GLU) += wt / 2; }
            else if (k < nelm) ACC(k) += wt;
            if (k == B_GAP) e += wt;
        }
    } else {                                                      // nuc2cvec / ntor, src/mseq.cc:366-384,447-453
        for (int i = 0; i < many; ++i) {
            const int k = cur[i];
            double wt = g.has_weight ? g.weight[i] : 1.0;
            if (k == B_GAP) e += wt;
            switch (k) {
            case 0: ACC(0) += wt; break;
            case 1: ACC(1) += wt; break;
            case 2: ACC(2) += wt; break;
            case 3: ACC(3) += wt; break;
            case 5: ACC(4) += wt; break;
            case 9: ACC(5) += wt; break;
            default:
                if (k >= 1 && k <= 16) {
                    const int m = b_nbits[k - B_GAP];
                    wt /= m;
                    const int *j = b_amblist + b_ambaddr[k];
                    for (int n = 0; n < m; ++n) ACC(j[n]) += wt;
                }
            }
        }
    }
    ACC(nelm - 1) = e;
    double *out = g.pseq + (size_t) p * g.nelm_out;
    if (!(g.need & G2G_NEED_VECPRO)) { for (int k = 0; k < nelm; ++k) out[k] = ACC(k); return; }
    const int dim = g.simdim;
    double *v = out + felm;
    for (int j = 0; j < felm; ++j) out[j] = ACC(j);
    for (int j = 0; j < dim; ++j) v[j] = 0.;                      // (the host allocates the new vectors zeroed)
    if (g.molc != 1) {                                            // profile_n, src/mseq.cc:392-411
        v[0] = 0;
        for (int i = 1; i < felm; ++i) {
            const int k = b_decompact[i];
            double s = 0;
            for (int j = 1; j < felm; ++j) s += sm[(size_t) k * dim + b_decompact[j]] * ACC(j);
            v[k] = s;
        }
        for (int i = 3 + 1; i < g.max_code; ++i) {
            const int m = b_nbits[i - B_GAP];
            if (m == 1) continue;
            double s = 0;
            const int *j = b_amblist + b_ambaddr[i];
            for (int n = 0; n < m; ++n) s += v[j[n]];
            v[i] = s / m;
        }
    } else if (dim == g.simrows) {                                // profile_p, :413-424
        v[0] = 0;
        for (int i = 1; i < felm; ++i) {
            double s = 0;
            for (int j = 1; j < felm; ++j) s += sm[(size_t) i * dim + j] * ACC(j);
            v[i] = s;
        }
        v[B_ASX] = (v[B_ASN] + v[B_ASP]) / 2;
        v[B_GLX] = (v[B_GLN] + v[B_GLU]) / 2;
    } else {                                                      // profile, :426-435
        v[0] = 0;
        for (int i = 1; i < dim; ++i) {
            double s = 0;
            for (int j = 1; j < felm; ++j) s += sm[(size_t) i * dim + j] * ACC(j);
            v[i] = s;
        }
    }
    out[g.nelm_out - 1] = ACC(felm);
#undef ACC
}

// ---- the gap profile: one wave per group walks the columns -------------------------------------------------------------------
__device__ __forceinline__ double bg_readlane(const double x, const int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ void bg_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <bool FILL>
__global__ __launch_bounds__(64) void g2g_build_gfq_kernel(const BGroup *G, const int *gidx)
{
    extern __shared__ char bg_lds[];
    const BGroup g = G[gidx[blockIdx.x]];
    const int many = g.many, len = g.len, lane = (int) threadIdx.x;
    int *run = (int *) bg_lds;                                   // per member: length of the gap run it is in
    double *w = (double *) (bg_lds + (((size_t) many * 4 + 15) & ~(size_t) 15));
    double *st_w = w + many;                                     // staging for the re-numbering of the classes
    int *st_len = (int *) (st_w + 64), *st_n = st_len + 64;
    for (int i = lane; i < many; i += 64) { run[i] = 0; w[i] = g.has_weight ? g.weight[i] : 1.0; }
    bg_sync();
    int cl_len = 0, cl_tn = 0, cl_sn = 0;                        // lane k <-> class k, youngest first
    double cl_tw = 0, cl_sw = 0;
    int ncls = 0, so = 0, to = 0, ro = 0, most = 0, longest = 1, overflow = 0;
    // position -1 (src/gfreq.cc:264-282): everybody is a leading end gap of weight sumwt x terminal-gap factor
    {
        const bool lead = g.lead > 0;
        if (FILL && lane == 0) {
            g.off[0][0] = 0; g.off[1][0] = 0; g.off[2][0] = 0;
            g.glen[0][0] = -1; g.freq[0][0] = 0;
            if (lead) {
                g.glen[1][0] = 0; g.freq[1][0] = g.total * g.lead; g.glen[1][1] = -1; g.freq[1][1] = 0;
                g.glen[2][0] = 0; g.freq[2][0] = g.total * g.lead; g.glen[2][1] = -1; g.freq[2][1] = 0;
            } else { g.glen[1][0] = -1; g.freq[1][0] = 0; g.glen[2][0] = -1; g.freq[2][0] = 0; }
        }
        so = 1; to = ro = lead ? 2 : 1;
        if (lead) longest = 2;
    }
    // Everything a column needs from HBM -- its members' codes (up to 256 members: four registers per lane; larger groups read
    // theirs in place) and its three sums -- is loaded FOUR COLUMNS AHEAD into a rotating set of registers: the walk is a chain of
    // dependent steps of about half a microsecond, an HBM load takes two, and a load waited for inside the step that uses it made
    // the walk take 5.7 us per column.
    const bool pre = many <= 256;
    int pf_code[4][4];
    double pf_w[4][3];
    int pf_n[4][3];
#define BG_FETCH(U, POS) { \
        const int pp_ = (POS); \
        if (pp_ < len) { \
            const uint8_t *q_ = g.seq + (size_t) (pp_ + 1) * many; \
            if (pre) { _Pragma("unroll") for (int u = 0; u < 4; ++u) { const int i_ = u * 64 + lane; pf_code[U][u] = i_ < many ? (int) q_[i_] : 2; } } \
            _Pragma("unroll") for (int u = 0; u < 3; ++u) { pf_w[U][u] = g.colsum[(size_t) pp_ * 3 + u]; pf_n[U][u] = g.coln[(size_t) pp_ * 3 + u]; } \
        } }
    auto column = [&](const int pos, const int c0_, const int c1_, const int c2_, const int c3_, const double plain_w, const double post_w, const double open_w,
                      const int plain_n, const int post_n, const int open_n) __attribute__((always_inline)) {
        const int cu_code[4] = {c0_, c1_, c2_, c3_};
        if (lane < ncls) { cl_len += 1; cl_sw = 0; cl_sn = 0; }
        // members that leave a gap run here: out of their class's running weight, into its s face -- one by one, in member order
        const uint8_t *cur = g.seq + (size_t) (pos + 1) * many;
        for (int c = 0; c < many; c += 64) {
            const int i = c + lane;
            const bool valid = i < many;
            const int u_ = c >> 6;
            const int code = !valid ? 2 : pre ? (u_ == 0 ? cu_code[0] : u_ == 1 ? cu_code[1] : u_ == 2 ? cu_code[2] : cu_code[3]) : (int) cur[i];
            const int r = valid ? run[i] : 0;
            const double wv = valid ? w[i] : 0.;                // (read with the chunk, handed out by readlane: no LDS latency per leaving member)
            const bool is_gap = code == B_GAP;
            unsigned long long leaving = __ballot(valid && !is_gap && r > 0);
            while (leaving) {
                const int j = __builtin_ctzll(leaving);
                leaving &= leaving - 1;
                const int li = __builtin_amdgcn_readlane(r, j);
                const double wi = bg_readlane(wv, j);
                const unsigned long long hit = __ballot(lane < ncls && cl_len == li);
                if (hit) {
                    const int k = __builtin_ctzll(hit);
                    if (lane == k) { cl_tw -= 1.0 * wi; cl_tn -= 1; cl_sw += wi; cl_sn += 1; }
                }
            }
            if (valid) run[i] = is_gap ? r + 1 : 0;
        }
        // s view: [plain] + the classes somebody left at this column, suffix sums from the far end
        {
            const unsigned long long bs = __ballot(lane < ncls && cl_sn > 0);
            const int hp = plain_n ? 1 : 0, ns = hp + __builtin_popcountll(bs);
            if (FILL) {
                double acc = 0, mine = 0;
                unsigned long long rest = bs;
                while (rest) {
                    const int k = 63 - __builtin_clzll(rest);
                    rest &= ~(1ull << k);
                    acc += bg_readlane(cl_sw, k);
                    if (lane == k) mine = acc;
                }
                if (hp) acc += plain_w;
                if (lane == 0) { g.off[0][pos + 1] = so; if (hp) { g.glen[0][so] = 0; g.freq[0][so] = acc; } g.glen[0][so + ns] = -1; g.freq[0][so + ns] = 0; }
                if (lane < ncls && cl_sn > 0) {
                    const int e = so + hp + __builtin_popcountll(bs & ((1ull << lane) - 1));
                    g.glen[0][e] = cl_len; g.freq[0][e] = mine;
                }
            }
            so += ns + 1;
            if (ns + 1 > longest) longest = ns + 1;
        }
        // t view = the classes that live on (the runs that open here in front), r view = [post] + t with every length + 1
        {
            const bool keep = lane < ncls && cl_tn > 0;
            const unsigned long long bt = __ballot(keep);
            const int ho = open_n ? 1 : 0, nt = ho + __builtin_popcountll(bt), hq = post_n ? 1 : 0, nr = hq + nt;
            const int nidx = ho + __builtin_popcountll(bt & ((1ull << lane) - 1));
            if (nt > 64) { overflow = 1; return; }
            if (FILL) {
                if (lane == 0) {
                    g.off[1][pos + 1] = to; g.off[2][pos + 1] = ro;
                    if (ho) { g.glen[1][to] = 0; g.freq[1][to] = open_w; g.glen[2][ro + hq] = 1; g.freq[2][ro + hq] = open_w; }
                    if (hq) { g.glen[2][ro] = 0; g.freq[2][ro] = post_w; }
                    g.glen[1][to + nt] = -1; g.freq[1][to + nt] = 0;
                    g.glen[2][ro + nr] = -1; g.freq[2][ro + nr] = 0;
                }
                if (keep) {
                    g.glen[1][to + nidx] = cl_len; g.freq[1][to + nidx] = cl_tw;
                    g.glen[2][ro + hq + nidx] = cl_len + 1; g.freq[2][ro + hq + nidx] = cl_tw;
                }
            }
            to += nt + 1; ro += nr + 1;
            if (nt > most) most = nt;
            if (nr + 1 > longest) longest = nr + 1;
            // the classes of the next column: [opening] + survivors, re-numbered through LDS
            if (keep) { st_len[nidx] = cl_len; st_w[nidx] = cl_tw; st_n[nidx] = cl_tn; }
            if (ho && lane == 0) { st_len[0] = 0; st_w[0] = open_w; st_n[0] = open_n; }
            bg_sync();
            ncls = nt;
            if (lane < ncls) { cl_len = st_len[lane]; cl_tw = st_w[lane]; cl_tn = st_n[lane]; }
            cl_sw = 0; cl_sn = 0;
            bg_sync();
        }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) { _Pragma("unroll") for (int k = 0; k < 4; ++k) pf_code[u][k] = 2; _Pragma("unroll") for (int k = 0; k < 3; ++k) { pf_w[u][k] = 0; pf_n[u][k] = 0; } }
    BG_FETCH(0, 0) BG_FETCH(1, 1) BG_FETCH(2, 2) BG_FETCH(3, 3)
    for (int pos0 = 0; pos0 < len && !overflow; pos0 += 4) {
#define BG_STEP(U) { \
            const int pos_ = pos0 + U; \
            if (pos_ < len && !overflow) { \
                const int a0_ = pf_code[U][0], a1_ = pf_code[U][1], a2_ = pf_code[U][2], a3_ = pf_code[U][3]; \
                const double w0_ = pf_w[U][0], w1_ = pf_w[U][1], w2_ = pf_w[U][2]; \
                const int n0_ = pf_n[U][0], n1_ = pf_n[U][1], n2_ = pf_n[U][2]; \
                BG_FETCH(U, pos_ + 4) \
                column(pos_, a0_, a1_, a2_, a3_, w0_, w1_, w2_, n0_, n1_, n2_); \
            } }
        BG_STEP(0) BG_STEP(1) BG_STEP(2) BG_STEP(3)
#undef BG_STEP
    }
#undef BG_FETCH
    if (lane == 0) {
        if (FILL) { g.off[0][len + 1] = so; g.off[1][len + 1] = to; g.off[2][len + 1] = ro; }
        else { g.counts[0] = so; g.counts[1] = to; g.counts[2] = ro; g.counts[3] = most; g.counts[4] = longest; g.counts[5] = overflow; }
    }
}

}   // namespace g2gb

// ---- host side ---------------------------------------------------------------------------------------------------------------
int g2g_device_derive(g2g_ctx *ctx, const g2g_params *prm, int n, g2g_group *const *groups, const int *need)
{
    using namespace g2gb;
    if (!ctx || !prm || n < 0 || (n && (!groups || !need))) return G2G_ERR_ARG;
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    if (n == 0) return G2G_OK;
    HIPCHK(hipSetDevice(ctx->device));
    // what this path takes
    for (int k = 0; k < n; ++k) {
        const g2g_group &g = *groups[k];
        if (g.nils || g.exgl || g.exgr || g.many > 4096 || g.len < 1) return G2G_ERR_MODE;
        if (memchr(g.seq.data() + g.many, 0, (size_t) g.len * g.many)) return G2G_ERR_MODE;        // a nil code inside
        if ((need[k] & G2G_NEED_VECPRO) && !prm->simmtx) return G2G_ERR_ARG;
    }
    const bool dbg = g2g_opt(ctx, "DEBUG_PREP") != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[g2g build] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    // 1. host image of the inputs: table, matrix, weights, residues
    std::vector<BGroup> tab((size_t) n);
    Blob bl(ctx);
    size_t est = sizeof(BGroup) * (size_t) n + 8 * (size_t) (prm->simmtx ? prm->simdim * prm->simrows : 0) + 4096;
    for (int k = 0; k < n; ++k) est += groups[k]->seq.size() + 8 * (size_t) groups[k]->many + 64;
    bl.grow(est);
    const size_t tab_off = bl.put(0, 0);
    bl.extend(tab_off + sizeof(BGroup) * (size_t) n);
    const size_t sm_off = prm->simmtx ? bl.put(prm->simmtx, sizeof(double) * (size_t) prm->simdim * prm->simrows) : 0;
    size_t out_bytes = 0;                                          // device-only outputs behind the inputs
    auto out_take = [&](size_t bytes) { const size_t o = out_bytes; out_bytes = (out_bytes + bytes + 255) & ~(size_t) 255; return o; };
    std::vector<size_t> o_thk((size_t) n, 0), o_cs((size_t) n, 0), o_cn((size_t) n, 0), o_ps((size_t) n, 0), o_cnt((size_t) n, 0);
    std::vector<int2> cblocks, vblocks;
    std::vector<int> gfq_groups;
    size_t max_many = 1;
    int max_nelm = 1;
    for (int k = 0; k < n; ++k) {
        g2g_group &g = *groups[k];
        BGroup &b = tab[(size_t) k];
        memset(&b, 0, sizeof b);
        b.many = g.many; b.len = g.len; b.molc = g.molc; b.need = need[k]; b.max_code = g.max_code;
        b.has_weight = g.has_weight ? 1 : 0; b.own_weights = (g.has_weight && g.many > 1) ? 1 : 0; b.dels = g.dels;
        if (g.sumwt == 0) {                                        // Seq::sumwt as mkthick leaves it: the weights in member order
            if (b.own_weights) for (int i = 0; i < g.many; ++i) g.sumwt += g.weight[i];
            else g.sumwt = g.many;
        }
        b.total = g.sumwt; b.lead = g.tgapf; b.trail = g.tgapf;
        b.seq = OFF<const uint8_t>(bl.put(g.seq.data(), g.seq.size()));
        if (g.has_weight) b.weight = OFF<const double>(bl.put(g.weight.data(), sizeof(double) * (size_t) g.many));
        if (g.dels) {
            o_thk[k] = out_take(sizeof(double) * 3 * (size_t) (g.len + 2));
            for (int c = 0; c < g.len; c += 256) cblocks.push_back(make_int2(k, c));
            if (need[k] & G2G_NEED_GFQ) gfq_groups.push_back(k);
            max_many = std::max(max_many, (size_t) g.many);
        }
        if ((need[k] & (G2G_NEED_VECTOR | G2G_NEED_VECPRO)) && g.vect == 0) {
            b.felm = (g.molc == 1) ? (int) B_ASX : 6;              // prepseq, src/mseq.cc:485-502
            b.nelm_vec = b.felm + 1;
            b.simdim = prm->simdim; b.simrows = prm->simrows;
            b.nelm_out = (need[k] & G2G_NEED_VECPRO) ? b.felm + prm->simdim + 1 : b.nelm_vec;
            o_ps[k] = out_take(sizeof(double) * (size_t) b.nelm_out * (size_t) (g.len + 2));
            for (int c = 0; c < g.len + 2; c += BV_THREADS) vblocks.push_back(make_int2(k, c));
            max_nelm = std::max(max_nelm, b.nelm_vec);
        }
    }
    const size_t dl_bytes = out_bytes;                            // what goes back to the host ends here; device-only scratch behind it
    for (int k = 0; k < n; ++k) {
        const g2g_group &g = *groups[k];
        if (!g.dels) continue;
        o_cs[k] = out_take(sizeof(double) * 3 * (size_t) g.len);
        o_cn[k] = out_take(sizeof(int) * 3 * (size_t) g.len);
        o_cnt[k] = out_take(sizeof(int) * 8);
    }
    const size_t cb_off = bl.put(cblocks.data(), sizeof(int2) * cblocks.size());
    const size_t vb_off = bl.put(vblocks.data(), sizeof(int2) * vblocks.size());
    const size_t gi_off = bl.put(gfq_groups.data(), sizeof(int) * gfq_groups.size());
    bl.flush();
    if (bl.oom) { g2g_set_error("%s", "g2g_device_derive: host staging buffer: out of (pinned) memory"); return G2G_ERR_NOMEM; }
    const size_t in_bytes = (bl.size() + 255) & ~(size_t) 255;
    lap("host image of the residues");
    size_t cap1 = 0;
    char *d1 = (char *) pool_take(ctx, in_bytes + out_bytes + 256, &cap1);
    if (!d1) { g2g_set_error("%s", "g2g_device_derive: out of device memory"); return G2G_ERR_NOMEM; }
    char *dout = d1 + in_bytes;
    for (int k = 0; k < n; ++k) {
        BGroup &b = tab[(size_t) k];
        rebase(b.seq, d1); rebase(b.weight, d1);
        if (groups[k]->dels) { b.thk = (double *) (dout + o_thk[k]); b.colsum = (double *) (dout + o_cs[k]); b.coln = (int *) (dout + o_cn[k]); b.counts = (int *) (dout + o_cnt[k]); }
        if (b.nelm_out) b.pseq = (double *) (dout + o_ps[k]);
    }
    memcpy(bl.data() + tab_off, tab.data(), sizeof(BGroup) * (size_t) n);
    int rc = G2G_OK;
    char *d2 = 0; size_t cap2 = 0;
    auto fail = [&](int code, const char *what, hipError_t e) {
        g2g_set_error(what, e == hipSuccess ? "" : hipGetErrorString(e));
        (void) hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        pool_give(ctx, d1, cap1); if (d2) pool_give(ctx, d2, cap2);
        return code;
    };
    hipError_t e = hipMemcpyAsync(d1, bl.data(), bl.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_device_derive: upload: %s", e);
    const BGroup *dG = (const BGroup *) (d1 + tab_off);
    if (!cblocks.empty()) {
        hipLaunchKernelGGL(g2g_build_cols_kernel, dim3((unsigned) cblocks.size()), dim3(256), 0, ctx->stream, dG, (const int2 *) (d1 + cb_off));
        if ((e = hipGetLastError()) != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_build_cols_kernel: %s", e);
        if (dbg) { e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[g2g build] cols kernel (%zu blocks): %s\n", cblocks.size(), hipGetErrorString(e)); lap("upload + cols kernel"); }
    }
    if (!vblocks.empty()) {
        hipLaunchKernelGGL(g2g_build_vec_kernel, dim3((unsigned) vblocks.size()), dim3(BV_THREADS), sizeof(double) * (size_t) max_nelm * BV_THREADS, ctx->stream,
                           dG, (const int2 *) (d1 + vb_off), (const double *) (prm->simmtx ? d1 + sm_off : 0));
        if ((e = hipGetLastError()) != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_build_vec_kernel: %s", e);
        if (dbg) { e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[g2g build] vec kernel (%zu blocks): %s\n", vblocks.size(), hipGetErrorString(e)); lap("vec kernel"); }
    }
    const size_t gfq_lds = (((size_t) max_many * 4 + 15) & ~(size_t) 15) + 8 * max_many + 8 * 64 + 4 * 128 + 64;
    std::vector<int> counts;
    if (!gfq_groups.empty()) {
        hipLaunchKernelGGL(g2g_build_gfq_kernel<false>, dim3((unsigned) gfq_groups.size()), dim3(64), gfq_lds, ctx->stream, dG, (const int *) (d1 + gi_off));
        if ((e = hipGetLastError()) != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_build_gfq_kernel (count): %s", e);
        if (dbg) { e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[g2g build] gfq count kernel (%zu groups, lds %zu): %s\n", gfq_groups.size(), gfq_lds, hipGetErrorString(e)); lap("gfq count kernel"); }
        // the pool sizes come back, the pools are laid out, the second pass fills them
        counts.assign(8 * gfq_groups.size(), 0);
        for (size_t q = 0; q < gfq_groups.size() && e == hipSuccess; ++q)
            e = hipMemcpyAsync(&counts[8 * q], dout + o_cnt[gfq_groups[q]], sizeof(int) * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_device_derive: counts: %s", e);
        lap("upload, column kernels, count pass");
        size_t pool_bytes = 0;
        std::vector<size_t> o_off((size_t) 3 * gfq_groups.size()), o_gl((size_t) 3 * gfq_groups.size()), o_fr((size_t) 3 * gfq_groups.size());
        for (size_t q = 0; q < gfq_groups.size(); ++q) {
            if (counts[8 * q + 5]) return fail(G2G_ERR_MODE, "g2g_device_derive: more than 64 gap classes alive at one column%s", hipSuccess);
            const g2g_group &g = *groups[gfq_groups[q]];
            for (int v = 0; v < 3; ++v) {
                auto take = [&](size_t bytes) { const size_t o = pool_bytes; pool_bytes = (pool_bytes + bytes + 255) & ~(size_t) 255; return o; };
                o_off[3 * q + v] = take(sizeof(int) * (size_t) (g.len + 2));
                o_gl[3 * q + v] = take(sizeof(int) * (size_t) counts[8 * q + v]);
                o_fr[3 * q + v] = take(sizeof(double) * (size_t) counts[8 * q + v]);
            }
        }
        d2 = (char *) pool_take(ctx, pool_bytes + 256, &cap2);
        if (!d2) return fail(G2G_ERR_NOMEM, "g2g_device_derive: out of device memory%s", hipSuccess);
        lap("device memory for the pools");
        for (size_t q = 0; q < gfq_groups.size(); ++q) {
            BGroup &b = tab[(size_t) gfq_groups[q]];
            for (int v = 0; v < 3; ++v) { b.off[v] = (int *) (d2 + o_off[3 * q + v]); b.glen[v] = (int *) (d2 + o_gl[3 * q + v]); b.freq[v] = (double *) (d2 + o_fr[3 * q + v]); }
        }
        memcpy(bl.data() + tab_off, tab.data(), sizeof(BGroup) * (size_t) n);
        e = hipMemcpyAsync(d1 + tab_off, bl.data() + tab_off, sizeof(BGroup) * (size_t) n, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_device_derive: table: %s", e);
        hipLaunchKernelGGL(g2g_build_gfq_kernel<true>, dim3((unsigned) gfq_groups.size()), dim3(64), gfq_lds, ctx->stream, dG, (const int *) (d1 + gi_off));
        if ((e = hipGetLastError()) != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_build_gfq_kernel (fill): %s", e);
    }
    // 2. results into the groups' own arrays
    std::vector<std::pair<void *, std::pair<const char *, size_t>>> copies;         // (host destination, (device source, bytes))
    for (int k = 0; k < n; ++k) {
        g2g_group &g = *groups[k];
        const BGroup &b = tab[(size_t) k];
        if (g.dels) {
            g.has_internalres = true; g.internal_pos.assign((size_t) g.many, 0);
            g.thk_len = g.len; g.thk.resize((size_t) (g.len + 2) * 3);
            copies.push_back({g.thk.data(), {(const char *) b.thk, sizeof(double) * g.thk.size()}});
            g.thk_done = true;
        }
        if (b.nelm_out) {
            g.felm = b.felm; g.nelm = b.nelm_out; g.vect = (need[k] & G2G_NEED_VECPRO) ? 3 : 1;
            g.pseq.resize((size_t) (g.len + 2) * b.nelm_out);
            copies.push_back({g.pseq.data(), {(const char *) b.pseq, sizeof(double) * g.pseq.size()}});
        }
    }
    for (size_t q = 0; q < gfq_groups.size(); ++q) {
        g2g_group &g = *groups[gfq_groups[q]];
        const BGroup &b = tab[(size_t) gfq_groups[q]];
        GapProfile *gp = new GapProfile();
        gp->hetero = counts[8 * q + 3] + 1;
        for (int v = 0; v < 3; ++v) {
            gp->off[v].resize((size_t) g.len + 2); gp->glen[v].resize((size_t) counts[8 * q + v]); gp->freq[v].resize((size_t) counts[8 * q + v]);
            copies.push_back({gp->off[v].data(), {(const char *) b.off[v], sizeof(int) * gp->off[v].size()}});
            copies.push_back({gp->glen[v].data(), {(const char *) b.glen[v], sizeof(int) * gp->glen[v].size()}});
            copies.push_back({gp->freq[v].data(), {(const char *) b.freq[v], sizeof(double) * gp->freq[v].size()}});
        }
        delete g.gfq; g.gfq = gp;
    }
    // device -> pinned staging in a few large copies (the outputs are contiguous per slab), then into the vectors on host threads
    {
        size_t used2 = 0;
        for (size_t q = 0; q < gfq_groups.size(); ++q) { const BGroup &b = tab[(size_t) gfq_groups[q]]; used2 = std::max(used2, (size_t) ((const char *) b.freq[2] - d2) + sizeof(double) * (size_t) counts[8 * q + 2]); }
        const size_t s1 = dl_bytes, s2 = used2;
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_device_derive: kernels: %s", e);      // (the staging buffer is reused below: no upload may still read it)
        lap("fill pass");
        bl.sz = 0;
        bl.grow(s1 + s2 + 512);
        if (bl.oom) return fail(G2G_ERR_NOMEM, "g2g_device_derive: host staging buffer%s", hipSuccess);
        char *h1 = bl.data(), *h2 = bl.data() + ((s1 + 255) & ~(size_t) 255);
        e = s1 ? hipMemcpyAsync(h1, dout, s1, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
        if (e == hipSuccess && used2) e = hipMemcpyAsync(h2, d2, used2, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(G2G_ERR_DEVICE, "g2g_device_derive: download: %s", e);
        lap("download into the staging buffer");
        unsigned nthr = std::thread::hardware_concurrency();
        if (nthr > 16) nthr = 16;
        if (nthr < 1) nthr = 1;
        if (nthr > copies.size()) nthr = (unsigned) std::max<size_t>(1, copies.size());
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (size_t k; (k = next.fetch_add(1)) < copies.size(); ) {
                const char *src = copies[k].second.first;
                const char *hsrc = (d2 && src >= d2 && src < d2 + cap2) ? h2 + (src - d2) : h1 + (src - dout);
                memcpy(copies[k].first, hsrc, copies[k].second.second);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthr; ++t) { try { th.emplace_back(work); } catch (...) { break; } }
        work();
        for (auto &t : th) t.join();
        lap("results into the groups");
    }
    // 3. the device images stay: every group keeps twins of its arrays (g2g_side::dev) and a share in the slabs they live in; the
    // slabs go back to the context's pool with the last group that refers to them (or to the driver, if the context went first)
    {
        struct Slab { g2g_ctx *ctx; std::shared_ptr<int> alive; char *p; size_t cap; };
        auto make = [&](char *p, size_t cap) {
            Slab *sl = new Slab{ctx, ctx->alive, p, cap};
            return std::shared_ptr<void>((void *) sl, [](void *q) {
                Slab *x = (Slab *) q;
                if (*x->alive) pool_give(x->ctx, x->p, x->cap); else hipFree(x->p);
                delete x;
            });
        };
        std::shared_ptr<void> s1 = make(d1, cap1), s2 = d2 ? make(d2, cap2) : std::shared_ptr<void>();
        for (int k = 0; k < n; ++k) {
            g2g_group &g = *groups[k];
            const BGroup &b = tab[(size_t) k];
            g.dev_slabs.push_back(s1);
            g.dev.ctx = ctx;
            g.dev.seq = b.seq;
            g.dev.weight = g.has_weight ? b.weight : 0;
            if (g.dels) g.dev.thk = b.thk;
            if (b.nelm_out) g.dev.pseq = b.pseq;
        }
        for (size_t q = 0; q < gfq_groups.size(); ++q) {
            g2g_group &g = *groups[gfq_groups[q]];
            const BGroup &b = tab[(size_t) gfq_groups[q]];
            g.dev_slabs.push_back(s2);
            for (int v = 0; v < 3; ++v) { g.dev.off[v] = b.off[v]; g.dev.glen[v] = b.glen[v]; g.dev.freq[v] = b.freq[v]; }
        }
    }
    return rc;
}
