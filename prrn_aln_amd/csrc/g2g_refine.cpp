// g2g_refine.cpp -- SURVEY.md section 8, row f2: prrn's randomised iterative refinement behind the C ABI (g2g_refine), host side
// in C++ on top of level 1 of the same ABI (g2g_group_create / g2g_pwdm_create / g2g_align2_batch / g2g_spscore_batch).
//
// What the reference does (Prrn::rir / onecycle / divideseq / gather / calcfact, reference src/prrn5.cc:414-666; Randiv and
// McRand, src/randiv.cc:34-239; synthgap / delcommongap / aggregate, src/mgaps.cc:181-369; gap2skl, src/gaps.cc:274): draw a
// branch of the weighting tree, split the MSA into the two groups on either side of it, drop the columns that became all-gap
// in each, re-align the two groups, and keep the new alignment if its weighted sum-of-pairs score beats the current one's.
// One accepted move changes the MSA every later division is taken from: a sequential hill climb.
//
// Here the SAME trajectory is produced with the DPs batched.  The branch sequence does not depend on outcomes (a mixed
// congruential generator), so a WINDOW of upcoming divisions is built from the current MSA (groups and PwdMs on host threads),
// evaluated in one g2g_align2_batch + one g2g_spscore_batch, looked at in generator order; the first improving division is
// applied and the ones behind it are drawn again.  With an exchange callback the window is sharded over ranks: every rank
// scores its share, the callback all-gathers fixed-size result slots (RCCL / gloo on the caller's side), and every rank takes
// the same decisions on the same numbers -- no broadcast of the MSA is ever needed.
//
// Own formulation, not the reference's data structures: the MSA is a (columns x members) matrix of residue codes instead of
// per-member gap-run lists; a division's current skeleton is read off that matrix; an accepted skeleton is applied by
// interleaving the two groups' columns.  The tree (topology, Kirchhoff vol / cur per node) is an input.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include "../../include/g2g.h"
#include "g2g_internal.h"

namespace {

const uint8_t GAP = 1;
const double FEPS = 1.0e-7;                                   // reference src/cmn.h:54

// glibc's rand() / srand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31]): McRand seeds itself from them
// (src/randiv.cc:41-51).  Restated so that the library neither depends on nor disturbs the host's generator.
struct GlibcRand {
    std::vector<uint32_t> r;
    explicit GlibcRand(uint32_t seed = 1) { srand_(seed); }
    void srand_(uint32_t seed)
    {
        if (seed == 0) seed = 1;
        int64_t s[34];
        s[0] = (int32_t) seed;
        for (int i = 1; i < 31; ++i) {
            const int64_t hi = s[i - 1] / 127773, lo = s[i - 1] % 127773;       // (C division: truncation toward zero)
            int64_t w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            s[i] = w;
        }
        for (int i = 31; i < 34; ++i) s[i] = s[i - 31];
        r.clear();
        for (int i = 0; i < 34; ++i) r.push_back((uint32_t) s[i]);
        for (int i = 0; i < 310; ++i) next();
    }
    uint32_t next()
    {
        const uint32_t v = r[r.size() - 31] + r[r.size() - 3];
        r.push_back(v);
        if (r.size() > 64) r.erase(r.begin(), r.end() - 34);
        return v;
    }
    int rand_() { return (int) (next() >> 1); }
};

// mixed congruential generator over [0, 2^p), src/randiv.cc:34-56, randiv.h:36-47
struct McRand {
    uint64_t mod, coef, val;
    McRand(int p, int rn)
    {
        GlibcRand libc;
        mod = (uint64_t) 1 << p;
        if (rn == 0) { coef = 1; val = mod - 1; return; }
        const int v = rn == 1 ? libc.rand_() : rn;
        libc.srand_((uint32_t) v);
        coef = (uint64_t) ((libc.rand_() / 4 * 4 + 5) % (int64_t) mod);
        val = (uint64_t) v % mod;
    }
    int next() { val = (coef * val + 1) % mod; return (int) val; }
};

struct Tree {
    int nn, nleaf;
    const int32_t *left, *right, *parent;
    const double *vol, *cur;
    void leaves(int tid, std::vector<int> &out) const
    {
        out.clear();
        std::vector<int> st(1, tid);
        while (!st.empty()) {
            const int k = st.back(); st.pop_back();
            if (left[k] < 0 && right[k] < 0) out.push_back(k);
            else { st.push_back(right[k]); st.push_back(left[k]); }
        }
        std::sort(out.begin(), out.end());
    }
    // Prrn::calcfact / childfact (src/prrn5.cc:414-441): weight of every member when the tree is cut above node tid
    double calcfact(int tid, std::vector<double> &w) const
    {
        w.assign((size_t) nleaf, 0.);
        std::vector<int> lv;
        auto child = [&](int node, double fact) { leaves(node, lv); for (int l : lv) w[l] = vol[l] * fact; };
        int node = tid;
        child(node, 1.0 / vol[node]);
        double fact = 1.0;
        while (parent[node] >= 0) {
            const int father = parent[node];
            const int other = left[father] != node ? left[father] : right[father];
            child(other, fact / vol[father]);
            node = father;
            fact *= cur[node];
        }
        return cur[tid];
    }
};

// The tree comes from the caller through the C ABI: every walk below follows its indices, so they are checked once -- index
// ranges, leaves 0 .. many-1 childless and every inner node with two children, parent / child links consistent, exactly one
// root, every node reached from it exactly once (no cycle, no shared subtree), vol > 0 where it divides.
const char *check_tree(const g2g_tree *t, int many)
{
    const int nn = t->n_nodes;
    int root = -1;
    for (int k = 0; k < nn; ++k) {
        const int l = t->left[k], r = t->right[k], p = t->parent[k];
        if (l < -1 || l >= nn || r < -1 || r >= nn || p < -1 || p >= nn) return "an index is out of range";
        if (k < many) { if (l != -1 || r != -1) return "nodes 0 .. many-1 must be leaves"; }
        else if (l < 0 || r < 0 || l == r || l == k || r == k) return "an inner node needs two different children";
        if (l >= 0 && t->parent[l] != k) return "a left child does not name its parent";
        if (r >= 0 && t->parent[r] != k) return "a right child does not name its parent";
        if (p < 0) { if (root >= 0) return "more than one root"; root = k; }
        else if (t->left[p] != k && t->right[p] != k) return "a node is not a child of its parent";
        if (!(t->vol[k] > 0) || !std::isfinite(t->vol[k]) || !std::isfinite(t->cur[k])) return "vol must be positive and finite, cur finite";
    }
    if (root < 0) return "no root";
    std::vector<char> seen((size_t) nn, 0);
    std::vector<int> st(1, root);
    int cnt = 0;
    while (!st.empty()) {
        const int k = st.back(); st.pop_back();
        if (seen[k]) return "a node is reachable twice";
        seen[k] = 1; ++cnt;
        if (t->left[k] >= 0) { st.push_back(t->left[k]); st.push_back(t->right[k]); }
    }
    if (cnt != nn) return "not every node hangs below the root";
    return 0;
}

// lt(0, delta), src/cmn.h:63
inline bool lt0(double d) { return 0.0 < d - FEPS * std::max(1.0, fabs(d)); }

typedef std::vector<g2g_skl> Skl;

struct Division {
    int branch;
    std::vector<int> la, lb;                                  // (larger group, smaller group); ties keep the complement first
    bool skip;
    double pwt;
    std::vector<uint8_t> a, b;                                // the groups without their all-gap columns, [row][member]
    int ra, rb;                                               // rows of a, b
    Skl old, neu, neu_ab;                                     // current / new skeleton in PwdM order; new in (la, lb) order
    g2g_group *ga, *gb;
    g2g_pwdm *pw;
    int swp;
    double scr, val_old, val_new, delta;
    int status;
    std::string err;                                          // what a builder thread reported (g2g_last_error is per thread)
    Division() : branch(0), skip(false), pwt(0), ra(0), rb(0), ga(0), gb(0), pw(0), swp(0), scr(0), val_old(0), val_new(0), delta(0), status(0) {}
};

struct Msa { int len, many; std::vector<uint8_t> c; uint8_t at(int r, int m) const { return c[(size_t) r * many + m]; } };

// the two groups of a division without their all-gap columns, and the skeleton of their CURRENT alignment (what delcommongap +
// gap2skl give the reference): corners (m, n) where the heading changes
void split_columns(const Msa &M, Division &d)
{
    const int na = (int) d.la.size(), nb = (int) d.lb.size();
    d.a.clear(); d.b.clear(); d.old.clear();
    std::vector<g2g_skl> pts;
    g2g_skl p = {0, 0};
    pts.push_back(p);
    // (every row is gathered into the next free row of its group and kept only if it holds a residue: no per-element growth)
    d.a.resize((size_t) M.len * na); d.b.resize((size_t) M.len * nb);
    pts.reserve((size_t) M.len + 1);
    uint8_t *pa = d.a.data(), *pb = d.b.data();
    const int *la = d.la.data(), *lb = d.lb.data();
    for (int r = 0; r < M.len; ++r) {
        const uint8_t *row = &M.c[(size_t) r * M.many];
        unsigned ka = 0, kb = 0;
        for (int i = 0; i < na; ++i) { const uint8_t c = row[la[i]]; pa[i] = c; ka |= (unsigned) (c != GAP); }
        for (int j = 0; j < nb; ++j) { const uint8_t c = row[lb[j]]; pb[j] = c; kb |= (unsigned) (c != GAP); }
        if (ka) { pa += na; ++p.m; }
        if (kb) { pb += nb; ++p.n; }
        if (ka || kb) pts.push_back(p);
    }
    d.ra = p.m; d.rb = p.n;
    d.a.resize((size_t) p.m * na); d.b.resize((size_t) p.n * nb);
    for (size_t k = 0; k < pts.size(); ++k) {
        bool corner = k == 0 || k + 1 == pts.size();
        if (!corner) corner = (pts[k + 1].m - pts[k].m != pts[k].m - pts[k - 1].m) || (pts[k + 1].n - pts[k].n != pts[k].n - pts[k - 1].n);
        if (corner) d.old.push_back(pts[k]);
    }
}

// apply a skeleton (synthgap, src/mgaps.cc:350): interleave the columns of the two groups
bool join_columns(const Division &d, const Skl &skl, int many, Msa &out)
{
    const int na = (int) d.la.size(), nb = (int) d.lb.size();
    out.many = many; out.c.clear();
    int rows = 0;
    // a skeleton may have crossed a process boundary: first corner (0, 0), last corner (ra, rb), monotone in between
    if (skl.size() < 2 || skl.front().m != 0 || skl.front().n != 0 || skl.back().m != d.ra || skl.back().n != d.rb) return false;
    for (size_t k = 0; k + 1 < skl.size(); ++k) {
        const int m0 = skl[k].m, n0 = skl[k].n, dm = skl[k + 1].m - m0, dn = skl[k + 1].n - n0;
        if (!(dm == dn || dm == 0 || dn == 0) || dm < 0 || dn < 0) return false;
        const int seg = std::max(dm, dn);
        out.c.resize((size_t) (rows + seg) * many, GAP);
        for (int t = 0; t < seg; ++t) {
            uint8_t *row = &out.c[(size_t) (rows + t) * many];
            if (dm) for (int i = 0; i < na; ++i) row[d.la[i]] = d.a[(size_t) (m0 + t) * na + i];
            if (dn) for (int j = 0; j < nb; ++j) row[d.lb[j]] = d.b[(size_t) (n0 + t) * nb + j];
        }
        rows += seg;
    }
    out.len = rows;
    return true;
}

void swap_skl(const Skl &in, Skl &out) { out.resize(in.size()); for (size_t k = 0; k < in.size(); ++k) { out[k].m = in[k].n; out[k].n = in[k].m; } }

void free_division(Division &d)
{
    if (d.pw) g2g_pwdm_free(d.pw);
    if (d.ga) g2g_group_free(d.ga);
    if (d.gb) g2g_group_free(d.gb);
    d.pw = 0; d.ga = d.gb = 0;
}

}   // namespace

extern "C" void g2g_ctx_counters(const g2g_ctx *c, long long out[4]);

extern "C" int g2g_refine(g2g_ctx *ctx, const g2g_params *prm, int many, int len, const uint8_t *codes, const g2g_tree *tree,
                          const g2g_refine_opts *opts, uint8_t **out_codes, int *out_len, g2g_refine_step **steps, int *nsteps,
                          g2g_refine_stats *stats)
{
    if (!prm || !codes || !tree || !out_codes || !out_len || many < 2 || len < 1) return G2G_ERR_ARG;
    if (!ctx && !(opts && opts->scorer)) return G2G_ERR_ARG;               // the DPs run on the GPU unless the caller scores them
    if (tree->n_nodes != 2 * many - 1 || !tree->left || !tree->right || !tree->parent || !tree->vol || !tree->cur) {
        g2g_set_error("%s", "g2g_refine: the tree must have 2 * many - 1 nodes (leaves 0 .. many - 1 = the members)");
        return G2G_ERR_ARG;
    }
    if (const char *why = check_tree(tree, many)) { g2g_set_error("g2g_refine: malformed tree: %s", why); return G2G_ERR_ARG; }
    g2g_refine_opts O;
    memset(&O, 0, sizeof O);
    if (opts) O = *opts;
    if (O.seed == 0) O.seed = 1;
    if (O.maxitr <= 0) O.maxitr = 10;
    if (O.window <= 0) O.window = 32;
    // The first window after an accepted move.  A window costs ONE DP latency whatever its size while the GPU has room (a 5e6-cell
    // DP takes 50-90 ms alone and sixteen of them take 60-100 ms together: a DP is a chain of rows + columns dependent steps), so
    // with large DPs it pays to start wide and throw the speculation behind an acceptance away; small DPs start at 2.
    // (a window costs one DP latency plus about a millisecond per division, whatever its size, and moves the trajectory to its first
    //  accepted division: with one division in six accepted, sixteen at a time is where the 256 x 1024 refinement is fastest -- 96.7 s
    //  against 104.3 s for windows of 8-16 and 97.9 s for 16-32; small MSAs accept more often and start at four: the 48 x 300 family, 38 %
    //  accepted, 8.5 s against 11.1 s from two and 8.7 s from eight)
    if (O.window_min <= 0) O.window_min = (long long) len * len >= (1LL << 22) ? 16 : 4;
    if (O.window_min > O.window) O.window_min = O.window;
    if (O.world <= 0) { O.world = 1; O.rank = 0; }
    if (O.slot_cap <= 0) O.slot_cap = 4096;
    const bool sharded = O.exchange && O.world > 1;
    if (sharded && (O.rank < 0 || O.rank >= O.world)) { g2g_set_error("%s", "g2g_refine: rank outside [0, world)"); return G2G_ERR_ARG; }
    const int slot_ints = 9 + 2 * O.slot_cap;

    Tree T;
    T.nn = tree->n_nodes; T.nleaf = many; T.left = tree->left; T.right = tree->right; T.parent = tree->parent; T.vol = tree->vol; T.cur = tree->cur;
    // Randiv in TREEDIV mode (src/randiv.cc:158-178,217-226)
    const int cycle = 2 * many - 3;
    int p2 = 0;
    for (int x = 1; x < cycle; x <<= 1) ++p2;
    McRand mcr(p2, O.seed);
    auto next_branch = [&]() { for (;;) { const int r = mcr.next(); if (r < cycle) return r; } };

    Msa M;
    M.len = len; M.many = many; M.c.assign(codes, codes + (size_t) len * many);
    std::vector<g2g_refine_step> log;
    g2g_refine_stats S;
    memset(&S, 0, sizeof S);
    int rc_all = G2G_OK;
    long long cnt0[4] = {0, 0, 0, 0};
    if (ctx) g2g_ctx_counters(ctx, cnt0);

    const int maxi = O.maxitr * cycle;
    int nrep = 0, it = 0, win = O.window_min;
    std::vector<int> pending;
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr > 16) nthr = 16;
    if (nthr < 1) nthr = 1;

    double t_build = 0, t_align = 0, t_rest = 0;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    while (it < maxi && rc_all == G2G_OK) {
        const double t0 = now();
        while ((int) pending.size() < std::min(win, maxi - it)) pending.push_back(next_branch());
        const int nw = std::min(win, (int) pending.size());
        std::vector<Division> D((size_t) nw);
        // A failure of THIS rank (a builder, a kernel, a skeleton too long for a slot) must not keep it out of the exchange: the
        // other ranks would wait in the collective for ever.  It is carried through the exchange as this rank's status word and
        // every rank leaves the loop with the same code after the gather.
        int local_rc = G2G_OK;
        std::string local_err;
        // ---- build the window's divisions from the current MSA (host threads: the builders touch only their own objects) ----
        std::atomic<int> nextd(0);
        auto build = [&]() {
            std::vector<int> inside;
            std::vector<double> w;
            for (int k; (k = nextd.fetch_add(1)) < nw; ) {
                Division &d = D[k];
                d.branch = pending[k];
                T.leaves(d.branch, inside);
                std::vector<char> in((size_t) many, 0);
                for (int l : inside) in[l] = 1;
                std::vector<int> outside;
                for (int i = 0; i < many; ++i) if (!in[i]) outside.push_back(i);
                // bin2lst2 + the swap of Prrn::divideseq: (larger group, smaller group)
                if (outside.size() < inside.size()) { d.la = inside; d.lb = outside; } else { d.la = outside; d.lb = inside; }
                d.pwt = T.calcfact(d.branch, w);
                split_columns(M, d);
                if (d.ra == M.len && d.rb == M.len) { d.skip = true; continue; }     // nothing to re-align (src/prrn5.cc:497-498,518-521)
                // a group of one member is the member itself in the reference (aliaseq): weight 1
                std::vector<double> wa, wb;
                for (int l : d.la) wa.push_back(d.la.size() > 1 ? w[l] : 1.0);
                for (int l : d.lb) wb.push_back(d.lb.size() > 1 ? w[l] : 1.0);
                d.ga = g2g_group_create(ctx, prm, (int) d.la.size(), d.ra, d.a.data(), wa.data());
                d.gb = g2g_group_create(ctx, prm, (int) d.lb.size(), d.rb, d.b.data(), wb.data());
                if (d.ga && d.gb) d.pw = g2g_pwdm_create(ctx, prm, d.ga, d.gb, &d.swp);
                if (!d.pw) { d.status = G2G_ERR_ARG; d.err = g2g_last_error(); continue; }      // (this thread's error string: kept for the caller's thread)
                if (d.swp) { Skl t; swap_skl(d.old, t); d.old.swap(t); }
            }
        };
        {
            std::vector<std::thread> th;
            const unsigned use = std::min<unsigned>(nthr, (unsigned) nw);
            for (unsigned t = 1; t < use; ++t) th.emplace_back(build);
            build();
            for (auto &t : th) t.join();
        }
        t_build += now() - t0;
        for (int k = 0; k < nw && local_rc == G2G_OK; ++k)
            if (D[k].status) { local_rc = D[k].status; local_err = "g2g_refine: building a division's groups failed: " + D[k].err; }
        // ---- score: my share of the window (largest rectangles first, round-robin), then the exchange ----
        std::vector<int> live;
        for (int k = 0; k < nw; ++k) if (!D[k].skip) live.push_back(k);
        std::vector<int> order(live.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int) i;
        if (sharded) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (long long) D[live[x]].ra * D[live[x]].rb > (long long) D[live[y]].ra * D[live[y]].rb; });
        std::vector<int> mine;
        for (size_t i = (size_t) (sharded ? O.rank : 0); i < order.size(); i += (size_t) (sharded ? O.world : 1)) mine.push_back(order[i]);
        const int nm = (int) mine.size();
        if (nm && local_rc == G2G_OK) {
            std::vector<g2g_pwdm *> pw((size_t) nm);
            for (int i = 0; i < nm; ++i) pw[i] = D[live[mine[i]]].pw;
            std::vector<double> scr((size_t) nm);
            std::vector<g2g_skl *> skl((size_t) nm, (g2g_skl *) 0);
            std::vector<int> nskl((size_t) nm, 0), st((size_t) nm, 0);
            int rc;
            if (O.scorer) {
                // the caller scores (tests drive the loop on a CPU with the checker in this seat): same inputs, same outputs
                std::vector<const g2g_skl *> cur((size_t) nm);
                std::vector<int> ncur((size_t) nm);
                std::vector<double> raw((size_t) nm), val((size_t) nm);
                for (int i = 0; i < nm; ++i) { const Division &d = D[live[mine[i]]]; cur[i] = d.old.data(); ncur[i] = (int) d.old.size(); }
                const double t1 = now();
                rc = O.scorer(O.scorer_user, nm, pw.data(), cur.data(), ncur.data(), scr.data(), skl.data(), nskl.data(), raw.data(), val.data());
                t_align += now() - t1;
                if (rc != G2G_OK) g2g_set_error("%s", "g2g_refine: the scorer callback failed");
                for (int i = 0; i < nm && rc == G2G_OK; ++i) {
                    Division &d = D[live[mine[i]]];
                    if (!skl[i] || nskl[i] < 2) { rc = G2G_ERR_DEVICE; g2g_set_error("%s", "g2g_refine: the scorer callback returned no skeleton"); break; }
                    d.scr = scr[i]; d.val_old = raw[i]; d.val_new = val[i];
                    d.neu.assign(skl[i], skl[i] + nskl[i]);
                }
            } else {
                // DPs, then calcSpScore of the current and of the new alignment of every division, on one resident batch
                std::vector<const g2g_skl *> cur((size_t) nm);
                std::vector<int> ncur((size_t) nm);
                for (int i = 0; i < nm; ++i) { const Division &d = D[live[mine[i]]]; cur[i] = d.old.data(); ncur[i] = (int) d.old.size(); }
                std::vector<g2g_fstat> fc((size_t) nm), fn((size_t) nm);
                const double t1 = now();
                rc = g2g_align2_score_batch(ctx, nm, pw.data(), cur.data(), ncur.data(), G2G_SP_NOSTATS, scr.data(), skl.data(), nskl.data(), st.data(), fc.data(), fn.data());
                t_align += now() - t1;
                if (const char *e = getenv("G2G_REFINE_SLOW_MS")) {        // diagnostics: windows whose scoring took unusually long (the scheduler's stalls)
                    const double dt = now() - t1;
                    if (dt > atof(e)) {
                        long long c[4] = {0, 0, 0, 0};
                        g2g_ctx_counters(ctx, c);
                        fprintf(stderr, "[g2g_refine] slow window %d: %d DPs scored in %.1f ms (wait time-outs so far %lld, recovered DPs %lld)\n", S.batches, nm, dt, c[1], c[2]);
                    }
                }
                for (int i = 0; i < nm && rc == G2G_OK; ++i) {
                    Division &d = D[live[mine[i]]];
                    if (st[i] != 0 || fc[i].status != 0 || fn[i].status != 0) {
                        rc = st[i] ? st[i] : fc[i].status ? fc[i].status : fn[i].status;
                        g2g_set_error("%s", "g2g_refine: a division's DP or its sum-of-pairs score failed");
                        break;
                    }
                    d.scr = scr[i]; d.val_old = fc[i].raw; d.val_new = fn[i].val;
                    d.neu.assign(skl[i], skl[i] + nskl[i]);
                }
            }
            for (int i = 0; i < nm; ++i) g2g_free(skl[i]);
            if (rc != G2G_OK) { local_rc = rc; local_err = g2g_last_error(); }
            else S.divisions_scored_here += nm;
        }
        ++S.batches;
        if (sharded) {
            // per rank: {status, 0} + fixed-size slots {index in the window's live list, 0, corners, DP score, raw current score, new
            // fstat.val, corners...}
            const int ntot = (int) live.size();
            const int nslots = (ntot + O.world - 1) / O.world;
            const int rank_ints = 2 + nslots * slot_ints;
            std::vector<int32_t> mybuf((size_t) rank_ints, -1), all((size_t) rank_ints * O.world, -1);
            for (int i = 0; i < nm && local_rc == G2G_OK; ++i) {
                const Division &d = D[live[mine[i]]];
                int32_t *s = &mybuf[2 + (size_t) i * slot_ints];
                if ((int) d.neu.size() > O.slot_cap) { local_rc = G2G_ERR_ARG; local_err = "g2g_refine: a skeleton exceeds the exchange slot capacity"; break; }
                s[0] = mine[i]; s[1] = 0; s[2] = (int32_t) d.neu.size();
                memcpy(s + 3, &d.scr, 8); memcpy(s + 5, &d.val_old, 8); memcpy(s + 7, &d.val_new, 8);
                for (size_t k = 0; k < d.neu.size(); ++k) { s[9 + 2 * k] = d.neu[k].m; s[10 + 2 * k] = d.neu[k].n; }
            }
            mybuf[0] = local_rc; mybuf[1] = 0;
            if (local_rc != G2G_OK) for (size_t q = 2; q < mybuf.size(); ++q) mybuf[q] = -1;       // a failed rank hands over no slot
            const int xrc = O.exchange(O.exchange_user, mybuf.data(), rank_ints, all.data());
            if (xrc != 0) {       // (the callback is the caller's collective: when it fails it must fail on every rank)
                g2g_set_error("%s", "g2g_refine: the exchange callback failed"); rc_all = G2G_ERR_DEVICE;
            } else {
                for (int r = 0; r < O.world && rc_all == G2G_OK; ++r) {
                    const int32_t st_r = all[(size_t) r * rank_ints];
                    if (st_r == G2G_OK) continue;
                    rc_all = st_r < 0 ? st_r : G2G_ERR_DEVICE;
                    if (r == O.rank && !local_err.empty()) g2g_set_error("%s", local_err.c_str());
                    else { char who[96]; snprintf(who, sizeof who, "g2g_refine: rank %d failed (its code is this call's return value)", r); g2g_set_error("%s", who); }
                }
            }
            int seen = 0;
            std::vector<char> got((size_t) (ntot > 0 ? ntot : 1), 0);
            for (int r = 0; r < O.world && rc_all == G2G_OK; ++r)
                for (int q = 0; q < nslots && rc_all == G2G_OK; ++q) {
                    const int32_t *s = &all[(size_t) r * rank_ints + 2 + (size_t) q * slot_ints];
                    if (s[0] < 0) continue;                                                     // an unused slot
                    // slot contents crossed a process boundary: nothing in them is trusted
                    if (s[0] >= ntot || got[s[0]] || s[2] < 2 || s[2] > O.slot_cap) {
                        g2g_set_error("%s", "g2g_refine: the exchange returned a malformed slot"); rc_all = G2G_ERR_DEVICE; break;
                    }
                    got[s[0]] = 1;
                    Division &d = D[live[s[0]]];
                    memcpy(&d.scr, s + 3, 8); memcpy(&d.val_old, s + 5, 8); memcpy(&d.val_new, s + 7, 8);
                    d.neu.resize((size_t) s[2]);
                    for (int k = 0; k < s[2]; ++k) { d.neu[k].m = s[9 + 2 * k]; d.neu[k].n = s[10 + 2 * k]; }
                    ++seen;
                }
            if (rc_all == G2G_OK && seen != ntot) { g2g_set_error("%s", "g2g_refine: the exchange did not return every division of the window"); rc_all = G2G_ERR_DEVICE; }
        } else if (local_rc != G2G_OK) {
            g2g_set_error("%s", local_err.c_str());
            rc_all = local_rc;
        }
        if (rc_all != G2G_OK) { for (auto &d : D) free_division(d); break; }
        for (int k : live) {
            Division &d = D[k];
            bool same = d.neu.size() == d.old.size();
            for (size_t q = 0; same && q < d.neu.size(); ++q) same = d.neu[q].m == d.old[q].m && d.neu[q].n == d.old[q].n;
            // Prrn::onecycle (src/prrn5.cc:523,535): the NEW alignment enters with Gsinfo.fstat.val (rescaled by PwdM::Vab), the
            // CURRENT one with the return value of calcSpScore(SKL*), which is not rescaled -- kept as the reference has it
            d.delta = same ? 0.0 : d.pwt * (d.val_new - d.val_old);
            if (d.swp) swap_skl(d.neu, d.neu_ab); else d.neu_ab = d.neu;
        }
        // ---- look at the window in generator order ----
        int consumed = 0;
        bool accepted = false;
        const double t_window = now() - t_begin;
        for (int k = 0; k < nw; ++k) {
            Division &d = D[k];
            ++consumed; ++it;
            g2g_refine_step e;
            memset(&e, 0, sizeof e);
            e.t_ms = t_window;
            e.branch = d.branch; e.na = (int) d.la.size(); e.nb = (int) d.lb.size();
            if (d.skip) {
                e.skipped = 1; e.delta = -INFINITY;
                log.push_back(e);
                ++nrep;
                if (nrep >= cycle || it >= maxi) break;
                continue;
            }
            const bool ok = lt0(d.delta);
            e.swp = d.swp; e.scr = d.scr; e.val_new = d.val_new; e.val_old = d.val_old; e.delta = d.delta; e.accepted = ok;
            log.push_back(e);
            if (ok) {
                Msa N;
                if (!join_columns(d, d.neu_ab, many, N)) { g2g_set_error("%s", "g2g_refine: a skeleton does not describe an alignment of its two groups"); rc_all = G2G_ERR_DEVICE; break; }
                M.len = N.len; M.c.swap(N.c);
                if (O.on_accept) O.on_accept(O.on_accept_user, d.branch, (int) d.la.size(), d.la.data(), (int) d.lb.size(), d.lb.data(), (int) d.neu_ab.size(), d.neu_ab.data());
                nrep = 1; accepted = true; ++S.accepted;
                break;
            }
            ++nrep;
            if (nrep >= cycle || it >= maxi) break;
        }
        S.divisions_wasted += nw - consumed;
        for (auto &d : D) free_division(d);
        pending.erase(pending.begin(), pending.begin() + consumed);
        if (rc_all != G2G_OK || nrep >= cycle) break;
        win = accepted ? O.window_min : std::min(O.window, 2 * win);
    }
    if (rc_all != G2G_OK) return rc_all;
    t_rest = now() - t_begin - t_build - t_align;
    if (getenv("G2G_REFINE_TIMES")) fprintf(stderr, "[g2g_refine] %d batches: build %.0f ms, scoring (align2 + calcSpScore, or the scorer callback) %.0f ms, rest %.0f ms\n", S.batches, t_build, t_align, t_rest);
    S.divisions = (int) log.size();
    if (ctx) {        // waits of the DP scheduler that ran into their limit during this call, and the DPs re-run for it (0 in an ordinary run)
        long long cnt1[4] = {0, 0, 0, 0};
        g2g_ctx_counters(ctx, cnt1);
        S.wait_timeouts = (int) (cnt1[1] - cnt0[1]); S.recovered_dps = (int) (cnt1[2] - cnt0[2]);
    }
    *out_len = M.len;
    *out_codes = (uint8_t *) malloc(M.c.size() ? M.c.size() : 1);
    if (!*out_codes) return G2G_ERR_NOMEM;
    memcpy(*out_codes, M.c.data(), M.c.size());
    if (steps && nsteps) {
        *nsteps = (int) log.size();
        *steps = (g2g_refine_step *) malloc(sizeof(g2g_refine_step) * (log.size() ? log.size() : 1));
        if (!*steps) { free(*out_codes); *out_codes = 0; return G2G_ERR_NOMEM; }
        memcpy(*steps, log.data(), sizeof(g2g_refine_step) * log.size());
    }
    if (stats) *stats = S;
    return G2G_OK;
}
