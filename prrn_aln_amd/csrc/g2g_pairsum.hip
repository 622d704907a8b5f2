// g2g_pairsum.hip -- f1: the naive branch of Ssrel::pairsum_ss (reference src/fspscore.cc:896-922 -> Sptree::sptree :795-806): the
// sum-of-pairs score of a sub-alignment of at most ndesc_thr = 60 members, Msap::ps_nml (src/maln2.cc:1801-1824):
//     per column:  crg1w / crg1i (:510-530, :1532-1552: gap-open count over member pairs, from the members' running gap lengths)
//                  + selfw / selfi (:495-508, :1214-1228: substitution scores over member pairs), pair-weighted or not;
//     with -yl3:   Gep1st::longup(s, n, gla) (src/mseq.cc:698-727), the long-gap count.
// The column values are independent once every member's running gap length and k1-th previous residue position are known per
// column, so: phase 1, one thread per MEMBER walks its row and writes those two numbers per column; phase 2, one thread per
// COLUMN sums its pairs in the reference's pair order; phase 3, one thread adds the columns in order (the score is a sum of
// doubles: the order is part of the result).  One workgroup per node.
#include <hip/hip_runtime.h>
#include <functional>

struct PairsumArgs {
    const uint8_t *codes;            // [len][n] residue codes of the node's members in leaf order
    const double *pw;                // pair weights, pair (i < j) at j (j - 1) / 2 + i; NULL: unweighted
    const double *simmtx; int simdim;
    int n, len, k1, ls3;
    double basic_gop, diffu;
    int *gla;                        // [len + 1][n]: running gap length BEFORE column c (row len: after the last column)
    int *old;                        // [len][n]: position of the k1-th previous residue column of the member before column c (0 if none)
    double *colv, *coll;             // [len]: column value, long-gap count
    double *out;
};

extern "C" __global__ void __launch_bounds__(256) g2g_pairsum_kernel(const PairsumArgs *args)
{
    const PairsumArgs A = args[blockIdx.x];
    const int n = A.n, len = A.len;
    // phase 1: per member (pregap with left = 0: all zero)
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        int g = 0, ring[32], rp = 0;
        for (int k = 0; k < 32; ++k) ring[k] = 0;
        for (int c = 0; c < len; ++c) {
            A.gla[(size_t) c * n + m] = g;
            if (A.ls3) A.old[(size_t) c * n + m] = ring[rp];
            const bool res = A.codes[(size_t) c * n + m] > 1;
            g = res ? 0 : g + 1;                                   // incrgap, src/mgaps.cc:431-440
            if (A.ls3 && res) { ring[rp] = c; if (++rp == A.k1) rp = 0; }      // Gep1st::shift -> Queue::shift (clib.h:322-326)
        }
        A.gla[(size_t) len * n + m] = g;
    }
    __syncthreads();
    // phase 2: per column
    for (int c = threadIdx.x; c < len; c += blockDim.x) {
        const uint8_t *as = A.codes + (size_t) c * n;
        const int *gl = A.gla + (size_t) c * n, *ga = A.gla + (size_t) (c + 1) * n;
        const double *q = A.pw;
        double g = 0, s = 0;
        for (int i = 1; i < n; ++i) {                              // crg1
            if (as[i] > 1) {
                for (int j = 0; j < i; ++j) {
                    const double au = as[j] <= 1 ? 1. : 0.;        // mSeq::gapdensity with tgapf = 1 (mseq.h:148-153)
                    if (au > 0 && gl[i] >= gl[j]) g += A.pw ? *q * au : au;
                    if (A.pw) ++q;
                }
            } else {                                               // (a gap: density 1)
                for (int j = 0; j < i; ++j) {
                    if (as[j] > 1 && gl[i] <= gl[j]) g += A.pw ? *q * 1. : 1.;
                    if (A.pw) ++q;
                }
            }
        }
        q = A.pw;
        for (int i = 1; i < n; ++i) {                              // sim1
            const double *row = A.simmtx + (size_t) as[i] * A.simdim;
            for (int j = 0; j < i; ++j) { s += A.pw ? row[as[j]] * *q : row[as[j]]; if (A.pw) ++q; }
        }
        A.colv[c] = g * A.basic_gop + s;
        double tlu = 0;
        if (A.ls3) {
            const int *od = A.old + (size_t) c * n;
            q = A.pw;
            for (int i = 1; i < n; ++i) {                          // Gep1st::longup(s, n, gla): the test reads s[i - 1] (mseq.cc:703-705)
                if (!(as[i - 1] > 1)) { if (A.pw) q += i; continue; }
                for (int j = 0; j < i; ++j) {
                    if (ga[i] > c - od[j]) tlu += A.pw ? *q : 1.;
                    if (A.pw) ++q;
                }
            }
        }
        A.coll[c] = tlu;
    }
    __syncthreads();
    // phase 3: the columns in order
    if (threadIdx.x == 0) {
        double scr = 0, lunp = 0;
        for (int c = 0; c < len; ++c) { scr += A.colv[c]; lunp += A.coll[c]; }
        if (A.ls3) scr += lunp * A.diffu;
        *A.out = scr;
    }
}

// ---- host side: Sptree::sptree (fspscore.cc:784-821) -------------------------------------------------------------------------
namespace {
struct PsTree {
    const g2g_tree *t;
    bool leaf(int k) const { return t->left[k] < 0 && t->right[k] < 0; }
    void leaves(int k, std::vector<int> &out) const { if (leaf(k)) { out.push_back(k); return; } leaves(t->left[k], out); leaves(t->right[k], out); }   // addleaf: left first
    int ndesc(int k) const { return leaf(k) ? 1 : ndesc(t->left[k]) + ndesc(t->right[k]); }
    // Ktree::recalcpw -> repairwt (src/phyl.cc:762-811): w_i w_j / vol(LCA)^2, leaves numbered in visiting order
    int repairwt(int k, int base, const std::vector<int> &lv, std::vector<double> &pw) const
    {
        if (leaf(k)) return 1;
        const int nl = repairwt(t->left[k], base, lv, pw), nr = repairwt(t->right[k], base + nl, lv, pw);
        const double wab = 1. / (t->vol[k] * t->vol[k]);
        for (int i = 0; i < nl; ++i)
            for (int j = 0; j < nr; ++j) {
                const int a = base + i, b = base + nl + j;
                pw[(size_t) b * (b - 1) / 2 + a] = wab * t->vol[lv[a]] * t->vol[lv[b]];
            }
        return nl + nr;
    }
};
struct PsGroup { std::vector<int> members; std::vector<double> weight; };      // what a node hands to its parent (`sprf`)
struct PsSmall { int node; std::vector<int> members; };
struct PsJoin { int node; PsGroup a, b; };
}

extern "C" int g2g_pairsum(g2g_ctx *ctx, const g2g_params *prm, int many, int len, const uint8_t *codes, const g2g_tree *tree,
                           int use_pw, double *out)
{
    if (!ctx || !prm || !codes || !tree || !out || many < 1 || len < 1) return G2G_ERR_ARG;
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    if (tree->n_nodes != 2 * many - 1) { g2g_set_error("%s", "g2g_pairsum: the tree must have 2 * many - 1 nodes"); return G2G_ERR_ARG; }
    *out = 0;
    if (many < 2) return G2G_OK;
    if ((float) prm->tgapf != 1.f || prm->u0 != 0) { g2g_set_error("%s", "g2g_pairsum: terminal-gap discount / ether term are not on this path"); return G2G_ERR_MODE; }
    if (prm->ls > 2 && prm->k1 > 32) return G2G_ERR_MODE;
    HIPCHK(hipSetDevice(ctx->device));
    PsTree T; T.t = tree;
    int root = 0;
    while (tree->parent[root] >= 0) root = tree->parent[root];
    const int ndesc_thr = 60;                                           // maln.h:37
    // the recursion, unrolled into two work lists: naive nodes (GPU kernel below) and joins of two groups (calcscore_grp)
    std::vector<PsSmall> small;
    std::vector<PsJoin> joins;
    std::function<PsGroup(int)> walk = [&](int node) -> PsGroup {
        PsGroup g;
        if (T.leaf(node)) {                                             // :789-794
            g.members.push_back(node);
            g.weight.push_back(tree->vol[node] / tree->vol[tree->parent[node]]);
        } else if (T.ndesc(node) <= ndesc_thr) {                        // :795-806 collectleaf
            T.leaves(node, g.members);
            for (int l : g.members) g.weight.push_back(tree->vol[l] * (tree->cur[node] / tree->vol[node]));
            PsSmall s; s.node = node; s.members = g.members;
            small.push_back(s);
        } else {                                                        // :807-819
            PsJoin j;
            j.node = node;
            j.a = walk(tree->left[node]);
            j.b = walk(tree->right[node]);
            g.members = j.a.members; g.members.insert(g.members.end(), j.b.members.begin(), j.b.members.end());
            for (double w : j.a.weight) g.weight.push_back(tree->cur[node] * w);     // fuseseq(sprf, sqs, (FTYPE) node->cur)
            for (double w : j.b.weight) g.weight.push_back(tree->cur[node] * w);
            joins.push_back(j);
        }
        return g;
    };
    walk(root);
    std::vector<double> sval(small.size(), 0.), jval(joins.size(), 0.);
    // ---- naive nodes: one workgroup each ----
    if (!small.empty()) {
        size_t bytes = 0;
        auto al = [](size_t x) { return (x + 255) & ~(size_t) 255; };
        const size_t nm = (size_t) prm->simdim * prm->simrows;
        struct Off { size_t codes, pw, gla, old, colv, coll; };
        std::vector<Off> off(small.size());
        const size_t o_mtx = 0; bytes = al(8 * nm);
        const size_t o_args = bytes; bytes = al(bytes + sizeof(PairsumArgs) * small.size());
        const size_t o_out = bytes; bytes = al(bytes + 8 * small.size());
        for (size_t k = 0; k < small.size(); ++k) {
            const size_t n = small[k].members.size();
            off[k].codes = bytes; bytes = al(bytes + (size_t) len * n);
            off[k].pw = bytes; bytes = al(bytes + 8 * (n * (n - 1) / 2 + 1));
            off[k].gla = bytes; bytes = al(bytes + 4 * (size_t) (len + 1) * n);
            off[k].old = bytes; bytes = al(bytes + 4 * (size_t) len * n);
            off[k].colv = bytes; bytes = al(bytes + 8 * (size_t) len);
            off[k].coll = bytes; bytes = al(bytes + 8 * (size_t) len);
        }
        std::vector<char> img(bytes, 0);
        memcpy(img.data() + o_mtx, prm->simmtx, 8 * nm);
        char *dev = 0;
        HIPCHK(hipMalloc((void **) &dev, bytes));
        for (size_t k = 0; k < small.size(); ++k) {
            const std::vector<int> &mb = small[k].members;
            const int n = (int) mb.size();
            uint8_t *c = (uint8_t *) img.data() + off[k].codes;
            for (int r = 0; r < len; ++r) for (int i = 0; i < n; ++i) c[(size_t) r * n + i] = codes[(size_t) r * many + mb[i]];
            if (use_pw) {
                std::vector<double> pw((size_t) n * (n - 1) / 2 + 1, 0.);
                T.repairwt(small[k].node, 0, mb, pw);
                memcpy(img.data() + off[k].pw, pw.data(), 8 * pw.size());
            }
            PairsumArgs a;
            a.codes = (const uint8_t *) (dev + off[k].codes); a.pw = use_pw ? (const double *) (dev + off[k].pw) : 0;
            a.simmtx = (const double *) (dev + o_mtx); a.simdim = prm->simdim;
            a.n = n; a.len = len; a.ls3 = prm->ls > 2; a.k1 = a.ls3 ? prm->k1 : 1;          // Msap::Msap: codonk1 = alnprm.k1 (maln2.cc:183)
            a.basic_gop = (double) ((float) prm->scale * -(float) prm->v);                   // :192
            a.diffu = a.ls3 ? (double) ((float) prm->scale * ((float) prm->u - (float) prm->u1)) : 0;   // :184
            a.gla = (int *) (dev + off[k].gla); a.old = (int *) (dev + off[k].old);
            a.colv = (double *) (dev + off[k].colv); a.coll = (double *) (dev + off[k].coll);
            a.out = (double *) (dev + o_out) + k;
            memcpy(img.data() + o_args + sizeof(PairsumArgs) * k, &a, sizeof a);
        }
        hipError_t e = hipMemcpyAsync(dev, img.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) { hipLaunchKernelGGL(g2g_pairsum_kernel, dim3((unsigned) small.size()), dim3(256), 0, ctx->stream, (const PairsumArgs *) (dev + o_args)); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(sval.data(), dev + o_out, 8 * small.size(), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(dev);
        if (e != hipSuccess) { g2g_set_error("g2g_pairsum: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_DEVICE; }
    }
    // ---- joins: calcscore_grp (fspscore.cc:624-659) = PwdM + SpScore::calcJxt (fspscore.h:178-199) on the two groups as they
    //      stand in the MSA -- level 1 builders, then calcSpScore along the trivial skeleton with the long-gap weight set to 0:
    //      calcJxt returns scr - tgap * v, which is calcSkl's unrescaled value without its diff_u * lunp term ----
    if (!joins.empty()) {
        const int nj = (int) joins.size();
        std::vector<g2g_group *> ga((size_t) nj, (g2g_group *) 0), gb((size_t) nj, (g2g_group *) 0);
        std::vector<g2g_pwdm *> pw((size_t) nj, (g2g_pwdm *) 0);
        std::vector<const g2g_problem *> probs((size_t) nj);
        int rc = G2G_OK;
        for (int k = 0; k < nj && rc == G2G_OK; ++k) {
            const PsGroup *gs[2] = {&joins[k].a, &joins[k].b};
            g2g_group *made[2] = {0, 0};
            for (int s = 0; s < 2; ++s) {
                const int n = (int) gs[s]->members.size();
                std::vector<uint8_t> c((size_t) len * n);
                for (int r = 0; r < len; ++r) for (int i = 0; i < n; ++i) c[(size_t) r * n + i] = codes[(size_t) r * many + gs[s]->members[i]];
                made[s] = g2g_group_create(ctx, prm, n, len, c.data(), gs[s]->weight.data());
            }
            ga[k] = made[0]; gb[k] = made[1];
            int swp = 0;
            if (ga[k] && gb[k]) pw[k] = g2g_pwdm_create(ctx, prm, ga[k], gb[k], &swp);
            if (!pw[k]) { rc = G2G_ERR_ARG; break; }
            probs[k] = g2g_pwdm_problem(pw[k]);
        }
        g2g_batch *b = 0;
        if (rc == G2G_OK) rc = g2g_batch_prepare(ctx, nj, probs.data(), &b);
        if (rc == G2G_OK) {
            std::vector<g2g_spparams> sp((size_t) nj);
            std::vector<g2g_skl> sk(2 * (size_t) nj);
            std::vector<const g2g_skl *> skp((size_t) nj);
            std::vector<int> ns((size_t) nj, 2);
            std::vector<g2g_fstat> fs((size_t) nj);
            for (int k = 0; k < nj; ++k) {
                g2g_pwdm_spparams(pw[k], &sp[k]);
                sp[k].diff_u = 0; sp[k].flags = G2G_SP_NOSTATS;
                sk[2 * k].m = probs[k]->a.left; sk[2 * k].n = probs[k]->b.left;
                sk[2 * k + 1].m = probs[k]->a.right; sk[2 * k + 1].n = probs[k]->b.right;
                skp[k] = &sk[2 * k];
            }
            rc = g2g_batch_spscore(b, sp.data(), skp.data(), ns.data(), fs.data());
            for (int k = 0; k < nj && rc == G2G_OK; ++k) { if (fs[k].status) rc = fs[k].status; jval[k] = fs[k].raw; }
        }
        if (b) g2g_batch_free(b);
        for (int k = 0; k < nj; ++k) { if (pw[k]) g2g_pwdm_free(pw[k]); if (ga[k]) g2g_group_free(ga[k]); if (gb[k]) g2g_group_free(gb[k]); }
        if (rc != G2G_OK) return rc;
    }
    // ---- the sum, in the order the recursion forms it: scr(node) = scr(left) + scr(right); scr += calcscore_grp ----
    std::function<double(int)> total = [&](int node) -> double {
        if (T.leaf(node)) return 0.;
        if (T.ndesc(node) <= ndesc_thr) { for (size_t k = 0; k < small.size(); ++k) if (small[k].node == node) return sval[k]; return 0.; }
        double scr = total(tree->left[node]);
        scr += total(tree->right[node]);
        for (size_t k = 0; k < joins.size(); ++k) if (joins[k].node == node) { scr += jval[k]; break; }
        return scr;
    };
    *out = total(root);
    return G2G_OK;
}
