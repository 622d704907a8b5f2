// g2g_host.cpp -- host side of level 1 (include/g2g.h): the operator surface mSeq / PwdM / align2().
//
// Everything the reference does on the CPU *around* the DP for one align2() call is restated here in
// C++: residue/terminal-gap normalisation, column thickness, frequency and
// profile vectors, static gap profiles, alignment-mode / scorer selection, the band, stdskl and the
// end check with its sh = -100 retry.  The DP itself (row a1-a5) is NOT here: g2g_align2*() hand the
// flattened problem to the GPU engine (g2g_engine.hip) and there is no host fallback.
//
// Residue codes are the reference's (src/cmn.h:111-114, src/seq.h:76-77): nil 0, gap 1, then
// protein AMB 2, ALA 3 ... VAL 22, ASX 23, GLX 24; nucleotide A 2, C 3, G 5, T 9 (IUPAC bit codes + 1).
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <vector>
#include "../../include/g2g.h"
#include "g2g_internal.h"
#include "g2g_group.h"

namespace {

enum { NIL_CODE = 0, GAP_CODE = 1 };
enum { MOLC_PROTEIN = 1, MOLC_DNA = 2 };
enum { RAWSEQ = 0, VECTOR = 1, VECPRO = 3 };                 // src/cmn.h:109
enum { ASN = 5, ASP = 6, GLN = 8, GLU = 9, ASX = 23, GLX = 24 };   // src/cmn.h:112-113
const int LARGEN = INT_MAX / 4 * 3;                          // src/aln.h:37
const int NOL = 3;                                           // src/aln.h:29
const int thr_gfq_21 = 8, thr_gfq_22 = 14;                   // src/maln.h:35-36

inline bool IsGap(uint8_t c) { return c <= GAP_CODE; }

struct Gfreq { int glen; double freq; int nres; };           // src/gfreq.h:25
const Gfreq zerogfq = {0, 0, 0}, delmgfq = {-1, 0, 0};
inline bool neogfq(const Gfreq &g) { return g.glen >= 0; }

}   // namespace

namespace {

// mSeq::gapdensity / postgapdensity, src/mseq.h:148-160 (pointer order == position order per member)
double gapdensity(const g2g_group &g, int pos, int i)
{
    const uint8_t c = g.at(pos, i);
    if (c > GAP_CODE) return 0;
    if (c == GAP_CODE || !g.has_internalres) return 1;
    if (pos < g.internal_pos[i]) return g.exgl ? 0 : g.tgapf;
    return g.exgr ? 0 : g.tgapf;
}
double postgapdensity(const g2g_group &g, int pos, int i)
{
    if (!g.has_internalres) return 1;    // the reference would dereference NULL only for nil codes, which then do not exist
    if (g.at(pos, i) == NIL_CODE && pos < g.internal_pos[i]) return g.exgl ? 0 : g.tgapf;
    if (g.at(pos + 1, i) == NIL_CODE && pos >= g.internal_pos[i]) return g.exgr ? 0 : g.tgapf;
    return 1;
}

// Seq::exg_seq(gl, gr), src/seq.cc:858-888 (algmode.qck == 0)
void exg_seq(g2g_group &g, int gl, int gr)
{
    g.exgl = gl ? 1 : 0;
    g.exgr = gr ? 1 : 0;
    gl = gl || g.tgapf < 1;
    gr = gr || g.tgapf < 1;
    const uint8_t codel = gl ? NIL_CODE : GAP_CODE, coder = gr ? NIL_CODE : GAP_CODE;
    for (int i = 0; i < g.many; ++i) { g.at(-1, i) = codel; g.at(g.len, i) = coder; }
    g.nils = gl || gr;
    for (int i = 0; i < g.many; ++i) {
        int s = 0;
        for ( ; s < g.len; ++s) {
            if (IsGap(g.at(s, i))) g.at(s, i) = codel; else break;
        }
        if (s >= g.len && (gl || !gr)) continue;
        for (s = g.len - 1; s > 0; --s) {
            if (IsGap(g.at(s, i))) g.at(s, i) = coder; else break;
        }
    }
}

// ---- column thickness -----------------------------------------------------------------------------------------
// What the reference leaves in mSeq's SeqThk table (mSeq::mkthick, src/mseq.cc:149-354, without the quick-mode DiThk
// part): per table row {cfq, dfq, efq} = weight of the members holding a residue, a gap, and the weight that counts as
// "present" for an unpaired column of the other group.  Three kinds of groups:
//   * no gap and no terminal nil anywhere : one constant row;
//   * terminal nils only                  : rows for the leading zone, one interior row, rows for the trailing zone;
//   * internal gaps                       : one row per column.
// Members inside their leading / trailing run of nil codes count with the terminal-gap factor of that end.  Bit-exact
// sums fix the ORDER of the additions: members ascending, except in the trailing zone, which the reference scans from
// the last member down.
struct ColumnTally { double nil, gap, res; int n_nil; };

void mkthick(g2g_group &g)
{
    if (g.thk_done) return;
    g.thk_done = true;
    const int many = g.many, len = g.len;
    const bool own_weights = g.has_weight && many > 1;
    auto weight_of = [&](int i) { return own_weights ? g.weight[i] : 1.0; };
    if (g.sumwt == 0) {
        if (own_weights) for (int i = 0; i < many; ++i) g.sumwt += g.weight[i];
        else g.sumwt = many;
    }
    const double total = g.sumwt;
    const double lead_f = g.exgl ? 0 : g.tgapf, trail_f = g.exgr ? 0 : g.tgapf;
    g.has_internalres = g.dels || g.nils;
    if (g.has_internalres) g.internal_pos.assign(many, 0);
    const int rows = g.thk_len = g.dels ? len : (g.has_internalres ? 2 : 0);
    g.thk.assign((size_t) (rows + 2) * 3, 0.);
    auto put = [&](int row, double cfq, double dfq, double efq) { double *t = g.T(row); t[0] = cfq; t[1] = dfq; t[2] = efq; };
    if (rows == 0) { put(-1, total, 0, total); put(0, total, 0, total); return; }
    put(-1, 0, total * lead_f, total * lead_f);              // before the first column: everybody is a leading end gap
    put(rows, 0, total * trail_f, 0);                         // behind the last column
    // tally of one column, members in the given order
    auto tally = [&](int col, bool descending, auto &&on_member) {
        ColumnTally c = {0, 0, 0, 0};
        for (int k = 0; k < many; ++k) {
            const int i = descending ? many - 1 - k : k;
            const double w = weight_of(i);
            const uint8_t code = g.at(col, i);
            if (code == NIL_CODE) { c.nil += w; ++c.n_nil; }
            else {
                on_member(i, w);
                if (code == GAP_CODE) c.gap += w; else c.res += w;
            }
        }
        return c;
    };
    // leading zone: columns up to and including the first one without a nil; remembers where each member's body starts
    int col = 0, row = 0;
    if (lead_f < 1.) {
        std::vector<char> waiting(many, 1);
        for (bool more = true; more && col < len; ++col, ++row) {
            const ColumnTally c = tally(col, false, [&](int i, double) { if (waiting[i]) { waiting[i] = 0; g.internal_pos[i] = col; } });
            put(row, c.res, c.gap + c.nil * lead_f, c.gap + c.res + c.nil * lead_f);
            more = c.n_nil != 0;
        }
    } else put(row, total, 0, total);                         // (has_internalres holds here: every body starts at column 0)
    const int body_row = row;
    // trailing zone, from the last column backwards to the first one without a nil.  efq: members seen for the first time
    // (coming from the right) count like the nils, the others fully.
    int end_col = len;
    if (trail_f < 1.) {
        std::vector<char> waiting(many, 1);
        int r = rows;
        for (bool more = true; more && end_col > 0; --end_col) {
            double fresh = 0, seen = 0;
            const ColumnTally c = tally(end_col - 1, true, [&](int i, double w) { if (waiting[i]) { waiting[i] = 0; fresh += w; } else seen += w; });
            if (--r >= -1) put(r, c.res, c.gap + c.nil * trail_f, seen + (c.nil + fresh) * trail_f);
            more = c.n_nil != 0;
        }
    }
    // interior columns of a group with internal gaps: gap weight against the rest
    if (g.dels)
        for (row = body_row; col < end_col; ++col, ++row) {
            double gap = 0, rest = 0;
            for (int i = 0; i < many; ++i) { if (g.at(col, i) == GAP_CODE) gap += weight_of(i); else rest += weight_of(i); }
            put(row, rest, gap, total);
        }
}

// SeqThk per position as mSeqItr yields it (src/mseq.h:222-250, src/mseq.cc:768-790)
void flatten_thk(g2g_group &g)
{
    const int mode = g.dels ? 2 : (g.nils ? 1 : 0);
    g.thk_pos.resize((size_t) (g.len + 2) * 3);
    for (int pos = -1; pos <= g.len; ++pos) {
        int j;
        if (mode == 2) j = pos;
        else if (mode == 1) j = pos < 0 ? -1 : (pos >= g.len - 1 ? 1 : 0);
        else j = -1;
        const double *t = g.T(j);
        double *o = &g.thk_pos[(size_t) (pos + 1) * 3];
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
    }
}

// ---- static gap profiles ---------------------------------------------------------------------------------------
// The three per-column views the reference's Gfq holds (Gfq::Gfq(mSeq*) and seq2gfq, src/gfreq.cc:134-312), built here from
// an explicit model instead of its three scratch arrays:
//
//   a GAP CLASS is the set of members whose current gap run opened at the same column, so all of them have the same
//   running length; classes are kept youngest first (= ascending length).  Per column a class has two faces:
//     t : weight still inside the run   (opening sum of weight x gap density, minus what members took out when they left)
//     s : weight of the members whose run ENDS at this column (they hold a residue here)
//   Besides the classes a column has: `opening` (t entry of length 0: runs that start here), `plain` (s entry of length 0:
//   residues that follow a residue), `tail` (s entry behind all classes: residues that follow a leading nil run of zero
//   density) and `post` (head of the r view: residue weight x density of the gap that may follow).
//   Views of a column:  s = [plain] + classes.s + [tail], in suffix-sum form;  t = [opening] + classes.t;
//                       r = [post] + t with every length + 1.
//   Entries nobody belongs to are dropped -- except in t when a trailing nil run starts at this column, which keeps them.
//
// What is part of the contract because sums are floating point: additions happen in member order; a class's t weight is
// the running value (added at the opening column, SUBTRACTED member by member later), never recomputed from its members.
// Two reference behaviours that are kept on purpose: the tail entry's length is the run length of the FIRST member that
// opens it, and a member of that kind arriving after a member that left a class is booked on THAT class's s face (the
// reference reuses one cursor for both, src/gfreq.cc:190-207).
struct GapEntry { int len; double w; int n; };
struct GapClass { int len; double t_w; int t_n; double s_w; int s_n; };

struct GfqBuilder {
    g2g_group &g;
    std::vector<GapClass> classes;       // youngest (shortest) first
    std::vector<int> run;                // per member: length of the gap / nil run it is in (0: none)
    std::vector<GapEntry> view[3];       // the column's packed views
    std::vector<GapClass> next;          // (scratch of column(): the classes that live on)
    std::vector<int> pick, seen;         // members column() visits; per member the last column it was visited at
    int t_count;                         // entries of the previous column's t view (hetero is its maximum)
    explicit GfqBuilder(g2g_group &gg) : g(gg), t_count(1) {}

    void column(int pos)
    {
        // (a t view that kept only an unborn `opening` entry -- see below -- starts with length -1: whatever stands behind it
        //  is dead for every reader, the reference's included, and falls away at the next packing)
        size_t live = 0;
        while (live < classes.size() && classes[live].len >= 0) ++live;
        for (size_t k = 0; k < live; ++k) { classes[k].len += 1; classes[k].s_w = 0; classes[k].s_n = 0; }
        GapEntry opening = {-1, 0, 0}, plain = {0, 0, 0}, tail = {-1, 0, 0}, post = {0, 0, 0};
        int cursor = -1;                  // -1: the tail entry; k >= 0: class k (see the note above)
        int trailing_starts = 0;
        // A member that goes on in a gap run (a gap code here AND in the column before: four cells of five in a refinement's MSA) only
        // gets older -- a gap code has density 1 whatever the end factors are.  Such members are not visited at all: the others are
        // picked by a branch-free pass over the two columns' codes, and a member's run length is brought up to date when it is visited
        // again (seen[i]: the last column it was visited at).
        const uint8_t *cur = &g.seq[(size_t) (pos + 1) * g.many], *prv = &g.seq[(size_t) pos * g.many];
        int npick = 0;
        const int first = pos == 0;            // (column -1 may hold gap codes too, but no run has started there: column 0 visits everybody)
        for (int i = 0; i < g.many; ++i) { pick[npick] = i; npick += (cur[i] != GAP_CODE) | (prv[i] != GAP_CODE) | first; }
        for (int q = 0; q < npick; ++q) {
            const int i = pick[q];
            const uint8_t code = cur[i];
            int &len_i = run[i];
            len_i += pos - 1 - seen[i];        // (the columns it was skipped at: one more gap each)
            seen[i] = pos;
            const double w = g.has_weight ? g.weight[i] : 1;
            const double dens_prev = gapdensity(g, pos - 1, i), dens_here = gapdensity(g, pos, i), dens_next = postgapdensity(g, pos, i);
            if (dens_here > 0) {                                   // inside a gap that counts
                if (len_i == 0) { opening.len = 0; opening.w += w * dens_here; ++opening.n; }
                len_i += 1;
                continue;
            }
            if (code == NIL_CODE) {                                // terminal nil of zero density
                if (len_i == 0) ++trailing_starts;
                len_i += 1;
                continue;
            }
            // a residue
            const bool after_free_nil = g.at(pos - 1, i) == NIL_CODE && dens_prev == 0;
            if (!after_free_nil && dens_next > 0) { post.w += w * dens_next; ++post.n; }
            if (after_free_nil) {
                if (cursor < 0) {
                    if (tail.len < 0) { tail.len = len_i; tail.w = 0; tail.n = 0; }
                    tail.w += w; ++tail.n;
                } else { classes[cursor].s_w += w; ++classes[cursor].s_n; }
                if (dens_next > 0) { post.w += w * dens_next; ++post.n; }
            } else if (len_i) {                                    // leaves the class of its length, if there is one
                int k = -1;
                for (int c = 0; c < (int) classes.size() && c < t_count; ++c) if (classes[c].len == len_i) { k = c; break; }
                if (k >= 0) {
                    classes[k].t_w -= dens_prev * w; --classes[k].t_n;
                    classes[k].s_w += w; ++classes[k].s_n;
                    cursor = k;
                }
            } else { plain.w += w; ++plain.n; }
            len_i = 0;
        }
        // s view: members present, suffix sums from the far end
        view[0].clear();
        if (plain.n) view[0].push_back(plain);
        for (size_t k = 0; k < live; ++k) if (classes[k].s_n) view[0].push_back(GapEntry{classes[k].len, classes[k].s_w, classes[k].s_n});
        if (tail.len >= 0 && tail.n) view[0].push_back(tail);
        { double acc = 0; for (size_t k = view[0].size(); k-- > 0; ) view[0][k].w = acc += view[0][k].w; }
        // t view and the classes that live on
        view[1].clear();
        next.clear();
        if (trailing_starts || opening.n) {
            view[1].push_back(opening);
            next.push_back(GapClass{opening.len, opening.w, opening.n, 0, 0});
        }
        for (size_t k = 0; k < live; ++k) {
            const GapClass &c = classes[k];
            if (trailing_starts || c.t_n) { view[1].push_back(GapEntry{c.len, c.t_w, c.t_n}); next.push_back(c); }
        }
        classes.swap(next);
        t_count = (int) view[1].size();
        // r view
        view[2].clear();
        if (post.n) view[2].push_back(post);
        for (const GapEntry &e : view[1]) { if (e.len < 0) break; view[2].push_back(GapEntry{e.len + 1, e.w, e.n}); }
    }

    static void emit(GapProfile *gp, int v, const std::vector<GapEntry> &l)
    {
        gp->off[v].push_back((int32_t) gp->glen[v].size());
        for (const GapEntry &e : l) { if (e.len < 0) break; gp->glen[v].push_back(e.len); gp->freq[v].push_back(e.w); }
        gp->glen[v].push_back(-1); gp->freq[v].push_back(0);           // the terminator is part of the pool
    }

    GapProfile *build()
    {
        // only groups with inex.dels get a gap profile (mSeq::convseq, src/mseq.cc:507)
        GapProfile *gp = new GapProfile();
        run.assign(g.many, 0);
        pick.assign((size_t) g.many + 1, 0);
        seen.assign(g.many, -1);
        classes.clear();
        for (int v = 0; v < 3; ++v) {        // (pools grow by amortised doubling otherwise: a few entries per column is the rule)
            gp->off[v].reserve((size_t) g.len + 3);
            gp->glen[v].reserve((size_t) 6 * g.len + 16); gp->freq[v].reserve((size_t) 6 * g.len + 16);
        }
        // position -1 (src/gfreq.cc:264-282): everybody is a leading end gap of weight sumwt x terminal-gap factor
        const double lead_f = g.exgl ? 0 : g.tgapf;
        const std::vector<GapEntry> none;
        emit(gp, 0, none);
        if (lead_f > 0) {
            const std::vector<GapEntry> lead(1, GapEntry{0, g.sumwt * lead_f, g.many});
            emit(gp, 1, lead); emit(gp, 2, lead);
        } else { emit(gp, 1, none); emit(gp, 2, none); }
        int most = 0;
        for (int pos = 0; pos < g.len; ++pos) {
            column(pos);
            most = std::max(most, t_count);
            for (int v = 0; v < 3; ++v) emit(gp, v, view[v]);
        }
        for (int v = 0; v < 3; ++v) gp->off[v].push_back((int32_t) gp->glen[v].size());
        gp->hetero = most + 1;
        return gp;
    }
};

// ---- frequency / profile vectors: mSeq::convseq, src/mseq.cc:447-587 -------------------------------
const int nbits_t[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};        // src/mseq.h:42-43
const int amblist[] = {2,3,2,3,5,2,5,3,5,2,3,5,9,2,9,3,9,2,3,9,5,9,2,5,9,3,5,9,2,3,5,9};   // :45-46 (A=2 C=3 G=5 T=9)
const int ambaddr[] = {0, 0, 0, 1, 2, 4, 5, 7, 9, 12, 13, 15, 17, 20, 22, 25, 28};      // :49-50
const int decompact[6] = {0, 1, 2, 3, 5, 9};                                       // :41

void ntor(double *v, int k, double w)                        // mSeq::ntor, src/mseq.cc:366-384
{
    switch (k) {
    case 0: v[NIL_CODE] += w; break;
    case 1: v[GAP_CODE] += w; break;
    case 2: v[2] += w; break;
    case 3: v[3] += w; break;
    case 5: v[4] += w; break;
    case 9: v[5] += w; break;
    default: {
        if (k < 1 || k > 16) break;
        const int m = nbits_t[k - GAP_CODE];
        w /= m;
        const int *j = amblist + ambaddr[k];
        for (int n = 0; n < m; ++n) v[*j++] += w;
        break; }
    }
}

struct Matrix { const double *m; int dim, rows; double at(int i, int j) const { return m[(size_t) i * dim + j]; } };

void convseq(g2g_group &g, int vect, const Matrix &sm)
{
    // (vectors rebuilt on the host below make a device twin of the old ones void)
    if ((g.vect == RAWSEQ && vect != RAWSEQ) || (g.vect != vect && vect == VECPRO && sm.m)) g.dev.pseq = 0;
    // (mkthick and the gap profile are made by the caller in the reference's order)
    const int many = g.many, len = g.len;
    if (g.vect == RAWSEQ) {
        if (vect == RAWSEQ) return;
        g.felm = (g.molc == MOLC_PROTEIN) ? ASX : 6;          // prepseq, src/mseq.cc:485-502 (simmtx not set yet)
        g.nelm = g.felm + 1;
        const int eth = g.nelm - 1;
        g.pseq.assign((size_t) (len + 2) * g.nelm, 0.);
        std::vector<double> wtb(many);
        for (int i = 0; i < many; ++i) wtb[i] = g.has_weight ? g.weight[i] : 1;
        for (int pos = -1; pos <= len; ++pos) {
            double *v = &g.pseq[(size_t) (pos + 1) * g.nelm];
            if (g.molc == MOLC_PROTEIN) {                     // aas2cvec :455-476
                for (int i = 0; i < many; ++i) {
                    const int k = g.at(pos, i);
                    const double wt = wtb[i];
                    if (k == ASX) { v[ASN] += wt / 2; v[ASP] += wt / 2; }
                    else if (k == GLX) { v[GLN] += wt / 2; v[GLU] += wt / 2; }
                    else if (k < g.nelm) v[k] += wt;
                }
            } else {                                          // nuc2cvec :447-453
                for (int i = 0; i < many; ++i) ntor(v, g.at(pos, i), wtb[i]);
            }
            // (the reference sums the gap members' weights a second time for the last element: the same addends in the same order as
            //  the gap element above -- nothing else is ever added to that one -- so it is the same double)
            v[eth] = v[GAP_CODE];
        }
        g.vect = VECTOR;
    }
    if (g.vect == vect) return;
    if (vect == VECPRO && sm.m) {
        const int felm = g.felm, nnelm = felm + sm.dim + 1, eth = nnelm - 1;
        std::vector<double> np((size_t) (len + 2) * nnelm, 0.);
        std::vector<double> smt((size_t) sm.dim * sm.rows, 0.);     // smt[j][i] = sm[i][j]
        for (int i = 0; i < sm.rows; ++i) for (int j = 0; j < sm.dim; ++j) smt[(size_t) j * sm.rows + i] = sm.at(i, j);
        for (int pos = -1; pos <= len; ++pos) {
            const double *w = &g.pseq[(size_t) (pos + 1) * g.nelm];
            double *nst = &np[(size_t) (pos + 1) * nnelm];
            double *v = nst + felm;
            for (int j = 0; j < felm; ++j) nst[j] = w[j];
            if (g.molc != MOLC_PROTEIN) {                     // profile_n :392-411
                v[NIL_CODE] = 0;
                for (int i = 1; i < felm; ++i) {
                    const int k = decompact[i];
                    v[k] = 0;
                    for (int j = 1; j < felm; ++j) v[k] += sm.at(k, decompact[j]) * w[j];
                }
                for (int i = 3 + 1; i < g.max_code; ++i) {    // `C` = 3
                    const int m = nbits_t[i - GAP_CODE];
                    if (m == 1) continue;
                    v[i] = 0;
                    const int *j = amblist + ambaddr[i];
                    for (int n = 0; n < m; ++n) v[i] += v[*j++];
                    v[i] /= m;
                }
            } else if (sm.dim == sm.rows) {                   // profile_p :413-424
                // (the reference's loops are i outside, j inside; here j outside over the TRANSPOSED matrix: every v[i] still gets its
                //  products in the order j = 1, 2, ..., each a separate multiply and add -- the same doubles -- and the inner loop runs
                //  over independent accumulators)
                v[NIL_CODE] = 0;
                for (int i = 1; i < felm; ++i) v[i] = 0;
                for (int j = 1; j < felm; ++j) {
                    const double wj = w[j];
                    const double *col = &smt[(size_t) j * sm.rows];
                    for (int i = 1; i < felm; ++i) v[i] += col[i] * wj;
                }
                v[ASX] = (v[ASN] + v[ASP]) / 2;
                v[GLX] = (v[GLN] + v[GLU]) / 2;
            } else {                                          // profile :426-435
                v[NIL_CODE] = 0;
                for (int i = 1; i < sm.dim; ++i) v[i] = 0;
                for (int j = 1; j < felm; ++j) {
                    const double wj = w[j];
                    const double *col = &smt[(size_t) j * sm.rows];
                    for (int i = 1; i < sm.dim; ++i) v[i] += col[i] * wj;
                }
            }
            nst[eth] = w[felm];
        }
        g.pseq.swap(np);
        g.nelm = nnelm;
        g.vect = VECPRO;
    }
}

void ensure_gfq(g2g_group &g)
{
    if (g.dels && !g.gfq) { GfqBuilder b(g); g.gfq = b.build(); }
}

}   // namespace

// <-> PwdM (src/maln.h:144-329, ctor src/maln2.cc:254-491) + PwdB (src/aln.h:228, src/aln2.cc:97-138)
struct g2g_pwdm {
    g2g_group *a, *b;
    g2g_params prm;
    int swp;
    int alnmode, a_mode, b_mode, aprof, bprof;
    g2g_problem prob;
    g2g_spparams sp;                 // PwdM::Vab and the PwdB gap-extension scalars calcSpScore needs
};

namespace {

void fill_side(g2g_group &g, g2g_side &s, bool ntv)
{
    memset(&s, 0, sizeof s);
    s.many = g.many; s.len = g.len; s.left = g.left; s.right = g.right;
    s.nils = g.nils; s.dels = g.dels;
    s.seq = g.seq.data();
    s.weight = 0;
    s.nelm = g.vect ? g.nelm : 0; s.felm = g.vect ? g.felm : 0;
    s.pseq = g.vect ? g.pseq.data() : 0;
    flatten_thk(g);
    s.thk = g.thk_pos.data();
    s.sumwt = g.sumwt;                                       // (set by mkthick)
    s.has_gfq = (g.gfq && g.dels) ? 1 : 0;
    if (s.has_gfq) {
        s.gfq.hetero = g.gfq->hetero;
        for (int v = 0; v < 3; ++v) {
            s.gfq.off[v] = g.gfq->off[v].data();
            s.gfq.glen[v] = g.gfq->glen[v].data();
            s.gfq.freq[v] = g.gfq->freq[v].data();
        }
    }
    if (g.dev.ctx && g.dev.seq) {            // device-resident twins (g2g_device_derive): only of what the host side hands out too
        static_assert(sizeof(int) == sizeof(int32_t), "int32");
        s.dev = &g.dev;
    }
    if (ntv) {
        const size_t n = (size_t) (g.len + 2) * g.many;
        g.gapdens.assign(n, 0.); g.postgapdens.assign(n, 0.);
        for (int pos = -1; pos < g.len; ++pos)
            for (int i = 0; i < g.many; ++i) {
                g.gapdens[(size_t) (pos + 1) * g.many + i] = gapdensity(g, pos, i);
                g.postgapdens[(size_t) (pos + 1) * g.many + i] = postgapdensity(g, pos, i);
            }
        s.gapdens = g.gapdens.data(); s.postgapdens = g.postgapdens.data();
    }
}

// stripe(), src/aln2.cc:156-174
void stripe(const g2g_group &a, const g2g_group &b, int sh, int *lw, int *up)
{
    if (sh < 0) {
        int shorter = std::min(a.right - a.left, b.right - b.left);
        sh = -sh * shorter / 100;
    }
    int u = b.right - a.right, l = b.left - a.left;
    if (u < l) std::swap(u, l);
    u += sh; l -= sh;
    int p;
    if ((p = b.right - a.left) < u) u = p;
    if ((p = b.left - a.right) > l) l = p;
    *lw = l; *up = u;
}

}   // namespace

extern "C" g2g_group *g2g_group_create(g2g_ctx *, const g2g_params *prm, int many, int len,
                                       const uint8_t *seq, const double *weight)
{
    if (!prm || many < 1 || len < 1 || !seq) { g2g_set_error("%s", "g2g_group_create: bad argument"); return NULL; }
    g2g_group *g = new g2g_group();
    g->many = many; g->len = len; g->left = 0; g->right = len;
    g->molc = prm->molc; g->max_code = prm->max_code;
    g->tgapf = prm->tgapf;
    g->seq.assign((size_t) (len + 2) * many, GAP_CODE);
    memcpy(&g->seq[many], seq, (size_t) len * many);
    g->has_weight = weight != 0;
    if (weight) g->weight.assign(weight, weight + many);
    g->dels = 0;                                             // Seq::test_gap_amb, src/seq.cc:890-906
    for (size_t k = 0; k < (size_t) len * many; ++k) if (IsGap(seq[k])) { g->dels = 1; break; }
    g->thk_done = false; g->sumwt = 0; g->thk_len = 0; g->has_internalres = false;
    g->vect = RAWSEQ; g->nelm = g->felm = 0; g->gfq = 0; g->nils = 0; g->exgl = g->exgr = 0;
    memset(&g->dev, 0, sizeof g->dev);
    exg_seq(*g, 0, 0);                                       // Prrn::gather / aln_main: global, both ends
    return g;
}

extern "C" void g2g_group_free(g2g_group *g) { delete g; }

// The PwdM constructor in three stages, so that the middle one -- the derived arrays of the two groups -- can run for one pair on
// host threads (g2g_pwdm_create) or for a whole batch of pairs on the device (g2g_pwdm_create_batch, g2g_build.hip).
namespace {
struct PwdmPlan {
    g2g_pwdm *P;
    g2g_group *a, *b;                 // after the swap
    int aprof, bprof, swp, alnmode, DvsP, Noll, codonk1;
    float f_scale, f_u, f_v, f_u1;
    double BasicGOP, BasicGEP, LongGOP, LongGEP, diffu;
};

// stage 1: PwdB, selAlnMode, the swap
bool pwdm_plan(const g2g_params *prm, g2g_group *ga, g2g_group *gb, PwdmPlan &L)
{
    if (!prm || !ga || !gb) { g2g_set_error("%s", "g2g_pwdm_create: bad argument"); return false; }
    if (prm->u0 > 0) { g2g_set_error("%s", "ether scorers (u0 > 0) are not on this path"); return false; }
    g2g_group *sq[2] = {ga, gb};
    // --- PwdB (before any swap) ---
    L.f_scale = (float) prm->scale; L.f_u = (float) prm->u; L.f_v = (float) prm->v; L.f_u1 = (float) prm->u1;
    L.DvsP = (ga->molc == MOLC_PROTEIN) + 2 * (gb->molc == MOLC_PROTEIN);
    L.Noll = std::max(2, std::min(NOL, (int) prm->ls));
    const double VabB = (double) (L.f_scale * ga->many * gb->many);          // axbscale, src/seq.h:1468 (float arithmetic)
    L.BasicGOP = (double) (-L.f_v * VabB); L.BasicGEP = (double) (-L.f_u * VabB); L.LongGEP = (double) (-L.f_u1 * VabB);
    L.diffu = L.LongGEP - L.BasicGEP;
    L.LongGOP = L.BasicGOP - L.diffu * prm->k1;
    const int step = (L.DvsP == 3) ? 1 : 3;
    L.codonk1 = prm->ls == 3 ? step * prm->k1 : LARGEN;
    // --- PwdM::selAlnMode, src/maln2.cc:81-154 ---
    int aprof = 0, bprof = 0, abgfq = 0;
    {   // advised_sim2 :43-60
        const int i = sq[0]->many < sq[1]->many, j = 1 - i;
        const int ni = sq[i]->many, nj = sq[j]->many, nt = 2 * nj + ni;
        if (nt >= thr_gfq_21) {
            bool apf = nt >= thr_gfq_22 || nj == 1;
            bool bpf = nj > sq[j]->max_code;
            if (i) std::swap(apf, bpf);
            aprof = apf; bprof = bpf;
            abgfq = 1;
        }
    }
    const bool agfq = sq[0]->dels, bgfq = sq[1]->dels;
    int alnmode;
    if (!agfq && !bgfq) alnmode = G2G_NGP_ALN;
    else if (!abgfq) alnmode = G2G_NTV_ALN;                  // htr == 0: no DNA x protein on this path
    else if (!agfq) alnmode = G2G_RHF_ALN;
    else if (!bgfq) alnmode = G2G_HLF_ALN;
    else alnmode = G2G_GPF_ALN;
    if (prm->banded) alnmode += G2G_NTV_ALN;
    int swp;
    switch (alnmode) {
    case G2G_HLF_ALN: case G2G_HLF_ALB: swp = 0; break;
    case G2G_RHF_ALN: case G2G_RHF_ALB: swp = 1; break;
    case G2G_GPF_ALN: case G2G_GPF_ALB: swp = !aprof && bprof; break;
    case G2G_NTV_ALN: swp = (sq[0]->right - sq[0]->left) < (sq[1]->right - sq[1]->left); break;
    default: swp = 0; break;                                 // a->inex.intr: no spliced input here
    }
    if (swp) { std::swap(sq[0], sq[1]); std::swap(aprof, bprof); }
    L.a = sq[0]; L.b = sq[1]; L.aprof = aprof; L.bprof = bprof; L.swp = swp; L.alnmode = alnmode;
    L.P = new g2g_pwdm();
    L.P->prm = *prm;
    return true;
}

// what stage 2 has to provide for each side (bits of G2G_NEED_*): mSeq::convseq begins with mkthick + the gap profile
// (src/mseq.cc:506-507); a profile side gets its frequency vectors, side a (or b when a is none) the profile vectors on top
void pwdm_needs(const PwdmPlan &L, int *need_a, int *need_b)
{
    *need_a = (L.a->dels ? G2G_NEED_GFQ : 0) | (L.aprof ? (G2G_NEED_VECTOR | G2G_NEED_VECPRO) : 0);
    *need_b = (L.b->dels ? G2G_NEED_GFQ : 0) | (L.bprof ? (G2G_NEED_VECTOR | (L.aprof ? 0 : G2G_NEED_VECPRO)) : 0);
}

// stage 2 on the host
void pwdm_derive_host(const g2g_params *prm, PwdmPlan &L)
{
    const Matrix sm = {prm->simmtx, prm->simdim, prm->simrows};
    g2g_group &a = *L.a, &b = *L.b;
    const int aprof = L.aprof, bprof = L.bprof;
    // exg_seq(lcl & 1, lcl & 2): algmode.lcl == 0 (global) -> already done at creation
    // The builders of one pair: thickness first (the gap profile reads its weight sum), then the gap profile of a, the vectors of
    // a and everything of b are independent of one another (they read the residues and write their own members): three tasks,
    // on helper threads while the machine has idle cores (a window of g2g_refine builds 2-16 pairs on 16+ cores; the big group's
    // gap profile and vectors are 10 ms each, the rest 1-2 ms), in line otherwise.  Same results either way.
    static const bool host_times = getenv("G2G_HOST_TIMES") != 0;
    static std::atomic<int> builders_active(0);
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = host_times ? tnow() : 0;
    mkthick(a);
    mkthick(b);
    double tt[3] = {0, 0, 0};
    auto task_gfq_a = [&]() { const double t = tnow(); ensure_gfq(a); tt[0] = tnow() - t; };
    auto task_vec_a = [&]() { const double t = tnow(); if (aprof) { convseq(a, VECTOR, sm); convseq(a, VECPRO, sm); } tt[1] = tnow() - t; };
    auto task_b = [&]() { const double t = tnow(); ensure_gfq(b); if (bprof) { convseq(b, VECTOR, sm); if (!aprof) convseq(b, VECPRO, sm); } tt[2] = tnow() - t; };
    const int active = builders_active.fetch_add(1) + 1;
    const unsigned cores = std::thread::hardware_concurrency();
    const bool split = !getenv("G2G_NO_BUILD_SPLIT") && (size_t) a.many * a.len + (size_t) b.many * b.len >= 100000 && cores >= 4 && (unsigned) active * 3 <= cores;
    // (std::thread's constructor throws std::system_error when the process is out of threads; nothing may cross the C ABI: a
    //  task whose helper could not be started runs in line)
    bool ran_a = false, ran_b = false;
    if (split) {
        std::thread t1, t2;
        try { t1 = std::thread(task_vec_a); ran_a = true; } catch (...) {}
        try { t2 = std::thread(task_b); ran_b = true; } catch (...) {}
        task_gfq_a();
        if (!ran_a) task_vec_a();
        if (!ran_b) task_b();
        if (t1.joinable()) t1.join();
        if (t2.joinable()) t2.join();
    } else { task_gfq_a(); task_vec_a(); task_b(); }
    builders_active.fetch_sub(1);
    if (host_times) fprintf(stderr, "[g2g_pwdm_create] a %d x %d, b %d x %d: %.2f ms (%s: gap profile of a %.2f, vectors of a %.2f, b %.2f)\n", a.many, a.len, b.many, b.len, tnow() - tb0, split ? "three tasks" : "in line", tt[0], tt[1], tt[2]);
}

// stage 3: scorer selection, the flattened problem, the band
void pwdm_finish(const g2g_params *prm, PwdmPlan &L)
{
    g2g_pwdm *P = L.P;
    g2g_group &a = *L.a, &b = *L.b;
    const int aprof = L.aprof, bprof = L.bprof, swp = L.swp, alnmode = L.alnmode, DvsP = L.DvsP;
    // --- rest of the PwdM ctor :266-285 ---
    const double *wta = a.has_weight ? a.weight.data() : 0, *wtb = b.has_weight ? b.weight.data() : 0;
    if (wta && !wtb && !bprof) { b.weight.assign(b.many, 1.); b.has_weight = true; wtb = b.weight.data(); }
    if (wtb && !wta && !aprof) { a.weight.assign(a.many, 1.); a.has_weight = true; wta = a.weight.data(); }
    const bool wwt = wta && wtb;
    const int a_mode = aprof ? 2 : a.many > 1, b_mode = bprof ? 2 : b.many > 1;
    int sim2 = -1, crg2 = 0;
    switch (3 * a_mode + b_mode) {                           // :347-399 (u0 == 0 branch)
    case 0: sim2 = G2G_SIM11; crg2 = 11; break;
    case 1: if (wtb) { sim2 = G2G_SIM12W; crg2 = 121; } else { sim2 = G2G_SIM12I; crg2 = 120; } break;
    case 2: sim2 = G2G_SIM13; break;
    case 3: if (wta) { sim2 = G2G_SIM21W; crg2 = 211; } else { sim2 = G2G_SIM21I; crg2 = 210; } break;
    case 4: if (wwt) { sim2 = G2G_SIM22W; crg2 = 221; } else { sim2 = G2G_SIM22I; crg2 = 220; } break;
    case 5: sim2 = wta ? G2G_SIM23W : G2G_SIM23I; break;
    case 6: sim2 = G2G_SIM31; break;
    case 7: sim2 = wtb ? G2G_SIM32W : G2G_SIM32I; break;
    case 8: sim2 = (DvsP == 0) ? G2G_SIM33N : G2G_SIM33; break;
    }
    P->a = &a; P->b = &b; P->swp = swp; P->alnmode = alnmode;
    P->a_mode = a_mode; P->b_mode = b_mode; P->aprof = aprof; P->bprof = bprof;
    // --- flatten ---
    g2g_problem &q = P->prob;
    memset(&q, 0, sizeof q);
    q.alnmode = alnmode; q.sim2_kind = sim2;
    const bool ntv = alnmode == G2G_NTV_ALB || alnmode == G2G_NTV_ALN;
    q.crg2_kind = ntv ? crg2 : 0;
    q.dvsp = DvsP;
    q.noll = L.Noll; q.codonk1 = L.codonk1;
    q.basic_gop = (double) (-L.f_scale * L.f_v);             // resetuab, src/maln2.cc:227-243 (float arithmetic)
    q.weighted_gop = (double) -L.f_v;
    q.u = (double) L.f_u;
    q.u2divu1 = L.BasicGEP < 0 ? L.LongGEP / L.BasicGEP : 0; // Fwd2c ctor, src/fwd2c.h:85-86
    q.v2divv1 = L.BasicGOP < 0 ? L.LongGOP / L.BasicGOP : 0;
    q.simmtx = prm->simmtx; q.simdim = prm->simdim; q.simrows = prm->simrows;
    fill_side(a, q.a, ntv);
    fill_side(b, q.b, ntv);
    q.a.weight = wta; q.b.weight = wtb;
    stripe(a, b, prm->sh, &q.lw, &q.up);
    {   // resetuab, src/maln2.cc:227-234: Vab = scale * wa * wb with the weight sums of the profile-mode sides
        const double wa = a_mode ? a.sumwt : 1, wb = b_mode ? b.sumwt : 1;
        P->sp.vab = (double) (prm->scale * wa * wb);
        P->sp.basic_gep = L.BasicGEP; P->sp.diffu = L.diffu;
        P->sp.diff_u = (double) (L.f_scale * (L.f_u - L.f_u1));  // resetuab, src/maln2.cc:233 (float arithmetic)
    }
}
}   // namespace

extern "C" g2g_pwdm *g2g_pwdm_create(g2g_ctx *, const g2g_params *prm, g2g_group *ga, g2g_group *gb, int *swapped)
{
    PwdmPlan L;
    if (!pwdm_plan(prm, ga, gb, L)) return NULL;
    pwdm_derive_host(prm, L);
    pwdm_finish(prm, L);
    if (swapped) *swapped = L.swp;
    return L.P;
}

// n PwdMs at once, the derived arrays of all their groups built ON THE DEVICE in one go (g2g_build.hip: thickness, frequency /
// profile vectors, gap profiles -- SURVEY.md section 8 rows a8 / a9); same objects, same arrays as n calls of g2g_pwdm_create.
// Groups the device builders do not take (nil codes: tgapf < 1; a group already vectorised half-way) are built on the host.
extern "C" int g2g_pwdm_create_batch(g2g_ctx *ctx, const g2g_params *prm, int n, g2g_group *const *ga, g2g_group *const *gb, int *swapped, g2g_pwdm **out)
{
    if (!ctx || !prm || n < 0 || (n && (!ga || !gb || !out))) { g2g_set_error("%s", "g2g_pwdm_create_batch: bad argument"); return G2G_ERR_ARG; }
    std::vector<PwdmPlan> L((size_t) n);
    for (int k = 0; k < n; ++k) out[k] = 0;
    for (int k = 0; k < n; ++k)
        if (!pwdm_plan(prm, ga[k], gb[k], L[(size_t) k])) { for (int j = 0; j < k; ++j) delete L[(size_t) j].P; return G2G_ERR_ARG; }
    // the groups and what each needs (a group may serve several pairs)
    std::vector<g2g_group *> groups;
    std::vector<int> need;
    auto want = [&](g2g_group *g, int bits) {
        for (size_t q = 0; q < groups.size(); ++q) if (groups[q] == g) { need[q] |= bits; return; }
        groups.push_back(g); need.push_back(bits);
    };
    for (int k = 0; k < n; ++k) { int na, nb; pwdm_needs(L[(size_t) k], &na, &nb); want(L[(size_t) k].a, na); want(L[(size_t) k].b, nb); }
    std::vector<g2g_group *> dev;
    std::vector<int> dneed;
    const bool use_dev = !g2g_get_option(ctx, "NO_DEVICE_BUILD");
    for (size_t q = 0; q < groups.size(); ++q) {
        g2g_group &g = *groups[q];
        int bits = need[q];
        if (g.gfq) bits &= ~G2G_NEED_GFQ;
        if (g.vect == VECPRO || (g.vect == VECTOR && !(bits & G2G_NEED_VECPRO))) bits &= ~(G2G_NEED_VECTOR | G2G_NEED_VECPRO);
        const bool halfway = g.vect == VECTOR && (bits & G2G_NEED_VECPRO);
        const bool work = (g.dels && !g.thk_done) || bits;
        if (use_dev && work && !halfway && !g.nils && !g.exgl && !g.exgr) { dev.push_back(&g); dneed.push_back(bits); }
    }
    if (!dev.empty()) {
        const int rc = g2g_device_derive(ctx, prm, (int) dev.size(), dev.data(), dneed.data());
        if (rc != G2G_OK && rc != G2G_ERR_MODE) { for (auto &l : L) delete l.P; return rc; }      // (G2G_ERR_MODE: not taken -- the host builds them below)
    }
    // whatever is still missing (gap-free groups' constant thickness rows; groups the device did not take): host, pairs in parallel
    {
        std::atomic<int> next(0);
        auto work = [&]() { for (int k; (k = next.fetch_add(1)) < n; ) pwdm_derive_host(prm, L[(size_t) k]); };
        unsigned nthr = std::thread::hardware_concurrency();
        if (nthr > 16) nthr = 16;
        if (nthr < 1) nthr = 1;
        // (a group shared by two pairs must not be derived by two threads at once: shared groups make the loop serial)
        if (groups.size() < 2 * (size_t) n) nthr = 1;
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthr && (int) t < n; ++t) { try { th.emplace_back(work); } catch (...) { break; } }
        work();
        for (auto &t : th) t.join();
    }
    for (int k = 0; k < n; ++k) {
        pwdm_finish(prm, L[(size_t) k]);
        out[k] = L[(size_t) k].P;
        if (swapped) swapped[k] = L[(size_t) k].swp;
    }
    return G2G_OK;
}

extern "C" void g2g_pwdm_free(g2g_pwdm *p) { delete p; }
extern "C" int g2g_pwdm_spparams(const g2g_pwdm *p, g2g_spparams *sp)
{
    if (!p || !sp) return G2G_ERR_ARG;
    *sp = p->sp;
    return G2G_OK;
}
// <-> PreSpScore::calcSpScore(Gsinfo*) for every (PwdM, skeleton) pair, src/fspscore.cc:584-622
extern "C" int g2g_spscore_batch_flags(g2g_ctx *ctx, int n, g2g_pwdm *const *p, const g2g_skl *const *skl, const int *nskl, int flags, g2g_fstat *out);
extern "C" int g2g_spscore_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *p, const g2g_skl *const *skl, const int *nskl, g2g_fstat *out)
{
    return g2g_spscore_batch_flags(ctx, n, p, skl, nskl, 0, out);
}
extern "C" int g2g_spscore_batch_flags(g2g_ctx *ctx, int n, g2g_pwdm *const *p, const g2g_skl *const *skl, const int *nskl, int flags, g2g_fstat *out)
{
    if (!ctx || n < 0 || (n && (!p || !skl || !nskl || !out))) return G2G_ERR_ARG;
    if (n == 0) return G2G_OK;
    std::vector<const g2g_problem *> pp(n);
    std::vector<g2g_spparams> sp(n);
    for (int i = 0; i < n; ++i) { if (!p[i]) return G2G_ERR_ARG; pp[i] = &p[i]->prob; sp[i] = p[i]->sp; sp[i].flags = flags; }
    g2g_batch *b = 0;
    int rc = g2g_batch_prepare(ctx, n, pp.data(), &b);
    if (rc) return rc;
    rc = g2g_batch_spscore(b, sp.data(), skl, nskl, out);
    g2g_batch_free(b);
    return rc;
}
extern "C" const g2g_problem *g2g_pwdm_problem(const g2g_pwdm *p) { return p ? &p->prob : 0; }

// <-> stdskl(SKL**), reference src/gaps.cc:139-174: turn the raw traceback records into the skeleton -- corners in
// ascending order, every change between a diagonal run and a gap made explicit.  Output: the reference's skl[1..n]
// (its skl[0] header and EOS sentinel are left to the caller's wrapper).
// Own formulation in two passes: (1) order the records by (m, n) -- a total order, so any sort gives the reference's
// sequence -- and keep the monotone chain (repeats and records that step back in n are passed over); (2) walk the chain
// segment by segment.  A segment is a pure diagonal, a pure gap, or a diagonal followed by a gap; its start point is a
// corner when the heading changes there (a run of horizontal gaps keeps every start point: the reference tests `!dm`),
// and a mixed segment adds the point where its diagonal part ends.
extern "C" g2g_skl *g2g_stdskl(const g2g_skl *in, int num, int *nout)
{
    if (!nout || num < 0 || (num && !in)) return NULL;
    std::vector<g2g_skl> chain(in, in + num);
    if (num >= 2) {
        std::sort(chain.begin(), chain.end(), [](const g2g_skl &x, const g2g_skl &y) { return x.m != y.m ? x.m < y.m : x.n < y.n; });
        size_t kept = 1;
        for (int i = 1; i < num; ++i) {
            const g2g_skl &last = chain[kept - 1], &p = chain[i];
            if (p.m < last.m || p.n < last.n || (p.m == last.m && p.n == last.n)) continue;
            chain[kept++] = p;
        }
        chain.resize(kept);
    }
    g2g_skl *out = (g2g_skl *) malloc(sizeof(g2g_skl) * (2 * chain.size() + 1));
    int w = 0;
    enum { NONE = 2 };                            // heading of a segment: -1 more rows than columns, 0 diagonal, +1 more columns
    int heading = NONE;
    for (size_t k = 0; k + 1 < chain.size(); ++k) {
        const g2g_skl from = chain[k], to = chain[k + 1];
        const int rows = to.m - from.m, cols = to.n - from.n;
        const int diag = std::min(rows, cols);
        const int turn = cols > rows ? 1 : cols < rows ? -1 : 0;
        if (diag && turn) {                       // diagonal run, then a gap
            if (heading != 0) out[w++] = from;    // (coming out of a diagonal the start point is not a corner)
            out[w].m = from.m + diag; out[w].n = from.n + diag; ++w;
        } else if (turn != heading || rows == 0) out[w++] = from;
        heading = turn;
    }
    if (!chain.empty()) out[w++] = chain.back();
    *nout = w;
    return out;
}

// <-> align2(), src/maln2.cc:1875-1973 for the Fwd2c modes: forward + traceback on the GPU, stdskl,
// end check; on a mismatch the band is reopened to sh = -100 and the DP repeated (:1946-1952).
extern "C" int g2g_align2_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *pw, double *scr,
                                g2g_skl **skl, int *nskl, int *status)
{
    if (!ctx || n < 0 || (n && (!pw || !scr || !skl || !nskl))) return G2G_ERR_ARG;
    std::vector<int> todo(n), st(n, G2G_OK);
    for (int i = 0; i < n; ++i) { todo[i] = i; skl[i] = 0; nskl[i] = 0; scr[i] = 0; }
    for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
        std::vector<const g2g_problem *> pp;
        for (int i : todo) pp.push_back(&pw[i]->prob);
        std::vector<g2g_result> rr(pp.size());
        int rc = g2g_forward_batch(ctx, (int) pp.size(), pp.data(), rr.data());
        if (rc) {                                             // (a failed call hands out nothing: pass 0's skeletons go too)
            for (int i = 0; i < n; ++i) { g2g_free(skl[i]); skl[i] = 0; nskl[i] = 0; scr[i] = 0; }
            return rc;
        }
        std::vector<int> again;
        for (size_t k = 0; k < todo.size(); ++k) {
            const int i = todo[k];
            g2g_pwdm *P = pw[i];
            if (rr[k].status) { st[i] = rr[k].status; continue; }
            int ns = 0;
            g2g_skl *s = g2g_stdskl(rr[k].trace, rr[k].ntrace, &ns);
            g2g_free(rr[k].trace);
            const g2g_group &a = *P->a, &b = *P->b;
            if (ns < 1 || s[0].m != a.left || s[ns - 1].m != a.right || s[0].n != b.left || s[ns - 1].n != b.right) {
                g2g_free(s);
                if (pass == 0) {                              // pwdm->alnprm.sh = -100; goto retry
                    P->prm.sh = -100;
                    stripe(a, b, -100, &P->prob.lw, &P->prob.up);
                    again.push_back(i);
                } else st[i] = G2G_ERR_ENDS;
                continue;
            }
            scr[i] = rr[k].score; skl[i] = s; nskl[i] = ns;
        }
        todo.swap(again);
    }
    int worst = G2G_OK;
    for (int i = 0; i < n; ++i) { if (status) status[i] = st[i]; if (st[i] && !worst) worst = st[i]; }
    return status ? G2G_OK : worst;
}

// One window of the refinement per call: align2() of every pair plus calcSpScore of its CURRENT alignment (`cur`) and of the new
// one (<-> what Prrn::onecycle needs of a division, src/prrn5.cc:522-535).  Same results as g2g_align2_batch followed by
// g2g_spscore_batch_flags on 2 n (PwdM, skeleton) pairs -- and that is what runs when a DP needs the sh = -100 retry, fails, or the
// window does not fit one batch -- but here the problems are packed and uploaded once and both sets of walks share one launch.
extern "C" int g2g_batch_spscore_sets(g2g_batch *b, int nsets, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, g2g_fstat *out);
extern "C" int g2g_batch_spscore_begin(g2g_batch *b, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, void **handle);
extern "C" int g2g_batch_spscore_end(g2g_batch *b, void *handle, g2g_fstat *out);
extern "C" int g2g_align2_score_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *pw, const g2g_skl *const *cur, const int *ncur, int flags,
                                      double *scr, g2g_skl **skl, int *nskl, int *status, g2g_fstat *fs_cur, g2g_fstat *fs_new)
{
    if (!ctx || n < 0 || (n && (!pw || !cur || !ncur || !scr || !skl || !nskl || !status || !fs_cur || !fs_new))) return G2G_ERR_ARG;
    if (n == 0) return G2G_OK;
    for (int i = 0; i < n; ++i) { if (!pw[i]) return G2G_ERR_ARG; skl[i] = 0; nskl[i] = 0; scr[i] = 0; status[i] = G2G_OK; }
    bool plain = n <= 4096 && !g2g_get_option(ctx, "NO_SCORE_BATCH");      // (the option: a test seat for the general route)
    g2g_batch *b = 0;
    if (plain) {
        std::vector<const g2g_problem *> pp(n);
        for (int i = 0; i < n; ++i) pp[i] = &pw[i]->prob;
        std::vector<g2g_result> rr(n);
        for (auto &r : rr) { r.trace = 0; r.ntrace = 0; }
        int rc = g2g_batch_prepare(ctx, n, pp.data(), &b);
        // the walks over the CURRENT alignments need nothing the DPs make: they start now, on a stream of their own, beside the DP kernels
        std::vector<g2g_spparams> spc((size_t) n);
        for (int i = 0; i < n; ++i) { spc[i] = pw[i]->sp; spc[i].flags = flags; }
        void *walk = 0;
        if (!rc) rc = g2g_batch_spscore_begin(b, spc.data(), cur, ncur, &walk);
        if (!rc) rc = g2g_batch_run(b);
        if (!rc) rc = g2g_batch_fetch(b, rr.data());
        if (rc) plain = false;
        for (int i = 0; i < n && plain; ++i) {
            if (rr[i].status) { plain = false; break; }
            int ns = 0;
            g2g_skl *s = g2g_stdskl(rr[i].trace, rr[i].ntrace, &ns);
            const g2g_group &a = *pw[i]->a, &bb = *pw[i]->b;
            if (!s || ns < 1 || s[0].m != a.left || s[ns - 1].m != a.right || s[0].n != bb.left || s[ns - 1].n != bb.right) { g2g_free(s); plain = false; break; }
            scr[i] = rr[i].score; skl[i] = s; nskl[i] = ns;
        }
        for (auto &r : rr) g2g_free(r.trace);
        if (plain) {
            std::vector<g2g_spparams> sp((size_t) 2 * n);
            std::vector<const g2g_skl *> sk((size_t) 2 * n);
            std::vector<int> ns((size_t) 2 * n);
            std::vector<g2g_fstat> fs((size_t) 2 * n);
            for (int i = 0; i < n; ++i) {
                sp[i] = pw[i]->sp; sp[i].flags = flags; sp[n + i] = sp[i];
                sk[i] = cur[i]; ns[i] = ncur[i]; sk[n + i] = skl[i]; ns[n + i] = nskl[i];
            }
            if (walk) {
                rc = g2g_batch_spscore_sets(b, 1, sp.data() + n, sk.data() + n, ns.data() + n, fs.data() + n);
                const int rc2 = g2g_batch_spscore_end(b, walk, fs.data());
                walk = 0;
                if (!rc) rc = rc2;
            } else rc = g2g_batch_spscore_sets(b, 2, sp.data(), sk.data(), ns.data(), fs.data());
            g2g_batch_free(b);
            if (rc) { for (int i = 0; i < n; ++i) { g2g_free(skl[i]); skl[i] = 0; nskl[i] = 0; scr[i] = 0; } return rc; }
            for (int i = 0; i < n; ++i) { fs_cur[i] = fs[i]; fs_new[i] = fs[n + i]; }
            return G2G_OK;
        }
        if (walk) { (void) g2g_batch_spscore_end(b, walk, 0); walk = 0; }
        if (b) g2g_batch_free(b);
        for (int i = 0; i < n; ++i) { g2g_free(skl[i]); skl[i] = 0; nskl[i] = 0; scr[i] = 0; }
    }
    // the general route
    int rc = g2g_align2_batch(ctx, n, pw, scr, skl, nskl, status);
    if (rc) return rc;
    std::vector<g2g_pwdm *> pw2(pw, pw + n); pw2.insert(pw2.end(), pw, pw + n);
    std::vector<const g2g_skl *> sk((size_t) 2 * n);
    std::vector<int> ns((size_t) 2 * n);
    std::vector<g2g_fstat> fs((size_t) 2 * n);
    for (int i = 0; i < n; ++i) { sk[i] = cur[i]; ns[i] = ncur[i]; sk[n + i] = skl[i]; ns[n + i] = nskl[i]; }
    rc = g2g_spscore_batch_flags(ctx, 2 * n, pw2.data(), sk.data(), ns.data(), flags, fs.data());
    if (rc) { for (int i = 0; i < n; ++i) { g2g_free(skl[i]); skl[i] = 0; nskl[i] = 0; } return rc; }
    for (int i = 0; i < n; ++i) { fs_cur[i] = fs[i]; fs_new[i] = fs[n + i]; }
    return G2G_OK;
}

extern "C" int g2g_homscore(g2g_ctx *ctx, g2g_pwdm *p, double *scr, int64_t rr[2])
{
    if (!ctx || !p || !scr) return G2G_ERR_ARG;
    const g2g_group &a = *p->a, &b = *p->b;
    if (a.left == a.right || b.left == b.right) {            // maln2.cc:1843-1849
        if (rr) { rr[0] = b.left - a.left; rr[1] = b.right - a.right; }
        *scr = 0;
        return G2G_OK;
    }
    const g2g_problem *pp = &p->prob;
    g2g_result r;
    int rc = g2g_forward_batch(ctx, 1, &pp, &r);
    if (rc) return rc;
    if (r.status) return r.status;
    g2g_free(r.trace);
    *scr = r.score;
    if (rr) { rr[0] = r.rr[0]; rr[1] = r.rr[1]; }
    return G2G_OK;
}

extern "C" int g2g_align2(g2g_ctx *ctx, g2g_pwdm *p, double *scr, g2g_skl **skl, int *nskl)
{
    if (!p) return G2G_ERR_ARG;
    g2g_pwdm *pp[1] = {p};
    return g2g_align2_batch(ctx, 1, pp, scr, skl, nskl, 0);
}
