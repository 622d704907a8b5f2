// g2g_kernels_v4.hip -- forward kernel of the _pf engine (both groups carry gap profiles; Noll 2/3):
// EIGHT LANES PER CELL inside the single-wave tile framework of v3.
//
// A _pf cell needs six (eight) two-list merges that are independent of each other (Fwd2c<_pf>::gapopen,
// reference src/fwd2c.cc:203-212 -> newgap, src/gfreq.cc:507-521).  One lane per cell (v3) does them one after
// the other and needs all three static lists of its row in registers -- over 400 VGPRs, one wave per SIMD, and
// a single wave cannot issue more than one instruction every four cycles.  Here a team of 8 lanes owns a cell:
//   * each lane has ONE role = one merge, and keeps only the static list of a that its merge walks (s, t or r
//     view of the team's row) in registers -- 48 VGPRs -- so that several waves fit on a SIMD;
//   * all eight merges run as ONE instruction stream: the outer loop walks the lane's register list with a
//     uniform index, the inner pointer walks the column's list (LDS ring, first four entries cached in
//     registers); the reference's two loop orders ("a inside" / "b inside") differ only in the comparison and
//     in when the product is added, which a per-lane flag selects;
//   * a strip is 8 rows: tiles are short pipelines (C + 7 steps for C columns), the column ring holds 10
//     columns, LDS per wave is ~12 KB;
//   * record scalars stay in registers (every lane of a team replays the cheap decision logic) and go to the
//     team below by an 8-lane shuffle; record lists live in LDS rings as in v3; no workgroup barrier anywhere.
// Arithmetic, its order and the traceback bits are those of the other generations (bit-exact with the reference).
#include <hip/hip_runtime.h>

#ifndef G2G_V4_UNROLL
#define G2G_V4_UNROLL 1
#endif
struct V4Lds { int rows, black, stsc, boff, bring_g, bring_f, svals, sink, total; };
#define V4_R 8
#define V4_RC 10                              // columns in the LDS ring (R + 2)
#define V4_MLB 16                             // longest static list of b the loader handles (entries incl. terminator)

__device__ __forceinline__ RS rs_up8(const RS &x)
{
    RS r; r.val = __shfl_up(x.val, 8); r.dir = __shfl_up(x.dir, 8); r.glb = __shfl_up(x.glb, 8); return r;
}

// one cell by one team; xl: the record's dla list (dlb follows ca4 dwords later); d*: destination lists
template <bool NOLL3, int N>
__device__ __forceinline__ void v4_cell(const DevProb &P, const int ca4, const int role, const int (&ag)[N], const double (&af)[N],
    const LList bsl, const LList btl, const LList brl, lu32 *sink,
    const RS &hd, const lu32 *hdl, const RS &hu, const lu32 *hul, const RS &gu, const lu32 *gul,
    const RS &g2u, const lu32 *g2ul, const RS &hl, const lu32 *hll, const RS &fl, const lu32 *fll,
    const RS &f2l, const lu32 *f2ll,
    lu32 *dh, lu32 *dg, lu32 *dg2, lu32 *df, lu32 *df2,
    const bool do_vert, const bool do_hori, const double dab, const double pua, const double pub,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb)
{
    // ---- phase A: one merge per lane ----------------------------------------------------------------------
    //   role 0: newgap(a.s, dla | b.t, dlb) of hd      role 1: newgap(b.s, dlb | a.t, dla) of hd
    //   role 2: newgap(a.s | b.r) of gu   3: of hu   7: of g2u          (vertical, "a inside")
    //   role 4: newgap(b.s | a.r) of fl   5: of hl   6: of f2l          (horizontal, "b inside")
    const bool isY = role == 0 || role == 2 || role == 3 || role == 7;
    const lu32 *rcl = (role < 2) ? hdl : (role == 2) ? gul : (role == 3) ? hul : (role == 4) ? fll
                    : (role == 5) ? hll : (role == 6) ? f2ll : g2ul;
    const LList bsel = (role == 0) ? btl : isY ? brl : bsl;
    const bool on = role < 2 || ((role == 2 || role == 3) && do_vert) || ((role == 4 || role == 5) && do_hori)
                    || (NOLL3 && ((role == 7 && do_vert) || (role == 6 && do_hori)));
    const DHead ha = dh_load<true>(rcl), hb = dh_load<true>(rcl + ca4);
    const BHead bl = bh_load(bsel);
    double g = 0;
    {
        // "a inside" consumes column entries while w <= v, "b inside" skips them while w < v: one compare w < v + yinc
        const int yinc = isY ? 1 : 0;
        int k = 0, bg = bl.g0;
        int w = bg >= 0 ? bg + dh_nins(bg, hb) : 0;
        bool live = on;
#pragma unroll G2G_V4_UNROLL
        for (int i = 0; i < N; ++i) {
            if (wave_none(live)) break;
            const int a_g = ag[i];
            const double a_f = af[i];
            live = live && a_g >= 0;
            const int v = a_g + dh_nins(a_g, ha) + yinc;
            bool adv = live && bg >= 0 && w < v;
            while (__ballot(adv)) {                          // (predicated body: no divergent branch inside)
                const double bf = bh_freq(bl, k);
                g = (adv && isY) ? g + a_f * bf : g;
                k += adv ? 1 : 0;
                bg = bh_glen(bl, k);
                w = bg >= 0 ? bg + dh_nins(bg, hb) : 0;
                adv = adv && bg >= 0 && w < v;
            }
            live = live && bg >= 0;
            const double bf = bh_freq(bl, k);
            g = (live && !isY) ? g + bf * a_f : g;
        }
    }
    const double r = g * P.basic_gop;
    Costs c;
    c.d0 = __shfl(r, 0, 8); c.d1 = __shfl(r, 1, 8);
    c.gnpv = __shfl(r, 2, 8); c.gopv = __shfl(r, 3, 8);
    c.gnph = __shfl(r, 4, 8); c.goph = __shfl(r, 5, 8);
    c.gnph2 = NOLL3 ? __shfl(r, 6, 8) : 0; c.gnpv2 = NOLL3 ? __shfl(r, 7, 8) : 0;
    // ---- decisions, replayed by every lane ------------------------------------------------------------------
    const Dec d = v3_decide<2, NOLL3>(P, c, hd, hu, gu, g2u, hl, fl, f2l, do_vert, do_hori, dab, pua, pub);
    const int win = d.win;
    // ---- phase B: list updates (update(), fwd2c.cc:216-231) --------------------------------------------------
    //   role 1 (holds a.t): newdelta of G.dla (G2.dla) and of a diagonal H.dla
    //   role 0 (holds b.t's head): newdelta of F.dlb (F2.dlb) and of a diagonal H.dlb
    //   role 3: incdelta G.dlb   5: incdelta F.dla   2: incdelta G2.dlb   6: incdelta F2.dla
    team_sync();
    const lu32 *gsl = d.g_from_h ? hul : gul, *gs2l = d.g2_from_h ? hul : g2ul;
    const lu32 *fsl = d.f_from_h ? hll : fll, *fs2l = d.f2_from_h ? hll : f2ll;
    lu32 *const nul = (lu32 *) 0;
    if (role < 2) {
        const int o = role == 0 ? ca4 : 0;                 // role 0 works on the dlb lists
        const bool on1 = role == 1 ? do_vert : do_hori;
        const DHead h1 = dh_load<true>((role == 1 ? gsl : fsl) + o);
        const DHead h3 = dh_load<true>((role == 1 ? gs2l : fs2l) + o);
        const DHead h2 = role == 1 ? ha : hb;              // hd's list: loaded for the merge already
        lu32 *const d1 = (role == 1 ? dg : df) + o, *const d3 = (role == 1 ? dg2 : df2) + o, *const d2 = dh + o;
        lu32 *const d1b = (win == (role == 1 ? 1 : 3)) ? d2 : nul, *const d3b = (win == (role == 1 ? 2 : 4)) ? d2 : nul;
        ND n1 = {0, 0, 0, on1}, n2 = {0, 0, 0, win == 0}, n3 = {0, 0, 0, on1 && NOLL3};
#pragma unroll G2G_V4_UNROLL
        for (int i = 0; i < N; ++i) {
            if (wave_none(n1.on || n2.on || (NOLL3 && n3.on))) break;
            const int e_g = role == 1 ? ag[i] : bh_glen(bl, i);
            nd_step(n1, h1, e_g, d1, d1b, sink);
            nd_step(n2, h2, e_g, d2, nul, sink);
            if (NOLL3) nd_step(n3, h3, e_g, d3, d3b, sink);
        }
        nd_fin(n1, on1, d1, d1b, sink);
        nd_fin(n2, win == 0, d2, nul, sink);
        if (NOLL3) nd_fin(n3, on1, d3, d3b, sink);
    } else if (role == 3 || role == 5 || (NOLL3 && (role == 2 || role == 6))) {
        const bool vert = role == 3 || role == 2;
        const lu32 *src = (role == 3) ? gsl + ca4 : (role == 5) ? fsl : (role == 2) ? gs2l + ca4 : fs2l;
        lu32 *const d1 = (role == 3) ? dg + ca4 : (role == 5) ? df : (role == 2) ? dg2 + ca4 : df2;
        const int wsel = (role == 3) ? 1 : (role == 5) ? 3 : (role == 2) ? 2 : 4;
        lu32 *const d2 = (win == wsel) ? (vert ? dh + ca4 : dh) : nul;
        incdelta_h(vert ? do_vert : do_hori, dh_load<true>(src), d1, d2, sink);
    }
    v3_outputs<2, NOLL3>(d, 0, 0, do_vert, do_hori, oH, oG, oG2, oF, oF2, trb);
    team_sync();
}

template <bool NOLL3, int N>
__device__ void v4_tile(const DevProb &Pmem, lchar *lds, const V4Lds LO, const int ti, const int tj, const int nsteps, const int C)
{
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int lane = threadIdx.x, role = lane & 7, team = lane >> 3;       // blockDim.x == 64
    const int capa = P.capa, capb = P.capb;
    const int ca4 = (capa + 3) & ~3, cb4 = (capb + 3) & ~3, lsz = ca4 + cb4;
    const int nslot = NOLL3 ? 9 : 6;
    const int pitch = v3_pitch(nslot, lsz);
    const int ndw = ((16 + 4 * (capa + capb) + 15) & ~15) / 4;
    lu32 *const rows = (lu32 *) (lds + LO.rows);           // row 0: staging (the strip above), row t+1: team t
    lu32 *const blk = (lu32 *) (lds + LO.black);
    lu32 *const stsc = (lu32 *) (lds + LO.stsc);           // staging scalars: H ring 0-2, G 3-4, G2 5-6
    li32 *const boff = (li32 *) (lds + LO.boff);           // b.off of the block's columns, 3 x (C + 2)
    li32 *const brg = (li32 *) (lds + LO.bring_g);         // column ring: [slot][view][V4_MLB]
    lf64 *const brf = (lf64 *) (lds + LO.bring_f);
    lu32 *const sink = (lu32 *) (lds + LO.sink) + lane;
#define V4_L(r, slot) (rows + (r) * pitch + (slot) * lsz)
    const size_t rbuf = (size_t) P.v2_rowstride * ndw;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const unsigned *rowHp = (const unsigned *) P.v2_rowH + bprev * rbuf, *rowGp = (const unsigned *) P.v2_rowG + bprev * rbuf;
    const unsigned *rowG2p = NOLL3 ? (const unsigned *) P.v2_rowG2 + bprev * rbuf : 0;
    unsigned *rowHc = (unsigned *) P.v2_rowH + bcur * rbuf, *rowGc = (unsigned *) P.v2_rowG + bcur * rbuf;
    unsigned *rowG2c = NOLL3 ? (unsigned *) P.v2_rowG2 + bcur * rbuf : 0;
    const unsigned *colH = (const unsigned *) P.v2_colH;
    unsigned *cbH = (unsigned *) P.v2_cbH, *cbF = (unsigned *) P.v2_cbF, *cbF2 = (unsigned *) P.v2_cbF2;
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const int m0 = a.left + ti * V4_R, m = m0 + team;
    const int mend = (m0 + V4_R < a.right) ? m0 + V4_R : a.right;
    const int tlast = mend - 1 - m0;                       // team of the strip's last row
    const int c0 = b.left + tj * C;
    int c1 = c0 + C; if (c1 > b.right) c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo > c0 ? nlo : c0, hi = nhi < c1 ? nhi : c1;      // ... clipped to this block
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left; if (cbase < c0) cbase = c0;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right; if (hi0 > c1) hi0 = c1;   // team 0's hi
    const bool vert0 = m0 > a.left;                        // the strip has a row above

    // ---- LDS init: black list, every ring slot black (reset(f1), reset(f2), fwd2c.h:385-386)
    if (lane < 4) blk[lane] = (lane == 1) ? (DL_END << 16) : 0;
    if (lane < 4) blk[ca4 + lane] = (lane == 1) ? (DL_END << 16) : 0;
    for (int sl = role; sl < nslot; sl += 8) {
        lu32 *p = V4_L(team + 1, sl);
        p[0] = 0; p[1] = DL_END << 16; p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    if (lane < nslot) {
        lu32 *p = V4_L(0, lane);
        p[0] = 0; p[1] = DL_END << 16; p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    for (int j = lane; j < 3 * (C + 2); j += 64) {         // list offsets of the block's columns (+ end)
        const int v = j / (C + 2), c = j - v * (C + 2);
        const int pos = c0 + c;
        const int *offv = v == 0 ? b.off[0] : v == 1 ? b.off[1] : b.off[2];   // (no runtime index into the descriptor copy)
        boff[j] = (pos <= b.len) ? offv[pos + 1] : offv[b.len + 1];
    }
    // ---- this lane's static list of a (by role): registers ---------------------------------------------------
    int a_g[N];
    double a_f[N];
    {
        const int view = (role == 1) ? 1 : (role >= 4 && role <= 6) ? 2 : 0;
        DevSide av = a;                                    // the role's view in slot 0 (no runtime index into the copy)
        av.off[0] = view == 0 ? a.off[0] : view == 1 ? a.off[1] : a.off[2];
        av.glen[0] = view == 0 ? a.glen[0] : view == 1 ? a.glen[1] : a.glen[2];
        av.freq[0] = view == 0 ? a.freq[0] : view == 1 ? a.freq[1] : a.freq[2];
        rl_load(a_g, a_f, av, 0, m, row_ok);
    }
    // column ring loader: lane l moves entry (l & 15) of view (l >> 4) of the column team 0 reaches next
    const int ld_v = lane >> 4, ld_k = lane & 15;
    const int *const ld_glen = ld_v == 0 ? b.glen[0] : ld_v == 1 ? b.glen[1] : b.glen[2];
    const double *const ld_freq = ld_v == 0 ? b.freq[0] : ld_v == 1 ? b.freq[1] : b.freq[2];
    auto ring_load = [&](int col, int &rg_, double &rf_) {
        if (ld_v < 3) {
            const int o = boff[ld_v * (C + 2) + (col - c0)], e = boff[ld_v * (C + 2) + (col - c0) + 1];
            if (ld_k < e - o) { rg_ = ld_glen[o + ld_k]; rf_ = ld_freq[o + ld_k]; }
        }
    };
    auto ring_store = [&](int slot, int rg_, double rf_) {
        if (ld_v < 3) { brg[(slot * 3 + ld_v) * V4_MLB + ld_k] = rg_; brf[(slot * 3 + ld_v) * V4_MLB + ld_k] = rf_; }
    };
    // ---- the records this row starts from (all lanes of the team keep the scalars) ------------------------------
    RS oH = rs_black(), oG = rs_black(), oG2 = rs_black(), oF = rs_black(), oF2 = rs_black();
    {
        const bool cont = row_ok && c0 - 1 >= nlo && c0 - 1 < nhi;         // continues from the block on the left
        const unsigned *src = 0;
        if (cont) src = cbH + (size_t) (m - a.left) * ndw;                 // corner (m+1, c0)
        else if (row_ok && c0 == b.left && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left)
            src = colH + (size_t) (m + 1 - a.left) * ndw;                  // left boundary corner (m+1, b.left)
        lu32 *p = V4_L(team + 1, SLOT_H(c0));
        if (src) {
            oH.val = *(const double *) src; oH.dir = (int) src[2]; oH.glb = (int) src[3];
            for (int k = role; k < capa; k += 8) p[k] = src[4 + k];
            for (int k = role; k < capb; k += 8) p[ca4 + k] = src[4 + capa + k];
        }
        if (cont && lo < hi) {
            const unsigned *s2 = cbF + (size_t) (m - a.left) * ndw;
            p = V4_L(team + 1, SLOT_F);
            oF.val = *(const double *) s2; oF.dir = (int) s2[2]; oF.glb = (int) s2[3];
            for (int k = role; k < capa; k += 8) p[k] = s2[4 + k];
            for (int k = role; k < capb; k += 8) p[ca4 + k] = s2[4 + capa + k];
            if (NOLL3) {
                s2 = cbF2 + (size_t) (m - a.left) * ndw;
                p = V4_L(team + 1, SLOT_F2);
                oF2.val = *(const double *) s2; oF2.dir = (int) s2[2]; oF2.glb = (int) s2[3];
                for (int k = role; k < capa; k += 8) p[k] = s2[4 + k];
                for (int k = role; k < capb; k += 8) p[ca4 + k] = s2[4 + capa + k];
            }
        }
    }
    // ---- staging row: records of the strip above for team 0's columns, one dword per lane -------------------------
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < ndw) {
            const unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * ndw : rowHp + (size_t) col * ndw;
            rh = s[lane];
            if (wantG) { rg = rowGp[(size_t) col * ndw + lane]; if (NOLL3) rg2 = rowG2p[(size_t) col * ndw + lane]; }
        }
    };
    auto stage_put = [&](int slot, int sid, unsigned v) {
        const int j = lane - 4;
        if (lane < 4) stsc[sid * 4 + lane] = v;
        else if (j < capa) V4_L(0, slot)[j] = v;
        else if (j < capa + capb) V4_L(0, slot)[ca4 + j - capa] = v;
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < ndw) {
            stage_put(SLOT_H(col), SLOT_H(col), rh);
            if (wantG) { stage_put(SLOT_G(col), 3 + (col & 1), rg); if (NOLL3) stage_put(SLOT_G2(col), 5 + (col & 1), rg2); }
        }
    };
    if (lane < 28) stsc[lane] = 0;
    team_sync();
    {
        unsigned rh = 0, rg = 0, rg2 = 0;
        stage_load(cbase, false, rh, rg, rg2);
        stage_store(cbase, false, rh, rg, rg2);
        if (cbase + 1 <= c1) {
            stage_load(cbase + 1, vert0, rh, rg, rg2);
            stage_store(cbase + 1, vert0, rh, rg, rg2);
        }
        int g_ = 0; double f_ = 0;
        ring_load(cbase, g_, f_);                          // column of step 0 -> ring slot 0
        ring_store(0, g_, f_);
    }
    // per-row constants and one-step-ahead register pipelines (column score, b's column thickness)
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    const double *simrow = row_ok ? P.v2_sim + P.v2_rowoff[m - a.left] - nlo : 0;
    double sim_cur = 0, bc_cur = 0;
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;                                  // staging registers hold column n0 + 1
    int rg_nx = 0; double rf_nx = 0; bool ring_prev = false;   // ring registers hold column n0 + 1
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;    // the previous step's trace byte
    const int ull = __builtin_amdgcn_readfirstlane(tlast * 8);
    int lhi = m0 + tlast + P.up + 1; if (lhi > b.right) lhi = b.right; if (lhi > c1) lhi = c1;
    int llo = m0 + tlast + P.lw; if (llo < b.left) llo = b.left; if (llo < c0) llo = c0;
    auto flush_rows = [&](const int nl) {                  // nl: the last row's column in the step being flushed
        if (nl >= llo && nl < lhi) {
            const int col = nl + 1;
            const int j = lane - 4;
#pragma unroll
            for (int x = 0; x < (NOLL3 ? 3 : 2); ++x) {
                const RS &r = (x == 0) ? oH : (x == 1) ? oG : oG2;
                const int slot = (x == 0) ? SLOT_H(col) : (x == 1) ? SLOT_G(col) : SLOT_G2(col);
                const unsigned v0 = (unsigned) __builtin_amdgcn_readlane(__double2loint(r.val), ull);
                const unsigned v1 = (unsigned) __builtin_amdgcn_readlane(__double2hiint(r.val), ull);
                const unsigned v2 = (unsigned) __builtin_amdgcn_readlane(r.dir, ull), v3 = (unsigned) __builtin_amdgcn_readlane(r.glb, ull);
                unsigned v = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
                if (lane >= 4 && lane < ndw) {
                    const lu32 *p = V4_L(tlast + 1, slot);
                    v = (j < capa) ? p[j] : (j < capa + capb) ? p[ca4 + j - capa] : 0;
                }
                unsigned *dst = (x == 0) ? rowHc : (x == 1) ? rowGc : rowG2c;
                if (lane < ndw) dst[(size_t) col * ndw + lane] = v;
            }
        }
    };
    int rslot = (V4_RC - team % V4_RC) % V4_RC;            // ring slot of column cbase + s - team
    int wslot = 1;                                         // ring slot of column cbase + s + 1
    team_sync();
    for (int s = 0; s < nsteps; ++s) {
        const int n = cbase + s - team;
        const int n0 = cbase + s;                          // team 0's column
        const bool active = row_ok && n >= lo && n < hi;
        // -- top of the step: consume last step's loads, issue last step's stores
        if (st_prev) stage_store(n0 + 1, vert0, st_h, st_g, st_g2);
        if (ring_prev) ring_store(wslot == 0 ? V4_RC - 1 : wslot - 1, rg_nx, rf_nx);     // column n0 -> this step's slot of team 0
        if (p_act && role == 0) P.trace[p_tri] = (uint8_t) p_trb;
        if (wr_rows && s > 0) flush_rows(n0 - 1 - tlast);
        team_sync();
        // -- hand-over from the team above
        hd = hu;
        hu = rs_up8(oH); gu = rs_up8(oG);
        if (NOLL3) g2u = rs_up8(oG2);
        {
            const lu32 *q = stsc + SLOT_H(n0) * 4;
            RS t; t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hd = rs_sel(team == 0, t, hd);
            q = stsc + SLOT_H(n0 + 1) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hu = rs_sel(team == 0, t, hu);
            q = stsc + (3 + ((n0 + 1) & 1)) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            gu = rs_sel(team == 0, t, gu);
            if (NOLL3) {
                q = stsc + (5 + ((n0 + 1) & 1)) * 4;
                t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
                g2u = rs_sel(team == 0, t, g2u);
            }
        }
        // -- loads for the next step
        double sim_nx = 0, bc_nx = 0;
        if (active) {
            if (!have) { sim_cur = simrow[n]; bc_cur = thk_at(b, n)[0]; }
            if (n + 1 < hi) { sim_nx = simrow[n + 1]; bc_nx = thk_at(b, n + 1)[0]; }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) stage_load(n0 + 2, vert0, st_h, st_g, st_g2);
        ring_prev = n0 + 1 < c1;
        if (ring_prev) ring_load(n0 + 1, rg_nx, rf_nx);
        RS myH = oH, myG = oG, myG2 = oG2;                 // (the produced records of this step)
        if (active) {
            const bool do_hori = n > b.left;
            const li32 *cg = brg + (size_t) rslot * 3 * V4_MLB;
            const lf64 *cf = brf + (size_t) rslot * 3 * V4_MLB;
            LList bsl, btl, brl;
            bsl.glen = cg; bsl.freq = cf;
            btl.glen = cg + V4_MLB; btl.freq = cf + V4_MLB;
            brl.glen = cg + 2 * V4_MLB; brl.freq = cf + 2 * V4_MLB;
            const bool up_in = do_vert && (n - (m - 1) <= P.up);          // cell (m-1, n) exists
            const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
            const RS bk = rs_black();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const lu32 *hdl = V4_L(team, SLOT_H(n));
            const lu32 *hul = up_in ? V4_L(team, SLOT_H(n + 1)) : blk;
            const lu32 *gul = up_in ? V4_L(team, SLOT_G(n + 1)) : blk;
            const lu32 *g2ul = (NOLL3 && up_in) ? V4_L(team, SLOT_G2(n + 1)) : blk;
            const lu32 *hll = left_in ? V4_L(team + 1, SLOT_H(n)) : blk;
            const lu32 *fll = left_in ? V4_L(team + 1, SLOT_F) : blk;
            const lu32 *f2ll = (NOLL3 && left_in) ? V4_L(team + 1, SLOT_F2) : blk;
            lu32 *dh = V4_L(team + 1, SLOT_H(n + 1));
            lu32 *dg = V4_L(team + 1, SLOT_G(n + 1));
            lu32 *dg2 = V4_L(team + 1, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
            lu32 *df = V4_L(team + 1, SLOT_F);
            lu32 *df2 = V4_L(team + 1, NOLL3 ? SLOT_F2 : SLOT_F);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            int trb = 0;
            v4_cell<NOLL3, N>(P, ca4, role, a_g, a_f, bsl, btl, brl, sink, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                              dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb);
            const int d = m + n;
            int mlo, mhi;
            diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (d - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; have = (n + 1 < hi);
            // block boundary: this row's corner and F records for the block on the right
            if (n == c1 - 1 && c1 < b.right) {
                unsigned *o1 = cbH + (size_t) (m - a.left) * ndw, *o2 = cbF + (size_t) (m - a.left) * ndw;
                if (role == 0) {
                    *(double *) o1 = myH.val; o1[2] = (unsigned) myH.dir; o1[3] = (unsigned) myH.glb;
                    *(double *) o2 = oF.val; o2[2] = (unsigned) oF.dir; o2[3] = (unsigned) oF.glb;
                }
                for (int k = role; k < capa; k += 8) { o1[4 + k] = dh[k]; o2[4 + k] = df[k]; }
                for (int k = role; k < capb; k += 8) { o1[4 + capa + k] = dh[ca4 + k]; o2[4 + capa + k] = df[ca4 + k]; }
                if (NOLL3) {
                    unsigned *o3 = cbF2 + (size_t) (m - a.left) * ndw;
                    if (role == 0) { *(double *) o3 = oF2.val; o3[2] = (unsigned) oF2.dir; o3[3] = (unsigned) oF2.glb; }
                    for (int k = role; k < capa; k += 8) o3[4 + k] = df2[k];
                    for (int k = role; k < capb; k += 8) o3[4 + capa + k] = df2[ca4 + k];
                }
            }
            if (m == a.right - 1 && n == b.right - 1 && role == 0) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        if (++rslot == V4_RC) rslot = 0;
        if (++wslot == V4_RC) wslot = 0;
        team_sync();
    }
    if (p_act && role == 0) P.trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - tlast);
#undef V4_L
}


// ---- v5: FOUR lanes per cell ---------------------------------------------------------------------------
// v4 spends most of a step on work that does not depend on the number of cells (hand-over, staging, strip boundary,
// decisions) and keeps only two of eight lanes busy in the list updates.  With four lanes per cell a wave covers 16
// rows: each lane runs TWO merges in one fused loop over its register list (both walk the same static list of a),
// the four (six) newdeltas take one lane each (entries of a.t / b.t are handed round by a 4-lane shuffle), and the
// per-step overhead is shared by twice as many cells.
#define V5_R 16
#define V5_RC 18
__device__ __forceinline__ RS rs_up4(const RS &x)
{
    RS r; r.val = __shfl_up(x.val, 4); r.dir = __shfl_up(x.dir, 4); r.glb = __shfl_up(x.glb, 4); return r;
}
struct MSt { int k, bg, w; bool live; double g; };
__device__ __forceinline__ void mst_init(MSt &s, const bool on, const BHead &bl, const DHead &hb)
{
    s.k = 0; s.bg = bl.g0; s.w = s.bg >= 0 ? s.bg + dh_nins(s.bg, hb) : 0; s.live = on; s.g = 0;
}
// one entry of a's list against the merge state (see v4_cell): yinc = 1 -> "a inside" (consume while w <= v),
// yinc = 0 -> "b inside" (skip while w < v, then add the product)
__device__ __forceinline__ void mst_step(MSt &s, const int a_g, const double a_f, const DHead &ha, const DHead &hb, const BHead &bl,
                                         const bool isY, const int yinc)
{
    s.live = s.live && a_g >= 0;
    const int v = a_g + dh_nins(a_g, ha) + yinc;
    bool adv = s.live && s.bg >= 0 && s.w < v;
    while (__ballot(adv)) {
        const double bf = bh_freq(bl, s.k);
        s.g = (adv && isY) ? s.g + a_f * bf : s.g;
        s.k += adv ? 1 : 0;
        s.bg = bh_glen(bl, s.k);
        s.w = s.bg >= 0 ? s.bg + dh_nins(s.bg, hb) : 0;
        adv = adv && s.bg >= 0 && s.w < v;
    }
    s.live = s.live && s.bg >= 0;
    const double bf = bh_freq(bl, s.k);
    s.g = (s.live && !isY) ? s.g + bf * a_f : s.g;
}

template <bool NOLL3, int N>
__device__ __forceinline__ void v5_cell(const DevProb &P, const int ca4, const int q, const int (&ag)[N], const double (&af)[N],
    const int (&ag2)[N], const double (&af2)[N],
    const LList bsl, const LList btl, const LList brl, lu32 *sink,
    const RS &hd, const lu32 *hdl, const RS &hu, const lu32 *hul, const RS &gu, const lu32 *gul,
    const RS &g2u, const lu32 *g2ul, const RS &hl, const lu32 *hll, const RS &fl, const lu32 *fll,
    const RS &f2l, const lu32 *f2ll,
    lu32 *dh, lu32 *dg, lu32 *dg2, lu32 *df, lu32 *df2,
    const bool do_vert, const bool do_hori, const double dab, const double pua, const double pub,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb)
{
    // ---- phase A: two merges per lane (slots A, B) -----------------------------------------------------------
    //   lane 0 (a.s): A newgap(a.s | b.t) of hd,  B newgap(a.s | b.r) of gu        ("a inside")
    //   lane 1 (a.s): A newgap(a.s | b.r) of hu,  B of g2u (long-gap penalty)
    //   lane 2 (a.r): A newgap(b.s | a.r) of fl,  B of hl                          ("b inside")
    //   lane 3 (a.t): A newgap(b.s | a.t) of hd,  B newgap(b.s | a.r) of f2l (long-gap penalty, second list)
    const bool isY = q < 2;
    const int yinc = isY ? 1 : 0;
    const lu32 *rcA = (q == 0 || q == 3) ? hdl : (q == 1) ? hul : fll;
    const lu32 *rcB = (q == 0) ? gul : (q == 1) ? g2ul : (q == 2) ? hll : f2ll;
    const LList blA_l = (q == 0) ? btl : (q == 1) ? brl : bsl;
    const LList blB_l = isY ? brl : bsl;
    const bool onA = (q == 0 || q == 3) || (q == 1 && do_vert) || (q == 2 && do_hori);
    const bool onB = (q == 0 && do_vert) || (q == 2 && do_hori) || (NOLL3 && ((q == 1 && do_vert) || (q == 3 && do_hori)));
    const DHead haA = dh_load<true>(rcA), hbA = dh_load<true>(rcA + ca4);
    const DHead haB = dh_load<true>(rcB), hbB = dh_load<true>(rcB + ca4);
    const BHead blA = bh_load(blA_l), blB = bh_load(blB_l);
    MSt sA, sB;
    mst_init(sA, onA, blA, hbA);
    mst_init(sB, onB, blB, hbB);
#pragma unroll G2G_V4_UNROLL
    for (int i = 0; i < N; ++i) {
        if (wave_none(sA.live || sB.live)) break;
        const int g1 = ag[i];
        const double f1 = af[i];
        const int g2 = (NOLL3 && q == 3) ? ag2[i] : g1;
        const double f2 = (NOLL3 && q == 3) ? af2[i] : f1;
        mst_step(sA, g1, f1, haA, hbA, blA, isY, yinc);
        mst_step(sB, g2, f2, haB, hbB, blB, isY, yinc);
    }
    const double rA = sA.g * P.basic_gop, rB = sB.g * P.basic_gop;
    Costs c;
    c.d0 = __shfl(rA, 0, 4); c.gnpv = __shfl(rB, 0, 4);
    c.gopv = __shfl(rA, 1, 4); c.gnpv2 = NOLL3 ? __shfl(rB, 1, 4) : 0;
    c.gnph = __shfl(rA, 2, 4); c.goph = __shfl(rB, 2, 4);
    c.d1 = __shfl(rA, 3, 4); c.gnph2 = NOLL3 ? __shfl(rB, 3, 4) : 0;
    // ---- decisions, replayed by every lane ------------------------------------------------------------------
    const Dec d = v3_decide<2, NOLL3>(P, c, hd, hu, gu, g2u, hl, fl, f2l, do_vert, do_hori, dab, pua, pub);
    const int win = d.win;
    // ---- phase B: list updates (update(), fwd2c.cc:216-231), one newdelta per lane ------------------------------
    //   lane 3: G.dla <- gs.dla   lane 2: H.dla <- hd.dla (diagonal H), then G2.dla     over a.t (lane 3's registers)
    //   lane 0: F.dlb <- fs.dlb   lane 1: H.dlb <- hd.dlb (diagonal H), then F2.dlb     over b.t (lane 0's slot A)
    //   incdelta: lane 0 G.dlb, lane 1 F.dla, lane 2 G2.dlb, lane 3 F2.dla
    team_sync();
    const lu32 *gsl = d.g_from_h ? hul : gul, *gs2l = d.g2_from_h ? hul : g2ul;
    const lu32 *fsl = d.f_from_h ? hll : fll, *fs2l = d.f2_from_h ? hll : f2ll;
    lu32 *const nul = (lu32 *) 0;
    {
        const lu32 *s1 = (q == 3) ? gsl : (q == 2) ? hdl : (q == 0) ? fsl + ca4 : hdl + ca4;
        const lu32 *s2 = (q == 2) ? gs2l : fs2l + ca4;                              // (lanes 2 and 1 only)
        const DHead h1 = dh_load<true>(s1), h2 = dh_load<true>(NOLL3 ? s2 : s1);
        lu32 *const d1 = (q == 3) ? dg : (q == 2) ? dh : (q == 0) ? df + ca4 : dh + ca4;
        lu32 *const d1b = (q == 3) ? (win == 1 ? dh : nul) : (q == 0) ? (win == 3 ? dh + ca4 : nul) : nul;
        lu32 *const d2 = (q == 2) ? dg2 : df2 + ca4;
        lu32 *const d2b = (q == 2) ? (win == 2 ? dh : nul) : (win == 4 ? dh + ca4 : nul);
        const bool on1 = (q == 3) ? do_vert : (q == 0) ? do_hori : (win == 0);
        const bool on2 = NOLL3 && ((q == 2 && do_vert) || (q == 1 && do_hori));
        ND n1 = {0, 0, 0, on1}, n2 = {0, 0, 0, on2};
#pragma unroll G2G_V4_UNROLL
        for (int i = 0; i < N; ++i) {
            if (wave_none(n1.on || (NOLL3 && n2.on))) break;
            const int mine = (q == 3) ? ag[i] : bh_glen(blA, i);                   // lane 3: a.t, lane 0: b.t
            const int e_g = __shfl(mine, q >= 2 ? 3 : 0, 4);
            nd_step(n1, h1, e_g, d1, d1b, sink);
            if (NOLL3) nd_step(n2, h2, e_g, d2, d2b, sink);
        }
        nd_fin(n1, on1, d1, d1b, sink);
        if (NOLL3) nd_fin(n2, on2, d2, d2b, sink);
    }
    {
        const lu32 *src = (q == 0) ? gsl + ca4 : (q == 1) ? fsl : (q == 2) ? gs2l + ca4 : fs2l;
        lu32 *const d1 = (q == 0) ? dg + ca4 : (q == 1) ? df : (q == 2) ? dg2 + ca4 : df2;
        const int wsel = (q == 0) ? 1 : (q == 1) ? 3 : (q == 2) ? 2 : 4;
        const bool vert = q == 0 || q == 2;
        lu32 *const d2 = (win == wsel) ? (vert ? dh + ca4 : dh) : nul;
        const bool on = (q < 2 || NOLL3) && (vert ? do_vert : do_hori);
        incdelta_h(on, dh_load<true>(src), d1, d2, sink);
    }
    v3_outputs<2, NOLL3>(d, 0, 0, do_vert, do_hori, oH, oG, oG2, oF, oF2, trb);
    team_sync();
}

template <bool NOLL3, int N>
__device__ void v5_tile(const DevProb &Pmem, lchar *lds, const V4Lds LO, const int ti, const int tj, const int nsteps, const int C)
{
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int lane = threadIdx.x, role = lane & 3, team = lane >> 2;       // blockDim.x == 64
    const int capa = P.capa, capb = P.capb;
    const int ca4 = (capa + 3) & ~3, cb4 = (capb + 3) & ~3, lsz = ca4 + cb4;
    const int nslot = NOLL3 ? 9 : 6;
    const int pitch = v3_pitch(nslot, lsz);
    const int ndw = ((16 + 4 * (capa + capb) + 15) & ~15) / 4;
    lu32 *const rows = (lu32 *) (lds + LO.rows);           // row 0: staging (the strip above), row t+1: team t
    lu32 *const blk = (lu32 *) (lds + LO.black);
    lu32 *const stsc = (lu32 *) (lds + LO.stsc);           // staging scalars: H ring 0-2, G 3-4, G2 5-6
    li32 *const boff = (li32 *) (lds + LO.boff);           // b.off of the block's columns, 3 x (C + 2)
    li32 *const brg = (li32 *) (lds + LO.bring_g);         // column ring: [slot][view][V4_MLB]
    lf64 *const brf = (lf64 *) (lds + LO.bring_f);
    lu32 *const sink = (lu32 *) (lds + LO.sink) + lane;
#define V5_L(r, slot) (rows + (r) * pitch + (slot) * lsz)
    const size_t rbuf = (size_t) P.v2_rowstride * ndw;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const unsigned *rowHp = (const unsigned *) P.v2_rowH + bprev * rbuf, *rowGp = (const unsigned *) P.v2_rowG + bprev * rbuf;
    const unsigned *rowG2p = NOLL3 ? (const unsigned *) P.v2_rowG2 + bprev * rbuf : 0;
    unsigned *rowHc = (unsigned *) P.v2_rowH + bcur * rbuf, *rowGc = (unsigned *) P.v2_rowG + bcur * rbuf;
    unsigned *rowG2c = NOLL3 ? (unsigned *) P.v2_rowG2 + bcur * rbuf : 0;
    const unsigned *colH = (const unsigned *) P.v2_colH;
    unsigned *cbH = (unsigned *) P.v2_cbH, *cbF = (unsigned *) P.v2_cbF, *cbF2 = (unsigned *) P.v2_cbF2;
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const int m0 = a.left + ti * V5_R, m = m0 + team;
    const int mend = (m0 + V5_R < a.right) ? m0 + V5_R : a.right;
    const int tlast = mend - 1 - m0;                       // team of the strip's last row
    const int c0 = b.left + tj * C;
    int c1 = c0 + C; if (c1 > b.right) c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo > c0 ? nlo : c0, hi = nhi < c1 ? nhi : c1;      // ... clipped to this block
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left; if (cbase < c0) cbase = c0;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right; if (hi0 > c1) hi0 = c1;   // team 0's hi
    const bool vert0 = m0 > a.left;                        // the strip has a row above

    // ---- LDS init: black list, every ring slot black (reset(f1), reset(f2), fwd2c.h:385-386)
    if (lane < 4) blk[lane] = (lane == 1) ? (DL_END << 16) : 0;
    if (lane < 4) blk[ca4 + lane] = (lane == 1) ? (DL_END << 16) : 0;
    for (int sl = role; sl < nslot; sl += 4) {
        lu32 *p = V5_L(team + 1, sl);
        p[0] = 0; p[1] = DL_END << 16; p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    if (lane < nslot) {
        lu32 *p = V5_L(0, lane);
        p[0] = 0; p[1] = DL_END << 16; p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    for (int j = lane; j < 3 * (C + 2); j += 64) {         // list offsets of the block's columns (+ end)
        const int v = j / (C + 2), c = j - v * (C + 2);
        const int pos = c0 + c;
        const int *offv = v == 0 ? b.off[0] : v == 1 ? b.off[1] : b.off[2];   // (no runtime index into the descriptor copy)
        boff[j] = (pos <= b.len) ? offv[pos + 1] : offv[b.len + 1];
    }
    // ---- this lane's static list of a (by role): registers ---------------------------------------------------
    int a_g[N], a2_g[N];
    double a_f[N], a2_f[N];
    {
        // role 0, 1: a.s (vertical merges); role 2: a.r (horizontal); role 3: a.t (diagonal "b inside", newdelta)
        // and, for the long-gap penalty only, a.r as a second list (horizontal2)
        const int view = (role < 2) ? 0 : (role == 2) ? 2 : 1;
        DevSide av = a;                                    // the role's view in slot 0 (no runtime index into the copy)
        av.off[0] = view == 0 ? a.off[0] : view == 1 ? a.off[1] : a.off[2];
        av.glen[0] = view == 0 ? a.glen[0] : view == 1 ? a.glen[1] : a.glen[2];
        av.freq[0] = view == 0 ? a.freq[0] : view == 1 ? a.freq[1] : a.freq[2];
        rl_load(a_g, a_f, av, 0, m, row_ok);
        if (NOLL3) rl_load(a2_g, a2_f, a, 2, m, row_ok && role == 3);
    }
    // column ring loader: lane l moves entry (l & 15) of view (l >> 4) of the column team 0 reaches next
    const int ld_v = lane >> 4, ld_k = lane & 15;
    const int *const ld_glen = ld_v == 0 ? b.glen[0] : ld_v == 1 ? b.glen[1] : b.glen[2];
    const double *const ld_freq = ld_v == 0 ? b.freq[0] : ld_v == 1 ? b.freq[1] : b.freq[2];
    auto ring_load = [&](int col, int &rg_, double &rf_) {
        if (ld_v < 3) {
            const int o = boff[ld_v * (C + 2) + (col - c0)], e = boff[ld_v * (C + 2) + (col - c0) + 1];
            if (ld_k < e - o) { rg_ = ld_glen[o + ld_k]; rf_ = ld_freq[o + ld_k]; }
        }
    };
    auto ring_store = [&](int slot, int rg_, double rf_) {
        if (ld_v < 3) { brg[(slot * 3 + ld_v) * V4_MLB + ld_k] = rg_; brf[(slot * 3 + ld_v) * V4_MLB + ld_k] = rf_; }
    };
    // ---- the records this row starts from (all lanes of the team keep the scalars) ------------------------------
    RS oH = rs_black(), oG = rs_black(), oG2 = rs_black(), oF = rs_black(), oF2 = rs_black();
    {
        const bool cont = row_ok && c0 - 1 >= nlo && c0 - 1 < nhi;         // continues from the block on the left
        const unsigned *src = 0;
        if (cont) src = cbH + (size_t) (m - a.left) * ndw;                 // corner (m+1, c0)
        else if (row_ok && c0 == b.left && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left)
            src = colH + (size_t) (m + 1 - a.left) * ndw;                  // left boundary corner (m+1, b.left)
        lu32 *p = V5_L(team + 1, SLOT_H(c0));
        if (src) {
            oH.val = *(const double *) src; oH.dir = (int) src[2]; oH.glb = (int) src[3];
            for (int k = role; k < capa; k += 4) p[k] = src[4 + k];
            for (int k = role; k < capb; k += 4) p[ca4 + k] = src[4 + capa + k];
        }
        if (cont && lo < hi) {
            const unsigned *s2 = cbF + (size_t) (m - a.left) * ndw;
            p = V5_L(team + 1, SLOT_F);
            oF.val = *(const double *) s2; oF.dir = (int) s2[2]; oF.glb = (int) s2[3];
            for (int k = role; k < capa; k += 4) p[k] = s2[4 + k];
            for (int k = role; k < capb; k += 4) p[ca4 + k] = s2[4 + capa + k];
            if (NOLL3) {
                s2 = cbF2 + (size_t) (m - a.left) * ndw;
                p = V5_L(team + 1, SLOT_F2);
                oF2.val = *(const double *) s2; oF2.dir = (int) s2[2]; oF2.glb = (int) s2[3];
                for (int k = role; k < capa; k += 4) p[k] = s2[4 + k];
                for (int k = role; k < capb; k += 4) p[ca4 + k] = s2[4 + capa + k];
            }
        }
    }
    // ---- staging row: records of the strip above for team 0's columns, one dword per lane -------------------------
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < ndw) {
            const unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * ndw : rowHp + (size_t) col * ndw;
            rh = s[lane];
            if (wantG) { rg = rowGp[(size_t) col * ndw + lane]; if (NOLL3) rg2 = rowG2p[(size_t) col * ndw + lane]; }
        }
    };
    auto stage_put = [&](int slot, int sid, unsigned v) {
        const int j = lane - 4;
        if (lane < 4) stsc[sid * 4 + lane] = v;
        else if (j < capa) V5_L(0, slot)[j] = v;
        else if (j < capa + capb) V5_L(0, slot)[ca4 + j - capa] = v;
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < ndw) {
            stage_put(SLOT_H(col), SLOT_H(col), rh);
            if (wantG) { stage_put(SLOT_G(col), 3 + (col & 1), rg); if (NOLL3) stage_put(SLOT_G2(col), 5 + (col & 1), rg2); }
        }
    };
    if (lane < 28) stsc[lane] = 0;
    team_sync();
    {
        unsigned rh = 0, rg = 0, rg2 = 0;
        stage_load(cbase, false, rh, rg, rg2);
        stage_store(cbase, false, rh, rg, rg2);
        if (cbase + 1 <= c1) {
            stage_load(cbase + 1, vert0, rh, rg, rg2);
            stage_store(cbase + 1, vert0, rh, rg, rg2);
        }
        int g_ = 0; double f_ = 0;
        ring_load(cbase, g_, f_);                          // column of step 0 -> ring slot 0
        ring_store(0, g_, f_);
    }
    // per-row constants and one-step-ahead register pipelines (column score, b's column thickness)
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    const double *simrow = row_ok ? P.v2_sim + P.v2_rowoff[m - a.left] - nlo : 0;
    double sim_cur = 0, bc_cur = 0;
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;                                  // staging registers hold column n0 + 1
    int rg_nx = 0; double rf_nx = 0; bool ring_prev = false;   // ring registers hold column n0 + 1
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;    // the previous step's trace byte
    const int ull = __builtin_amdgcn_readfirstlane(tlast * 4);
    int lhi = m0 + tlast + P.up + 1; if (lhi > b.right) lhi = b.right; if (lhi > c1) lhi = c1;
    int llo = m0 + tlast + P.lw; if (llo < b.left) llo = b.left; if (llo < c0) llo = c0;
    auto flush_rows = [&](const int nl) {                  // nl: the last row's column in the step being flushed
        if (nl >= llo && nl < lhi) {
            const int col = nl + 1;
            const int j = lane - 4;
#pragma unroll
            for (int x = 0; x < (NOLL3 ? 3 : 2); ++x) {
                const RS &r = (x == 0) ? oH : (x == 1) ? oG : oG2;
                const int slot = (x == 0) ? SLOT_H(col) : (x == 1) ? SLOT_G(col) : SLOT_G2(col);
                const unsigned v0 = (unsigned) __builtin_amdgcn_readlane(__double2loint(r.val), ull);
                const unsigned v1 = (unsigned) __builtin_amdgcn_readlane(__double2hiint(r.val), ull);
                const unsigned v2 = (unsigned) __builtin_amdgcn_readlane(r.dir, ull), v3 = (unsigned) __builtin_amdgcn_readlane(r.glb, ull);
                unsigned v = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
                if (lane >= 4 && lane < ndw) {
                    const lu32 *p = V5_L(tlast + 1, slot);
                    v = (j < capa) ? p[j] : (j < capa + capb) ? p[ca4 + j - capa] : 0;
                }
                unsigned *dst = (x == 0) ? rowHc : (x == 1) ? rowGc : rowG2c;
                if (lane < ndw) dst[(size_t) col * ndw + lane] = v;
            }
        }
    };
    int rslot = (V5_RC - team % V5_RC) % V5_RC;            // ring slot of column cbase + s - team
    int wslot = 1;                                         // ring slot of column cbase + s + 1
    team_sync();
    for (int s = 0; s < nsteps; ++s) {
        const int n = cbase + s - team;
        const int n0 = cbase + s;                          // team 0's column
        const bool active = row_ok && n >= lo && n < hi;
        // -- top of the step: consume last step's loads, issue last step's stores
        if (st_prev) stage_store(n0 + 1, vert0, st_h, st_g, st_g2);
        if (ring_prev) ring_store(wslot == 0 ? V5_RC - 1 : wslot - 1, rg_nx, rf_nx);     // column n0 -> this step's slot of team 0
        if (p_act && role == 0) P.trace[p_tri] = (uint8_t) p_trb;
        if (wr_rows && s > 0) flush_rows(n0 - 1 - tlast);
        team_sync();
        // -- hand-over from the team above
        hd = hu;
        hu = rs_up4(oH); gu = rs_up4(oG);
        if (NOLL3) g2u = rs_up4(oG2);
        {
            const lu32 *q = stsc + SLOT_H(n0) * 4;
            RS t; t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hd = rs_sel(team == 0, t, hd);
            q = stsc + SLOT_H(n0 + 1) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hu = rs_sel(team == 0, t, hu);
            q = stsc + (3 + ((n0 + 1) & 1)) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            gu = rs_sel(team == 0, t, gu);
            if (NOLL3) {
                q = stsc + (5 + ((n0 + 1) & 1)) * 4;
                t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
                g2u = rs_sel(team == 0, t, g2u);
            }
        }
        // -- loads for the next step
        double sim_nx = 0, bc_nx = 0;
        if (active) {
            if (!have) { sim_cur = simrow[n]; bc_cur = thk_at(b, n)[0]; }
            if (n + 1 < hi) { sim_nx = simrow[n + 1]; bc_nx = thk_at(b, n + 1)[0]; }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) stage_load(n0 + 2, vert0, st_h, st_g, st_g2);
        ring_prev = n0 + 1 < c1;
        if (ring_prev) ring_load(n0 + 1, rg_nx, rf_nx);
        RS myH = oH, myG = oG, myG2 = oG2;                 // (the produced records of this step)
        if (active) {
            const bool do_hori = n > b.left;
            const li32 *cg = brg + (size_t) rslot * 3 * V4_MLB;
            const lf64 *cf = brf + (size_t) rslot * 3 * V4_MLB;
            LList bsl, btl, brl;
            bsl.glen = cg; bsl.freq = cf;
            btl.glen = cg + V4_MLB; btl.freq = cf + V4_MLB;
            brl.glen = cg + 2 * V4_MLB; brl.freq = cf + 2 * V4_MLB;
            const bool up_in = do_vert && (n - (m - 1) <= P.up);          // cell (m-1, n) exists
            const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
            const RS bk = rs_black();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const lu32 *hdl = V5_L(team, SLOT_H(n));
            const lu32 *hul = up_in ? V5_L(team, SLOT_H(n + 1)) : blk;
            const lu32 *gul = up_in ? V5_L(team, SLOT_G(n + 1)) : blk;
            const lu32 *g2ul = (NOLL3 && up_in) ? V5_L(team, SLOT_G2(n + 1)) : blk;
            const lu32 *hll = left_in ? V5_L(team + 1, SLOT_H(n)) : blk;
            const lu32 *fll = left_in ? V5_L(team + 1, SLOT_F) : blk;
            const lu32 *f2ll = (NOLL3 && left_in) ? V5_L(team + 1, SLOT_F2) : blk;
            lu32 *dh = V5_L(team + 1, SLOT_H(n + 1));
            lu32 *dg = V5_L(team + 1, SLOT_G(n + 1));
            lu32 *dg2 = V5_L(team + 1, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
            lu32 *df = V5_L(team + 1, SLOT_F);
            lu32 *df2 = V5_L(team + 1, NOLL3 ? SLOT_F2 : SLOT_F);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            int trb = 0;
            v5_cell<NOLL3, N>(P, ca4, role, a_g, a_f, a2_g, a2_f, bsl, btl, brl, sink, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                              dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb);
            const int d = m + n;
            int mlo, mhi;
            diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (d - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; have = (n + 1 < hi);
            // block boundary: this row's corner and F records for the block on the right
            if (n == c1 - 1 && c1 < b.right) {
                unsigned *o1 = cbH + (size_t) (m - a.left) * ndw, *o2 = cbF + (size_t) (m - a.left) * ndw;
                if (role == 0) {
                    *(double *) o1 = myH.val; o1[2] = (unsigned) myH.dir; o1[3] = (unsigned) myH.glb;
                    *(double *) o2 = oF.val; o2[2] = (unsigned) oF.dir; o2[3] = (unsigned) oF.glb;
                }
                for (int k = role; k < capa; k += 4) { o1[4 + k] = dh[k]; o2[4 + k] = df[k]; }
                for (int k = role; k < capb; k += 4) { o1[4 + capa + k] = dh[ca4 + k]; o2[4 + capa + k] = df[ca4 + k]; }
                if (NOLL3) {
                    unsigned *o3 = cbF2 + (size_t) (m - a.left) * ndw;
                    if (role == 0) { *(double *) o3 = oF2.val; o3[2] = (unsigned) oF2.dir; o3[3] = (unsigned) oF2.glb; }
                    for (int k = role; k < capa; k += 4) o3[4 + k] = df2[k];
                    for (int k = role; k < capb; k += 4) o3[4 + capa + k] = df2[ca4 + k];
                }
            }
            if (m == a.right - 1 && n == b.right - 1 && role == 0) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        if (++rslot == V5_RC) rslot = 0;
        if (++wslot == V5_RC) wslot = 0;
        team_sync();
    }
    if (p_act && role == 0) P.trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - tlast);
#undef V5_L
}

#define V4_KERNEL(NAME, TILE, N3, NA, WPE)                                                                \
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))      \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, V4Lds LO, int C) \
{                                                                                                   \
    extern __shared__ __attribute__((aligned(16))) char g2g_lds[];                                  \
    li32 *s_vals = (li32 *) ((lchar *) g2g_lds + LO.svals);                                         \
    for (;;) {                                                                                      \
        s_vals[threadIdx.x] = atomicAdd(qhead, threadIdx.x == 0 ? 1 : 0);                           \
        __syncthreads();                                                                            \
        const int t = __builtin_amdgcn_readfirstlane(s_vals[0]);                                    \
        __syncthreads();                                                                            \
        if (t >= ntiles) break;                                                                     \
        const V2Tile T = tiles[t];                                                                  \
        if (T.dep_up >= 0) v2_wait_flag(done + T.dep_up, gen, done + 16, t);                        \
        if (T.dep_left >= 0) v2_wait_flag(done + T.dep_left, gen, done + 16, t);                    \
        if (T.dep_diag >= 0) v2_wait_flag(done + T.dep_diag, gen, done + 16, t);                    \
        if (T.dep_war >= 0) v2_wait_flag(done + T.dep_war, gen, done + 16, t);                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
        TILE<N3, NA>(probs[T.prob], (lchar *) g2g_lds, LO, T.ti, T.tj, T.nsteps, C);                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __hip_atomic_store(done + T.self, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         \
    }                                                                                               \
}
#ifndef G2G_V4_WPE
#define G2G_V4_WPE 2
#endif
V4_KERNEL(g2g_v4_pf2, v4_tile, false, G2G_V3_NA, G2G_V4_WPE)
V4_KERNEL(g2g_v4_pf3, v4_tile, true, G2G_V3_NA, G2G_V4_WPE)
V4_KERNEL(g2g_v5_pf2, v5_tile, false, G2G_V3_NA, G2G_V4_WPE)
V4_KERNEL(g2g_v5_pf3, v5_tile, true, G2G_V3_NA, 1)
