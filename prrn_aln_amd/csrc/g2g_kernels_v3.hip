// g2g_kernels_v3.hip -- forward kernel of the gap-profile engines (_hf, _pf; Noll 2/3), ONE LANE PER CELL.
//
// Same recurrence and arithmetic order as v1/v2 (Fwd2c::forwardB, reference src/fwd2c.h:359-482 +
// src/fwd2c.cc:152-251 + src/gfreq.cc:493-605); the list primitives are those of g2g_kernels_v2.hip.
// The machine mapping differs:
//
//  * A tile is a strip of 64 rows x C columns and belongs to ONE WAVE: lane t owns row m0 + t and runs one
//    column behind lane t-1 (skewed wavefront inside the wave).  v2 spent 8 lanes on a cell to shorten one
//    DP's critical path; with the tile dataflow scheduler there are thousands of tiles in flight, so what
//    counts is cells per instruction: here every lane does a whole cell and a wave step yields 64 cells.
//  * Record SCALARS (val, dir, glb) never touch memory: a lane keeps the records it produced in registers
//    and hands them to the lane below with wave shuffles (DPP), one step later.
//  * Record LISTS ({glen,nins} pairs, 16+16 bit) live in LDS rings per row -- H: 3 slots, G: 2, F: 1
//    (+G2: 2, F2: 1) -- every list on a 16-byte boundary (one ds_read_b128 fetches the four leading
//    entries), row pitch an odd number of 16-byte units (conflict-free across lanes).
//  * The static gap profiles of the strip's rows and of the block's columns are contiguous ranges of the
//    HBM pools (lists of consecutive positions follow each other), so they are staged into LDS by three
//    coalesced range copies per side at tile start.
//  * Row 0 of the LDS rings is a staging row for the strip above (its last row's records, HBM rowH/rowG):
//    all 64 lanes move one record dword each, one column ahead of lane 0.  The strip's own last row goes to
//    HBM the same way, so HBM traffic is coalesced even though one lane produced the record.
// No workgroup barrier exists in the sweep (one wave): ordering LDS traffic between steps is a compiler
// fence only.
#include <hip/hip_runtime.h>

struct RS { double val; int dir, glb; };                  // record scalars
__device__ __forceinline__ RS rs_black() { RS r; r.val = NEVSEL; r.dir = 0; r.glb = 0; return r; }
// lane t <- lane t-1 over the whole wave: one DPP move per dword (wave_shr:1), no LDS crossbar round trip
__device__ __forceinline__ int dpp_up1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ RS rs_up(const RS &x)
{
    RS r;
    r.val = __hiloint2double(dpp_up1(__double2hiint(x.val)), dpp_up1(__double2loint(x.val)));
    r.dir = dpp_up1(x.dir); r.glb = dpp_up1(x.glb);
    return r;
}
__device__ __forceinline__ RS rs_sel(bool c, const RS &x, const RS &y)
{
    RS r; r.val = c ? x.val : y.val; r.dir = c ? x.dir : y.dir; r.glb = c ? x.glb : y.glb; return r;
}


// byte offsets of the LDS regions of a v3 launch (host: v3_lds_plan)
struct V3Lds { int rows, black, stsc, aglen, afreq, boff, bglen, bfreq, svals, sink, total; };

G2G_HD inline int v3_pitch(int nslot, int lsz)            // dwords per LDS row: odd number of 16-B units
{
    int p = nslot * lsz;
    if (((p >> 2) & 1) == 0) p += 4;
    return p;
}



// ---- branch-free list-head lookups ---------------------------------------------------------------------
// A dynamic list {glen,nins} is ascending in glen and ends with the terminator glen = 0xFFFF; what follows the
// terminator in memory is stale.  dhs_load() reads the four leading entries and overwrites everything behind the
// terminator by terminators ("sanitised" head): a lookup is then three independent unsigned compares of the
// packed key (g << 16 | 0xFFFF) against the packed entries -- no && chain (scalar mask logic), no branch per
// lookup.  Lists with more than three entries below g fall back to a scan of LDS behind ONE wave-uniform test.
__device__ __forceinline__ DHead dhs_load(const lu32 *p)
{
    const v4u32 v = *(const LDS v4u32 *) p;
    const unsigned T = DL_END << 16;
    DHead h;
    h.e0 = v.x;
    h.e1 = (v.x >= T) ? T : v.y;
    h.e2 = (h.e1 >= T) ? T : v.z;
    h.e3 = (h.e2 >= T) ? T : v.w;
    h.p = p;
    return h;
}
__device__ __forceinline__ int dhs_nins(const int g, const DHead &h)
{
    const unsigned key = ((unsigned) (g < 0 ? 0 : g) << 16) | 0xFFFFu;
    unsigned e = h.e0;
    e = key >= h.e1 ? h.e1 : e;
    e = key >= h.e2 ? h.e2 : e;
    const bool more = key >= h.e3;                          // (a terminator in e3 never compares below the key)
    e = more ? h.e3 : e;
    if (__ballot(more)) {                                   // rare: more than three entries below g
        if (more) {
            int k = 3;
            while (key >= ((h.p[k + 1] & 0xFFFF0000u) ) && k < DL_GUARD) ++k;
            e = h.p[k];
        }
    }
    return (int) (e & 0xFFFFu);
}

#ifndef G2G_V3_HF_UNROLL
#define G2G_V3_HF_UNROLL 16               // _hf merge loops fully unrolled (faster than s_set_gpr_idx indexing); _pf loops stay rolled (code size)
#endif
// ---- static lists in REGISTERS ------------------------------------------------------------------------
// A lane's row never changes inside a tile, so the row's three static lists (s/t/r views, <= NA entries
// each incl. the terminator) are loaded into registers once per tile.  Every loop over such a list runs
// with the SAME index in all lanes (fully unrolled, lanes that are done are masked by selects, the wave
// leaves when no lane is live), so the register arrays are only ever indexed by constants: no LDS traffic,
// no dependent-load chain, no divergent control flow in the merge loops.
// (separate plain arrays, not a struct of arrays: the backend promotes an array to registers -- indexed through
// s_set_gpr_idx by a uniform loop counter -- only when the array is an alloca of its own; build flag
// -amdgpu-promote-alloca-to-vector-limit raises the budget so that all six fit)
template <int N>
__device__ __forceinline__ void rl_load(int (&g)[N], double (&f)[N], const DevSide &s, int view, int pos, bool ok)
{
    const int o = ok ? s.off[view][pos + 1] : 0, len = ok ? s.off[view][pos + 2] - o : 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool v = i < len;
        g[i] = v ? s.glen[view][o + i] : -1;
        f[i] = v ? s.freq[view][o + i] : 0.;
    }
}
__device__ __forceinline__ bool wave_none(bool live) { return __ballot(live) == 0; }
// (the r view is not held: r = [head {glen 0, freq rhf}, if the position has one] + the t entries with glen + 1 -- DevSide::r_from_t,
// checked by the host; 48 registers less than a third register list, which is what lets the kernel run two waves per SIMD unspilled)
template <int N> struct ARegs { const int (&sg)[N]; const double (&sf)[N]; const int (&tg)[N]; const double (&tf)[N]; double rhf; bool rh; };
template <int N> __device__ __forceinline__ int areg_rg(const ARegs<N> &A, const int i)
{
    if (i == 0) return A.rh ? 0 : (A.tg[0] >= 0 ? A.tg[0] + 1 : -1);
    const int t = A.rh ? A.tg[i - 1] : A.tg[i];
    return t >= 0 ? t + 1 : -1;
}
template <int N> __device__ __forceinline__ double areg_rf(const ARegs<N> &A, const int i)
{
    if (i == 0) return A.rh ? A.rhf : A.tf[0];
    return A.rh ? A.tf[i - 1] : A.tf[i];
}

// ---- one cell by one lane ---------------------------------------------------------------------------
struct Costs { double d0, d1, gnpv, gopv, gnph, goph, gnpv2, gnph2; };
struct Dec {
    int win, bits;                                  // win: 0 diag, 1 G, 2 G2, 3 F, 4 F2
    double hval, gval, g2val, fval, f2val;
    int hdir, gdir, g2dir, fdir, f2dir;
    bool g_from_h, g2_from_h, f_from_h, f2_from_h;
};
// the decisions of one cell (fwd2c.h:395-453), from the gap-open costs and the neighbours' scalars
template <int KIND, bool NOLL3>
__device__ __forceinline__ Dec v3_decide(const DevProb &P, const Costs &c, const RS &hd, const RS &hu, const RS &gu,
    const RS &g2u, const RS &hl, const RS &fl, const RS &f2l, const bool do_vert, const bool do_hori,
    const double dab, const double pua, const double pub)
{
    Dec d;
    double gop = (KIND == 2) ? c.d0 + c.d1 : c.d0;
    d.hval = hd.val + (dab + gop);
    d.hdir = isdiag(hd.dir) ? D_DIAG : D_NEWD;
    d.bits = 0; d.win = 0;
    double mxval = NEVSEL;                          // mx = g: at the first row G is a black record
    d.gval = d.g2val = d.fval = d.f2val = 0;
    d.gdir = d.g2dir = d.fdir = d.f2dir = 0;
    d.g_from_h = d.g2_from_h = d.f_from_h = d.f2_from_h = false;
    if (do_vert) {
        const double gnp = c.gnpv;
        gop = c.gopv;
        const bool hu_nv = !isvert(hu.dir);
        d.g_from_h = hu_nv && (hu.val + gop > gu.val + gnp);
        d.gdir = ishori(d.g_from_h ? hu.dir : gu.dir) ? D_NEWV : D_VERT;
        d.gval = (d.g_from_h ? hu.val : gu.val) + (d.g_from_h ? gop : gnp);
        d.gval += pua;
        if (!d.g_from_h) d.bits |= T_GEXT;
        mxval = d.gval; d.win = 1;
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c.gnpv2;
            gop = P.v2divv1 * gop;
            d.g2_from_h = hu_nv && (hu.val + gop > g2u.val + gnp2);
            d.g2dir = ishori(d.g2_from_h ? hu.dir : g2u.dir) ? D_NEWV : D_VERT;
            d.g2val = (d.g2_from_h ? hu.val : g2u.val) + (d.g2_from_h ? gop : gnp2);
            d.g2val += P.u2divu1 * pua;
            if (!d.g2_from_h) d.bits |= T_G2EXT;
            if (d.g2val > mxval) { mxval = d.g2val; d.win = 2; }
        }
    } else {
        mxval = NEVSEL; d.win = 1;                   // first row: mx starts as the untouched black G
    }
    if (do_hori) {
        const double gnp = c.gnph;
        gop = c.goph;
        const bool hl_nh = !ishori(hl.dir);
        d.f_from_h = hl_nh && (hl.val + gop > fl.val + gnp);
        d.fdir = isvert(d.f_from_h ? hl.dir : fl.dir) ? D_NEWH : D_HORI;
        d.fval = (d.f_from_h ? hl.val : fl.val) + (d.f_from_h ? gop : gnp);
        d.fval += pub;
        if (!d.f_from_h) d.bits |= T_FEXT;
        if (d.fval >= mxval) { mxval = d.fval; d.win = 3; }
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c.gnph2;
            gop = P.v2divv1 * gop;
            d.f2_from_h = hl_nh && (hl.val + gop > f2l.val + gnp2);
            d.f2dir = isvert(d.f2_from_h ? hl.dir : f2l.dir) ? D_NEWH : D_HORI;
            d.f2val = (d.f2_from_h ? hl.val : f2l.val) + (d.f2_from_h ? gop : gnp2);
            d.f2val += P.u2divu1 * pub;
            if (!d.f2_from_h) d.bits |= T_F2EXT;
            if (d.f2val >= mxval) { mxval = d.f2val; d.win = 4; }
        }
    }
    if (!(mxval > d.hval)) d.win = 0;                  // diagonal wins ties (fwd2c.h:453)
    if (!do_vert && d.win == 1) d.win = 0;             // (the black G can never win)
    return d;
}
// scalars of the produced records (glb per fwd2c.cc:169,173,177) and the trace byte
template <int KIND, bool NOLL3>
__device__ __forceinline__ void v3_outputs(const Dec &d, const int gs_glb, const int gs2_glb, const bool do_vert, const bool do_hori,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb)
{
    const int glb_g = (KIND == 1 && do_vert) ? gs_glb + 1 : 0;
    const int glb_g2 = (KIND == 1 && NOLL3 && do_vert) ? gs2_glb + 1 : 0;
    if (do_vert) { oG.val = d.gval; oG.dir = d.gdir; oG.glb = glb_g; }
    if (NOLL3 && do_vert) { oG2.val = d.g2val; oG2.dir = d.g2dir; oG2.glb = glb_g2; }
    if (do_hori) { oF.val = d.fval; oF.dir = d.fdir; oF.glb = 0; }
    if (NOLL3 && do_hori) { oF2.val = d.f2val; oF2.dir = d.f2dir; oF2.glb = 0; }
    double hv = d.hval; int hdr = d.hdir, hg = 0;
    if (d.win == 1) { hv = d.gval; hdr = d.gdir; hg = glb_g; }
    else if (d.win == 2) { hv = d.g2val; hdr = d.g2dir; hg = glb_g2; }
    else if (d.win == 3) { hv = d.fval; hdr = d.fdir; }
    else if (d.win == 4) { hv = d.f2val; hdr = d.f2dir; }
    oH.val = hv; oH.dir = hdr; oH.glb = hg;
    int bits = d.bits;
    if (d.win == 2 || d.win == 4) bits |= T_SEL2;
    trb = bits | dir2code(hdr);
}

// xl: the record's dla list (its dlb list follows ca4 dwords later); d*: destination lists
template <int KIND, bool NOLL3>
__device__ __forceinline__ void v3_cell(const DevProb &P, const int ca4, const CellLists<LList> &L,
    const RS &hd, const lu32 *hdl, const RS &hu, const lu32 *hul, const RS &gu, const lu32 *gul,
    const RS &g2u, const lu32 *g2ul, const RS &hl, const lu32 *hll, const RS &fl, const lu32 *fll,
    const RS &f2l, const lu32 *f2ll,
    lu32 *dh, lu32 *dg, lu32 *dg2, lu32 *df, lu32 *df2,
    const bool do_vert, const bool do_hori, const double dab, const double pua, const double pub,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb)
{
    // ---- gap-open costs (fwd2c.cc:152-160, 203-212) ------------------------------------------------
    Costs c;
    c.d1 = c.gnpv = c.gopv = c.gnph = c.goph = c.gnpv2 = c.gnph2 = 0;
    if (KIND == 2) {
        c.d0 = p_newgap4<LList, LList, true>(L.as, hdl, L.bt, hdl + ca4) * P.basic_gop;
        c.d1 = p_newgap4<LList, LList, true>(L.bs, hdl + ca4, L.at, hdl) * P.basic_gop;
        if (do_vert) {
            c.gnpv = p_newgap4<LList, LList, true>(L.as, gul, L.br, gul + ca4) * P.basic_gop;
            c.gopv = p_newgap4<LList, LList, true>(L.as, hul, L.br, hul + ca4) * P.basic_gop;
            if (NOLL3) c.gnpv2 = p_newgap4<LList, LList, true>(L.as, g2ul, L.br, g2ul + ca4) * P.basic_gop;
        }
        if (do_hori) {
            c.gnph = p_newgap4<LList, LList, true>(L.bs, fll + ca4, L.ar, fll) * P.basic_gop;
            c.goph = p_newgap4<LList, LList, true>(L.bs, hll + ca4, L.ar, hll) * P.basic_gop;
            if (NOLL3) c.gnph2 = p_newgap4<LList, LList, true>(L.bs, f2ll + ca4, L.ar, f2ll) * P.basic_gop;
        }
    } else {
        c.d0 = p_newgap2<LList, true>(P, L.at, hd.glb, hdl);
        if (do_vert) {
            c.gnpv = p_newgap1<LList, true>(P, L.as, gul, gu.glb);
            c.gopv = p_newgap1<LList, true>(P, L.as, hul, hu.glb);
            if (NOLL3) c.gnpv2 = p_newgap1<LList, true>(P, L.as, g2ul, g2u.glb);
        }
        if (do_hori) {
            c.gnph = p_newgap2<LList, true>(P, L.ar, fl.glb, fll);
            c.goph = p_newgap2<LList, true>(P, L.ar, hl.glb, hll);
            if (NOLL3) c.gnph2 = p_newgap2<LList, true>(P, L.ar, f2l.glb, f2ll);
        }
    }
    const Dec d = v3_decide<KIND, NOLL3>(P, c, hd, hu, gu, g2u, hl, fl, f2l, do_vert, do_hori, dab, pua, pub);
    const int win = d.win;
    // ---- list updates (update(), fwd2c.cc:165-180 / 216-231); the winner's lists also become H's ----
    const lu32 *gsl = d.g_from_h ? hul : gul, *gs2l = d.g2_from_h ? hul : g2ul;
    const lu32 *fsl = d.f_from_h ? hll : fll, *fs2l = d.f2_from_h ? hll : f2ll;
    lu32 *const nul = (lu32 *) 0;
    if (do_vert) {
        p_newdelta<LList, true>(dg, win == 1 ? dh : nul, L.at, gsl);
        if (KIND == 2) p_incdelta(dg + ca4, win == 1 ? dh + ca4 : nul, gsl + ca4);
        if (NOLL3) {
            p_newdelta<LList, true>(dg2, win == 2 ? dh : nul, L.at, gs2l);
            if (KIND == 2) p_incdelta(dg2 + ca4, win == 2 ? dh + ca4 : nul, gs2l + ca4);
        }
    }
    if (do_hori) {
        if (KIND == 2) p_newdelta<LList, true>(df + ca4, win == 3 ? dh + ca4 : nul, L.bt, fsl + ca4);
        p_incdelta(df, win == 3 ? dh : nul, fsl);
        if (NOLL3) {
            if (KIND == 2) p_newdelta<LList, true>(df2 + ca4, win == 4 ? dh + ca4 : nul, L.bt, fs2l + ca4);
            p_incdelta(df2, win == 4 ? dh : nul, fs2l);
        }
    }
    if (win == 0) {
        p_newdelta<LList, true>(dh, nul, L.at, hdl);
        if (KIND == 2) p_newdelta<LList, true>(dh + ca4, nul, L.bt, hdl + ca4);
    }
    v3_outputs<KIND, NOLL3>(d, d.g_from_h ? hu.glb : gu.glb, d.g2_from_h ? hu.glb : g2u.glb, do_vert, do_hori, oH, oG, oG2, oF, oF2, trb);
}

// ---- the _hf cell with the row's static lists in registers -------------------------------------------
// All merges of the cell advance together through ONE uniform loop (entry i of the s, t and r lists in the
// same iteration): the five to seven recurrences are independent, so the instruction stream has that much
// ILP, and there is a single loop-exit test per entry instead of one per merge.
struct NG { bool live; double acc; };
// newgap(cf, dlc, j), gfreq.cc:523-532: first entry whose stretched length reaches glb
__device__ __forceinline__ void ng1_step(NG &s, const DHead &h, const int glb, const int cg, const double cf)
{
    s.live = s.live && cg >= 0;
    const bool hit = s.live && (cg + dh_nins(cg, h) >= glb);
    s.acc = hit ? cf : s.acc;
    s.live = s.live && !hit;
}
// newgap(df, i, dld), gfreq.cc:534-545: sum of the entries whose stretched length stays within glb
__device__ __forceinline__ void ng2_step(NG &s, const DHead &h, const int glb, const int dg, const double df)
{
    s.live = s.live && dg >= 0;
    s.live = s.live && !(glb < dg + dh_nins(dg, h));
    s.acc = s.live ? s.acc + df : s.acc;
}
__device__ __forceinline__ double ng1_fin(const DevProb &P, const NG &s, const DHead &h, const int glb, const int g0, const int g1, const double f0)
{   // newgap1, maln.h:296-301
    const double rm = P.weighted_gop * s.acc;
    const double rs = ((int) (h.e0 & 0xFFFFu) + g0 >= glb) ? (P.weighted_gop * f0) : 0;
    return g0 < 0 ? 0 : g1 >= 0 ? rm : rs;
}
__device__ __forceinline__ double ng2_fin(const DevProb &P, const NG &s, const DHead &h, const int glb, const int g0, const int g1, const double f0)
{   // newgap2, maln.h:303-308
    const double rm = P.weighted_gop * s.acc;
    const double rs = (glb >= (int) (h.e0 & 0xFFFFu) + g0) ? (P.weighted_gop * f0) : 0;
    return g0 < 0 ? 0 : g1 >= 0 ? rm : rs;
}
__device__ __forceinline__ DHead dh_sel(const bool c, const DHead &x, const DHead &y)
{
    DHead h; h.e0 = c ? x.e0 : y.e0; h.e1 = c ? x.e1 : y.e1; h.e2 = c ? x.e2 : y.e2; h.e3 = c ? x.e3 : y.e3; h.p = c ? x.p : y.p; return h;
}
// newdelta (gfreq.cc:570-587) as a step function; stores are unconditional, a lane that has nothing to
// store writes into its private sink dword
struct ND { int kd; unsigned tg, tn; bool on; };
__device__ __forceinline__ void nd_step(ND &s, const DHead &h, const int g, lu32 *d1, lu32 *d2, lu32 *sink)
{
    const unsigned sn = (unsigned) dh_nins(g, h);
    const bool emit = s.on && g >= 0 && sn > s.tn;
    const unsigned e = (s.tg << 16) | s.tn;
    *(emit ? d1 + s.kd : sink) = e;
    *((emit && d2) ? d2 + s.kd : sink) = e;
    s.kd += emit ? 1 : 0;
    s.tn = emit ? sn : s.tn;
    s.tg = emit ? (unsigned) (g + 1) : s.tg;
    s.on = s.on && g >= 0;
}
__device__ __forceinline__ void nd_fin(const ND &s, const bool was_on, lu32 *d1, lu32 *d2, lu32 *sink)
{
    const unsigned e = (s.tg << 16) | s.tn;
    *(was_on ? d1 + s.kd : sink) = e;
    *(was_on ? d1 + s.kd + 1 : sink) = DL_END << 16;
    *((was_on && d2) ? d2 + s.kd : sink) = e;
    *((was_on && d2) ? d2 + s.kd + 1 : sink) = DL_END << 16;
}
// incdelta (gfreq.cc:598-605) from a list head held in registers
__device__ __forceinline__ void incdelta_h(const bool on, const DHead &h, lu32 *d1, lu32 *d2, lu32 *sink)
{
    const bool t0 = (h.e0 >> 16) == DL_END, t1 = t0 || (h.e1 >> 16) == DL_END;
    const bool t2 = t1 || (h.e2 >> 16) == DL_END, t3 = t2 || (h.e3 >> 16) == DL_END;
    const bool o2 = on && d2;
    // entry k is written iff no earlier entry was the terminator; the terminator itself is copied unchanged
    const unsigned w0 = t0 ? h.e0 : h.e0 + 1, w1 = ((h.e1 >> 16) == DL_END) ? h.e1 : h.e1 + 1;
    const unsigned w2 = ((h.e2 >> 16) == DL_END) ? h.e2 : h.e2 + 1, w3 = ((h.e3 >> 16) == DL_END) ? h.e3 : h.e3 + 1;
    *(on ? d1 : sink) = w0;             *(o2 ? d2 : sink) = w0;
    *((on && !t0) ? d1 + 1 : sink) = w1; *((o2 && !t0) ? d2 + 1 : sink) = w1;
    *((on && !t1) ? d1 + 2 : sink) = w2; *((o2 && !t1) ? d2 + 2 : sink) = w2;
    *((on && !t2) ? d1 + 3 : sink) = w3; *((o2 && !t2) ? d2 + 3 : sink) = w3;
    if (on && !t3) {                                        // rare: more than three entries
        for (int k = 4; k < DL_GUARD; ++k) {
            unsigned e = h.p[k];
            if ((e >> 16) == DL_END) { d1[k] = e; if (d2) d2[k] = e; break; }
            e += 1;
            d1[k] = e; if (d2) d2[k] = e;
        }
    }
}

template <bool NOLL3, int N>
__device__ __forceinline__ void v3_cell_hf(const DevProb &P, const ARegs<N> &A, lu32 *sink,
    const RS &hd, const lu32 *hdl, const RS &hu, const lu32 *hul, const RS &gu, const lu32 *gul,
    const RS &g2u, const lu32 *g2ul, const RS &hl, const lu32 *hll, const RS &fl, const lu32 *fll,
    const RS &f2l, const lu32 *f2ll,
    lu32 *dh, lu32 *dg, lu32 *dg2, lu32 *df, lu32 *df2,
    const bool do_vert, const bool do_hori, const double dab, const double pua, const double pub,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb)
{
    // every dynamic-list head this cell reads, fetched up front (independent ds_read_b128s, one wait)
    const DHead h_hd = dh_load<true>(hdl), h_gu = dh_load<true>(gul), h_hu = dh_load<true>(hul);
    const DHead h_fl = dh_load<true>(fll), h_hl = dh_load<true>(hll);
    const DHead h_g2u = dh_load<true>(NOLL3 ? g2ul : gul), h_f2l = dh_load<true>(NOLL3 ? f2ll : fll);
    // ---- gap-open costs: one fused loop over the static entries ---------------------------------------
    const bool ms = A.sg[0] >= 0 && A.sg[1] >= 0, mt = A.tg[0] >= 0 && A.tg[1] >= 0, mr = areg_rg(A, 0) >= 0 && areg_rg(A, 1) >= 0;
    NG s_d = {mt, 0.}, s_gu = {ms && do_vert, 0.}, s_hu = {ms && do_vert, 0.}, s_g2 = {ms && do_vert && NOLL3, 0.};
    NG s_fl = {mr && do_hori, 0.}, s_hl = {mr && do_hori, 0.}, s_f2 = {mr && do_hori && NOLL3, 0.};
#pragma unroll G2G_V3_HF_UNROLL
    for (int i = 0; i < N; ++i) {
        if (wave_none(s_d.live || s_gu.live || s_hu.live || s_fl.live || s_hl.live || (NOLL3 && (s_g2.live || s_f2.live)))) break;
        ng2_step(s_d, h_hd, hd.glb, A.tg[i], A.tf[i]);
        ng1_step(s_gu, h_gu, gu.glb, A.sg[i], A.sf[i]);
        ng1_step(s_hu, h_hu, hu.glb, A.sg[i], A.sf[i]);
        const int rgi = areg_rg(A, i);
        const double rfi = areg_rf(A, i);
        ng2_step(s_fl, h_fl, fl.glb, rgi, rfi);
        ng2_step(s_hl, h_hl, hl.glb, rgi, rfi);
        if (NOLL3) {
            ng1_step(s_g2, h_g2u, g2u.glb, A.sg[i], A.sf[i]);
            ng2_step(s_f2, h_f2l, f2l.glb, rgi, rfi);
        }
    }
    Costs c;
    c.d1 = 0;
    c.d0 = ng2_fin(P, s_d, h_hd, hd.glb, A.tg[0], A.tg[1], A.tf[0]);
    c.gnpv = ng1_fin(P, s_gu, h_gu, gu.glb, A.sg[0], A.sg[1], A.sf[0]);
    c.gopv = ng1_fin(P, s_hu, h_hu, hu.glb, A.sg[0], A.sg[1], A.sf[0]);
    c.gnph = ng2_fin(P, s_fl, h_fl, fl.glb, areg_rg(A, 0), areg_rg(A, 1), areg_rf(A, 0));
    c.goph = ng2_fin(P, s_hl, h_hl, hl.glb, areg_rg(A, 0), areg_rg(A, 1), areg_rf(A, 0));
    c.gnpv2 = NOLL3 ? ng1_fin(P, s_g2, h_g2u, g2u.glb, A.sg[0], A.sg[1], A.sf[0]) : 0;
    c.gnph2 = NOLL3 ? ng2_fin(P, s_f2, h_f2l, f2l.glb, areg_rg(A, 0), areg_rg(A, 1), areg_rf(A, 0)) : 0;
    const Dec d = v3_decide<1, NOLL3>(P, c, hd, hu, gu, g2u, hl, fl, f2l, do_vert, do_hori, dab, pua, pub);
    const int win = d.win;
    // ---- list updates: the newdeltas of G (G2) and of a diagonal H share one loop over the t list -----
    lu32 *const nul = (lu32 *) 0;
    const DHead h_gs = dh_sel(d.g_from_h, h_hu, h_gu), h_gs2 = dh_sel(d.g2_from_h, h_hu, h_g2u);
    ND n_g = {0, 0, 0, do_vert}, n_h = {0, 0, 0, win == 0}, n_g2 = {0, 0, 0, do_vert && NOLL3};
    lu32 *const g_d2 = win == 1 ? dh : nul, *const g2_d2 = win == 2 ? dh : nul;
#pragma unroll G2G_V3_HF_UNROLL
    for (int i = 0; i < N; ++i) {
        if (wave_none(n_g.on || n_h.on || (NOLL3 && n_g2.on))) break;
        nd_step(n_g, h_gs, A.tg[i], dg, g_d2, sink);
        nd_step(n_h, h_hd, A.tg[i], dh, nul, sink);
        if (NOLL3) nd_step(n_g2, h_gs2, A.tg[i], dg2, g2_d2, sink);
    }
    nd_fin(n_g, do_vert, dg, g_d2, sink);
    nd_fin(n_h, win == 0, dh, nul, sink);
    if (NOLL3) nd_fin(n_g2, do_vert, dg2, g2_d2, sink);
    incdelta_h(do_hori, dh_sel(d.f_from_h, h_hl, h_fl), df, win == 3 ? dh : nul, sink);
    if (NOLL3) incdelta_h(do_hori, dh_sel(d.f2_from_h, h_hl, h_f2l), df2, win == 4 ? dh : nul, sink);
    v3_outputs<1, NOLL3>(d, d.g_from_h ? hu.glb : gu.glb, d.g2_from_h ? hu.glb : g2u.glb, do_vert, do_hori, oH, oG, oG2, oF, oF2, trb);
}

// ---- the _pf cell: the row's static lists in registers, the column's in LDS ---------------------------
// Fwd2c<_pf>::gapopen needs six merges of a static list of a against a static list of b, each stretched by the
// record's dynamic lists (newgap, gfreq.cc:507-521).  A two-pointer merge walks one list with a data-dependent
// index; that one is always taken to be b's (short, in LDS, its first four entries cached in registers), while
// a's list is the uniform outer loop.  For the merges whose reference form has a's list inside (cf = a.s), the
// loops are interchanged: entry ci of cf then consumes every entry di of df it is the match of, in di order, so
// the products are added in exactly the reference's order.
struct BHead { int g0, g1, g2, g3; double f0, f1, f2, f3; LList l; };
__device__ __forceinline__ BHead bh_load(const LList l)
{
    BHead h;
    h.g0 = l.glen[0]; h.g1 = l.glen[1]; h.g2 = l.glen[2]; h.g3 = l.glen[3];
    h.f0 = l.freq[0]; h.f1 = l.freq[1]; h.f2 = l.freq[2]; h.f3 = l.freq[3];
    h.l = l; return h;
}
// entry i: selects over the cached four; beyond that one LDS read behind a wave-uniform test (a conditional load
// inside the select chain would be compiled into a tree of divergent branches)
__device__ __forceinline__ int bh_glen(const BHead &h, const int i)
{
    int g = i == 0 ? h.g0 : i == 1 ? h.g1 : i == 2 ? h.g2 : h.g3;
    if (__ballot(i > 3)) { const int t = h.l.glen[i > 3 ? i : 0]; g = i > 3 ? t : g; }
    return g;
}
__device__ __forceinline__ double bh_freq(const BHead &h, const int i)
{
    double f = i == 0 ? h.f0 : i == 1 ? h.f1 : i == 2 ? h.f2 : h.f3;
    if (__ballot(i > 3)) { const double t = h.l.freq[i > 3 ? i : 0]; f = i > 3 ? t : f; }
    return f;
}
// cf = a's list (outer, registers), df = b's list (inner): newgap4(a.s, dla, b.t|b.r, dlb)
struct MY { int di, dg, j; bool live; double g; };
__device__ __forceinline__ void my_init(MY &s, const bool on, const BHead &df, const DHead &dd)
{
    s.di = 0; s.dg = df.g0; s.j = s.dg >= 0 ? s.dg + dh_nins(s.dg, dd) : 0; s.live = on; s.g = 0;
}
__device__ __forceinline__ void my_step(MY &s, const int cg, const double cff, const DHead &hc, const BHead &df, const DHead &dd)
{
    s.live = s.live && cg >= 0;
    const int gi = cg + dh_nins(cg, hc);
    bool adv = s.live && s.dg >= 0 && s.j <= gi;
    while (__ballot(adv)) {
        if (adv) {
            s.g += cff * bh_freq(df, s.di);
            ++s.di;
            s.dg = bh_glen(df, s.di);
            s.j = s.dg >= 0 ? s.dg + dh_nins(s.dg, dd) : 0;
        }
        adv = adv && s.dg >= 0 && s.j <= gi;
    }
    s.live = s.live && s.dg >= 0;
}
// df = a's list (outer, registers), cf = b's list (inner): newgap4(b.s, dlb, a.t|a.r, dla)
struct MX { int ci, cg, gi; bool live; double g; };
__device__ __forceinline__ void mx_init(MX &s, const bool on, const BHead &cf, const DHead &hc)
{
    s.ci = 0; s.cg = cf.g0; s.gi = s.cg >= 0 ? s.cg + dh_nins(s.cg, hc) : 0; s.live = on; s.g = 0;
}
__device__ __forceinline__ void mx_step(MX &s, const int dg, const double dff, const DHead &hd_, const BHead &cf, const DHead &hc)
{
    s.live = s.live && dg >= 0;
    const int j = dg + dh_nins(dg, hd_);
    bool adv = s.live && s.cg >= 0 && s.gi < j;
    while (__ballot(adv)) {
        if (adv) {
            ++s.ci;
            s.cg = bh_glen(cf, s.ci);
            s.gi = s.cg >= 0 ? s.cg + dh_nins(s.cg, hc) : 0;
        }
        adv = adv && s.cg >= 0 && s.gi < j;
    }
    s.live = s.live && s.cg >= 0;
    s.g = s.live ? s.g + bh_freq(cf, s.ci) * dff : s.g;
}


// record image in HBM: {f64 val; i32 dir; i32 glb; u32 dla[capa]; u32 dlb[capb]} (the v2 format, which the
// boundary-chain prologue writes): dword k of it <-> scalar k / list entry k - 4
__device__ __forceinline__ void v3_rec_load(const unsigned *src, int capa, int capb, RS &r, lu32 *la, lu32 *lb)
{
    r.val = *(const double *) src; r.dir = (int) src[2]; r.glb = (int) src[3];
    for (int k = 0; k < capa; ++k) la[k] = src[4 + k];
    for (int k = 0; k < capb; ++k) lb[k] = src[4 + capa + k];
}
__device__ __forceinline__ void v3_rec_store(unsigned *dst, int capa, int capb, const RS &r, const lu32 *la, const lu32 *lb)
{
    *(double *) dst = r.val; dst[2] = (unsigned) r.dir; dst[3] = (unsigned) r.glb;
    for (int k = 0; k < capa; ++k) dst[4 + k] = la[k];
    for (int k = 0; k < capb; ++k) dst[4 + capa + k] = lb[k];
}

template <int KIND, bool NOLL3, int NA>
__device__ __forceinline__ void v3_tile(const DevProb &Pmem, lchar *lds, const V3Lds LO, const int ti, const int tj, const int nsteps, const int C,
                        const int *prog_up = 0, int *prog_self = 0, int *dbg = 0, const int pgen = 0, const int pint = 32,
                        const int *prog_left = 0, double *simscr = 0, int *failp = 0)
{
    // SWEEP MODE (prog_self != 0): the tile is a whole strip (C covers the row range) and the dependency on the strip
    // above is a progress counter instead of tile-completion flags: the strip above publishes, every pint (16/32) steps, up to
    // which corner column its last row's records are final ((generation << 20) | column); this strip waits only
    // before it stages a column beyond what it has seen published.  Strips of one DP then run as a pipeline, each a
    // few dozen columns behind its predecessor: no fill/drain per column block, no block-boundary records, and a DP's
    // critical path is strips x (skew + publish interval) instead of (strips + blocks) x tile.
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int lane = threadIdx.x;                          // blockDim.x == 64
    const int capa = P.capa, capb = (KIND == 2) ? P.capb : 0;
    const int ca4 = (capa + 3) & ~3, cb4 = (capb + 3) & ~3, lsz = ca4 + cb4;
    const int nslot = NOLL3 ? 9 : 6;
    const int pitch = v3_pitch(nslot, lsz);
    const int ndw = ((16 + 4 * (capa + capb) + 15) & ~15) / 4;
    lu32 *const rows = (lu32 *) (lds + LO.rows);           // row 0: staging (the strip above), row t+1: lane t
    lu32 *const blk = (lu32 *) (lds + LO.black);
    lu32 *const stsc = (lu32 *) (lds + LO.stsc);           // staging scalars: H ring 0-2, G 3-4, G2 5-6
#define V3_L(r, slot) (rows + (r) * pitch + (slot) * lsz)
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
    const int m0 = a.left + ti * 64, m = m0 + lane;
    if (prog_left) {                                       // sweep mode: the left boundary chain runs beside the strips (v2_chain_tile)
        const int rows_ = m0 + 64 - a.left;
        const int wantl = ((pgen & 0x7FF) << 20) | (rows_ < 0xFFFFF ? rows_ : 0xFFFFF);
        (void) g2g_wait_ge(prog_left, wantl, dbg, failp, ti);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int mend = (m0 + 64 < a.right) ? m0 + 64 : a.right;
    const int llast = mend - 1 - m0;                       // lane of the strip's last row
    const int c0 = b.left + tj * C;
    int c1 = c0 + C; if (c1 > b.right) c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo > c0 ? nlo : c0, hi = nhi < c1 ? nhi : c1;      // ... clipped to this block
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left; if (cbase < c0) cbase = c0;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right; if (hi0 > c1) hi0 = c1;   // lane 0's hi
    const bool vert0 = m0 > a.left;                        // the strip has a row above

    // ---- LDS init: black list, every ring slot black (reset(f1), reset(f2), fwd2c.h:385-386; G of the
    // DP's first row is never written and is read as black by the row below, fwd2c.h:401)
    if (lane < 4) blk[lane] = (lane == 1) ? (DL_END << 16) : 0;
    if (lane < 4) blk[ca4 + lane] = (lane == 1) ? (DL_END << 16) : 0;
    for (int sl = 0; sl < nslot; ++sl) {
        lu32 *p = V3_L(lane + 1, sl);
        p[0] = 0; p[1] = DL_END << 16;
        if (KIND == 2) { p[ca4] = 0; p[ca4 + 1] = DL_END << 16; }
    }
    if (lane < nslot) {
        lu32 *p = V3_L(0, lane);
        p[0] = 0; p[1] = DL_END << 16;
        if (KIND == 2) { p[ca4] = 0; p[ca4 + 1] = DL_END << 16; }
    }
    // ---- static lists of the strip's rows / the block's columns: contiguous pool ranges -> LDS --------
    CellLists<LList> L;
    constexpr int NN = NA > 0 ? NA : 2;
    int a_sg[NN], a_tg[NN];
    double a_sf[NN], a_tf[NN], a_rhf = 0;
    bool a_rh = false;
    if (NA > 0) {
        rl_load(a_sg, a_sf, a, 0, m, row_ok); rl_load(a_tg, a_tf, a, 1, m, row_ok);
        if (row_ok) {                                      // the r view's head entry, if any (glen 0; the t entries follow with glen + 1)
            const int o = a.off[2][m + 1], lt_ = a.off[1][m + 2] - a.off[1][m + 1] - 1;
            if (a.off[2][m + 2] - o - 1 > lt_) { a_rh = true; a_rhf = a.freq[2][o]; }
        }
        L.as.glen = (li32 *) (lds + LO.aglen); L.as.freq = (lf64 *) (lds + LO.afreq);
        L.at = L.ar = L.bs = L.bt = L.br = L.as;
    } else {
        a_sg[0] = a_sg[1] = a_tg[0] = a_tg[1] = -1; a_sf[0] = a_sf[1] = a_tf[0] = a_tf[1] = 0;      // (the register lists are not used)
        li32 *ag = (li32 *) (lds + LO.aglen);
        lf64 *af = (lf64 *) (lds + LO.afreq);
        int acc = 0;
        LList lv[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int o0 = a.off[v][m0 + 1], cnt = a.off[v][mend + 1] - o0;
            const int *sg = a.glen[v] + o0;
            const double *sf = a.freq[v] + o0;
            for (int k = lane; k < cnt; k += 64) { ag[acc + k] = sg[k]; af[acc + k] = sf[k]; }
            const int mine = acc + (row_ok ? a.off[v][m + 1] - o0 : 0);
            lv[v].glen = ag + mine; lv[v].freq = af + mine;
            acc += cnt;
        }
        L.as = lv[0]; L.at = lv[1]; L.ar = lv[2];
        L.bs = L.bt = L.br = L.as;
    }
    const ARegs<NN> A = {a_sg, a_sf, a_tg, a_tf, a_rhf, a_rh};
    li32 *const boff = (li32 *) (lds + LO.boff);
    li32 *const bg = (li32 *) (lds + LO.bglen);
    lf64 *const bf = (lf64 *) (lds + LO.bfreq);
    const int bcols = c1 - c0;
    if (KIND == 2) {
        int acc = 0;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int o0 = b.off[v][c0 + 1], cnt = b.off[v][c1 + 1] - o0;
            const int *sg = b.glen[v] + o0;
            const double *sf = b.freq[v] + o0;
            for (int k = lane; k < cnt; k += 64) { bg[acc + k] = sg[k]; bf[acc + k] = sf[k]; }
            for (int j = lane; j < bcols; j += 64) boff[v * C + j] = acc + b.off[v][c0 + 1 + j] - o0;
            acc += cnt;
        }
    }
    // ---- the records this row starts from ------------------------------------------------------------
    RS oH = rs_black(), oG = rs_black(), oG2 = rs_black(), oF = rs_black(), oF2 = rs_black();
    {
        const bool cont = row_ok && c0 - 1 >= nlo && c0 - 1 < nhi;         // continues from the block on the left
        const unsigned *src = 0;
        if (cont) src = cbH + (size_t) (m - a.left) * ndw;                 // corner (m+1, c0)
        else if (row_ok && c0 == b.left && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left)
            src = colH + (size_t) (m + 1 - a.left) * ndw;                  // left boundary corner (m+1, b.left)
        if (src) { lu32 *p = V3_L(lane + 1, SLOT_H(c0)); v3_rec_load(src, capa, capb, oH, p, p + ca4); }
        if (cont && lo < hi) {
            lu32 *p = V3_L(lane + 1, SLOT_F);
            v3_rec_load(cbF + (size_t) (m - a.left) * ndw, capa, capb, oF, p, p + ca4);
            if (NOLL3) { p = V3_L(lane + 1, SLOT_F2); v3_rec_load(cbF2 + (size_t) (m - a.left) * ndw, capa, capb, oF2, p, p + ca4); }
        }
    }
    // ---- staging row: records of the strip above for lane 0's columns, one dword per lane --------------
    // corner (m0, col): the previous strip's last row (top boundary chain for strip 0), or the left chain
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < ndw) {
            const unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * ndw : rowHp + (size_t) col * ndw;
            rh = G2G_XLD(s + lane);
            if (wantG) { rg = G2G_XLD(rowGp + (size_t) col * ndw + lane); if (NOLL3) rg2 = G2G_XLD(rowG2p + (size_t) col * ndw + lane); }
        }
    };
    auto stage_put = [&](int slot, int sid, unsigned v) {
        const int j = lane - 4;
        if (lane < 4) stsc[sid * 4 + lane] = v;
        else if (j < capa) V3_L(0, slot)[j] = v;
        else if (j < capa + capb) V3_L(0, slot)[ca4 + j - capa] = v;
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < ndw) {
            stage_put(SLOT_H(col), SLOT_H(col), rh);
            if (wantG) { stage_put(SLOT_G(col), 3 + (col & 1), rg); if (NOLL3) stage_put(SLOT_G2(col), 5 + (col & 1), rg2); }
        }
    };
    int avail = prog_up ? 0 : 0x7fffffff;                    // corner columns of the strip above known to be final
    const int penc = (pgen & 0x7FF) << 20;
    auto need = [&](const int col) {                       // wave-uniform: every lane polls, nobody branches alone
        const int want = penc | (col < 0xFFFFF ? col : 0xFFFFF);
        if (prog_up && want > avail) {
            avail = g2g_wait_ge(prog_up, want, dbg, failp, ti);
            G2G_ACQUIRE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto publish = [&](const int col) {                    // corners <= col of this strip's last row are in HBM
        if (prog_self) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2G_RELEASE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(prog_self + G2G_DIAG + 8, col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (as the v2 strips: the wave's last publish, for the time-out report)
            G2G_POST(prog_self, penc | (col < 0 ? 0 : col < 0xFFFFF ? col : 0xFFFFF));
        }
    };
    if (lane < 28) stsc[lane] = 0;
    team_sync();
    need(cbase + 1 <= c1 ? cbase + 1 : cbase);
    if (prog_self) {                                       // where this strip runs: for the time-out report of whoever waits for it (g2g_wait_ge)
        __hip_atomic_store(prog_self + G2G_DIAG + 3, (int) __builtin_amdgcn_s_getreg((31 << 11) | 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(prog_self + G2G_DIAG + 4, 0x100 | ((int) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        unsigned rh = 0, rg = 0, rg2 = 0;
        stage_load(cbase, false, rh, rg, rg2);
        stage_store(cbase, false, rh, rg, rg2);
        if (cbase + 1 <= c1) {
            stage_load(cbase + 1, vert0, rh, rg, rg2);
            stage_store(cbase + 1, vert0, rh, rg, rg2);
        }
    }
    // per-row constants and one-step-ahead register pipelines (column score, b's column thickness)
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    const bool own_sim = prog_self != 0 && simscr != 0;    // sweep mode: the strip makes its column scores block by block
    const double *simrow = (row_ok && !own_sim) ? P.v2_sim + P.v2_rowoff[m - a.left] - nlo : 0;
    SimBlk SB; SB.buf = (GLBV3 double *) simscr; SB.cbase = cbase;
    // sim31 (a profile row against ONE residue of b, maln.h:168): the score of a cell is a single entry of the row's profile,
    // picked by b's residue -- read it where it is (the row's 200 bytes stay in cache along the row) instead of writing a block
    // of scores to scratch and reading it back (16 bytes of fabric traffic per cell)
    const bool sim_direct = own_sim && P.sim2_kind == 31 && b.many == 1 && a.pseq != 0;
    const double *sim_arow = (sim_direct && row_ok) ? vss_at(a, m) + a.felm : 0;
    const uint8_t *sim_bres = b.seq + 1;                   // residue of column n at sim_bres[n]
    if (own_sim && !sim_direct) { simblk_fill(P, SB, 0, m0, lane); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    double sim_cur = 0, bc_cur = 0;
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    team_sync();
    // Software pipeline of the HBM traffic.  Whatever a step loads (next column's score, the strip above's records
    // two columns ahead) is consumed at the TOP of the next step, and whatever a step produces for HBM (trace byte,
    // the last row's records) is stored at the top of the next step too, right after that consumption: the only
    // vmcnt wait of a step then covers operations that have had a whole step to complete.
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;                                  // staging registers hold column n0 + 1
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;    // the previous step's trace byte
    const int ull = __builtin_amdgcn_readfirstlane(llast);
    int lhi = m0 + llast + P.up + 1; if (lhi > b.right) lhi = b.right; if (lhi > c1) lhi = c1;
    int llo = m0 + llast + P.lw; if (llo < b.left) llo = b.left; if (llo < c0) llo = c0;
    // strip boundary: the last row's newest corner goes to HBM for the strip below, one dword per lane
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
                    const lu32 *p = V3_L(llast + 1, slot);
                    v = (j < capa) ? p[j] : (j < capa + capb) ? p[ca4 + j - capa] : 0;
                }
                unsigned *dst = (x == 0) ? rowHc : (x == 1) ? rowGc : rowG2c;
                if (lane < ndw) G2G_XST(dst + (size_t) col * ndw + lane, v);
            }
        }
    };
    for (int s = 0; s < nsteps; ++s) {
        const int n = cbase + s - lane;
        const int n0 = cbase + s;                          // lane 0's column
        const bool active = row_ok && n >= lo && n < hi;
        // -- top of the step: consume last step's loads, issue last step's stores
        if (st_prev) stage_store(n0 + 1, vert0, st_h, st_g, st_g2);
        if (p_act) P.trace[p_tri] = (uint8_t) p_trb;
        if (wr_rows && s > 0) flush_rows(n0 - 1 - llast);
        G2G_HB_STEP(prog_self, 0, s)                       // (places, v3: 1 top of the step, 5 behind publish / score block, 6 loads issued, 7 cell done)
        G2G_HB(prog_self, 0, 1)
        if (prog_self && s > 0 && (s & (pint - 1)) == 0) publish(n0 - llast);
        if (own_sim && !sim_direct && (s & 63) == 0) { simblk_fill(P, SB, (s >> 6) + 1, m0, lane); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        G2G_HB(prog_self, 0, 5)
        // -- hand-over from the row above: what lane t-1 produced one step ago is my upper neighbour, what it
        // produced two steps ago (= my previous upper neighbour) is my diagonal neighbour
        hd = hu;
        hu = rs_up(oH); gu = rs_up(oG);
        if (NOLL3) g2u = rs_up(oG2);
        {
            const lu32 *q = stsc + SLOT_H(n0) * 4;
            RS t; t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hd = rs_sel(lane == 0, t, hd);
            q = stsc + SLOT_H(n0 + 1) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            hu = rs_sel(lane == 0, t, hu);
            q = stsc + (3 + ((n0 + 1) & 1)) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
            gu = rs_sel(lane == 0, t, gu);
            if (NOLL3) {
                q = stsc + (5 + ((n0 + 1) & 1)) * 4;
                t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = (int) q[3];
                g2u = rs_sel(lane == 0, t, g2u);
            }
        }
        // -- loads for the next step: next column's score/thickness; the strip above's records two columns ahead
        double sim_nx = 0, bc_nx = 0;
        if (active) {
            if (!have) { sim_cur = sim_direct ? sim_arow[sim_bres[n]] : own_sim ? (double) *simblk_at(SB, lane, n) : simrow[n]; bc_cur = thk_at(b, n)[0]; }
            if (n + 1 < hi) { sim_nx = sim_direct ? sim_arow[sim_bres[n + 1]] : own_sim ? (double) *simblk_at(SB, lane, n + 1) : simrow[n + 1]; bc_nx = thk_at(b, n + 1)[0]; }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) { need(n0 + 2); stage_load(n0 + 2, vert0, st_h, st_g, st_g2); }
        RS myH = oH, myG = oG, myG2 = oG2;                 // (the produced records of this step)
        if (active) {
            const bool do_hori = n > b.left;
            if (KIND == 2) {
                const int j = n - c0;
                const int o0 = boff[j], o1 = boff[C + j], o2 = boff[2 * C + j];
                L.bs.glen = bg + o0; L.bs.freq = bf + o0;
                L.bt.glen = bg + o1; L.bt.freq = bf + o1;
                L.br.glen = bg + o2; L.br.freq = bf + o2;
            }
            const bool up_in = do_vert && (n - (m - 1) <= P.up);          // cell (m-1, n) exists
            const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
            const RS bk = rs_black();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const lu32 *hdl = V3_L(lane, SLOT_H(n));
            const lu32 *hul = up_in ? V3_L(lane, SLOT_H(n + 1)) : blk;
            const lu32 *gul = up_in ? V3_L(lane, SLOT_G(n + 1)) : blk;
            const lu32 *g2ul = (NOLL3 && up_in) ? V3_L(lane, SLOT_G2(n + 1)) : blk;
            const lu32 *hll = left_in ? V3_L(lane + 1, SLOT_H(n)) : blk;
            const lu32 *fll = left_in ? V3_L(lane + 1, SLOT_F) : blk;
            const lu32 *f2ll = (NOLL3 && left_in) ? V3_L(lane + 1, SLOT_F2) : blk;
            lu32 *dh = V3_L(lane + 1, SLOT_H(n + 1));
            lu32 *dg = V3_L(lane + 1, SLOT_G(n + 1));
            lu32 *dg2 = V3_L(lane + 1, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
            lu32 *df = V3_L(lane + 1, SLOT_F);
            lu32 *df2 = V3_L(lane + 1, NOLL3 ? SLOT_F2 : SLOT_F);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            int trb = 0;
            if (KIND == 1 && NA > 0)
                v3_cell_hf<NOLL3, (NA > 0 ? NA : 2)>(P, A, (lu32 *) (lds + LO.sink) + lane, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                                 dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb);
            else
                v3_cell<KIND, NOLL3>(P, ca4, L, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                                 dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb);
            const int d = m + n;
            int mlo, mhi;
            diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (d - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; have = (n + 1 < hi);
            // block boundary: this row's corner and F records for the block on the right
            if (n == c1 - 1 && c1 < b.right) {
                v3_rec_store(cbH + (size_t) (m - a.left) * ndw, capa, capb, myH, dh, dh + ca4);
                v3_rec_store(cbF + (size_t) (m - a.left) * ndw, capa, capb, oF, df, df + ca4);
                if (NOLL3) v3_rec_store(cbF2 + (size_t) (m - a.left) * ndw, capa, capb, oF2, df2, df2 + ca4);
            }
            if (m == a.right - 1 && n == b.right - 1) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        team_sync();
    }
    if (p_act) P.trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - llast);
    publish(0xFFFFF);
#undef V3_L
}

#define V3_KERNEL(NAME, KIND, N3, NA, WPE)                                                           \
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))      \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, V3Lds LO, int C, int sweep, int pro_off, double *simscr) \
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
        if (T.ti < 0) {           /* a boundary chain (v2_chain_tile, g2g_kernels_v2.hip) */         \
            v2_chain_tile<KIND>(probs[T.prob], (lchar *) g2g_lds, T.ti, done + T.self, gen, pro_off); \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        int *failp = done + done[G2G_HDR + 2] + T.prob;                                                      \
        if (threadIdx.x == 0) s_vals[0] = g2g_dp_failed(failp) ? 1 : 0;      /* (one reader: the branch must be uniform) */ \
        __syncthreads();                                                                            \
        const int dp_dead = s_vals[0];                                                              \
        __syncthreads();                                                                            \
        if (dp_dead) {                    /* this DP lost a wait: its strips are skipped, dependents released */ \
            if (threadIdx.x == 0) G2G_POST(done + T.self, sweep ? (((gen & 0x7FF) << 20) | 0xFFFFF) : gen); \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        const int *pl = (sweep && T.dep_left >= 0) ? done + T.dep_left : (const int *) 0;            \
        /* sweep: strips as a pipeline on progress counters; else tiles on completion flags.  ONE call site of the   \
           tile function, or it is not inlined and its frame lands in scratch memory */                             \
        const int *pu = (sweep && T.dep_up >= 0) ? done + T.dep_up : (const int *) 0;                \
        int *ps = sweep ? done + T.self : (int *) 0;                                                \
        if (!sweep) {                                                                               \
            if (T.dep_up >= 0) v2_wait_flag(done + T.dep_up, gen, done + G2G_HDR, t, failp);             \
            if (T.dep_left >= 0) v2_wait_flag(done + T.dep_left, gen, done + G2G_HDR, t, failp);         \
            if (T.dep_diag >= 0) v2_wait_flag(done + T.dep_diag, gen, done + G2G_HDR, t, failp);         \
            if (T.dep_war >= 0) v2_wait_flag(done + T.dep_war, gen, done + G2G_HDR, t, failp);           \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
        }                                                                                           \
        __syncthreads();                                                                            \
        v3_tile<KIND, N3, NA>(probs[T.prob], (lchar *) g2g_lds, LO, T.ti, sweep ? 0 : T.tj, T.nsteps, C, pu, ps, done + G2G_HDR, gen, sweep, pl, \
                              (sweep && simscr) ? simscr + (size_t) blockIdx.x * G2G_SIMBLK_STRIDE : (double *) 0, failp); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
        if (!sweep) {                                                                               \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
            G2G_POST(done + T.self, gen);     \
        }                                                                                           \
    }                                                                                               \
}
#ifndef G2G_V3R_WPE
#define G2G_V3R_WPE 2
#endif
#ifndef G2G_V3_NA
#define G2G_V3_NA 16                 // register-resident static lists: up to this many entries (incl. terminator)
#endif
#ifdef G2G_TU_V3
V3_KERNEL(g2g_v3_hf2, 1, false, 0, 2)
V3_KERNEL(g2g_v3_hf3, 1, true, 0, 2)
V3_KERNEL(g2g_v3r_hf2, 1, false, G2G_V3_NA, G2G_V3R_WPE)
V3_KERNEL(g2g_v3r_hf3, 1, true, G2G_V3_NA, 1)
#else
#define V3_KERNEL_DECL(NAME) extern "C" __global__ void NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, V3Lds LO, int C, int sweep, int pro_off, double *simscr);
V3_KERNEL_DECL(g2g_v3_hf2) V3_KERNEL_DECL(g2g_v3_hf3) V3_KERNEL_DECL(g2g_v3r_hf2) V3_KERNEL_DECL(g2g_v3r_hf3)
#endif
