// g2g_kernels_v6.hip -- forward kernel of the full gap-profile engine (_pf; Noll 2/3): ONE LANE PER CELL with the list
// merges in RANK form instead of two-pointer walks.
//
// Same recurrence and arithmetic order as every other generation (Fwd2c<DPunit_pf>::forwardB, reference src/fwd2c.h:359-482,
// gapopen/update src/fwd2c.cc:203-233, newgap/newdelta/incdelta src/gfreq.cc:507-521,570-605).  What is new is how a cell's
// six (eight with Noll 3) gap-open costs are evaluated.  newgap(cf, dlc, df, dld) walks df and keeps a pointer into cf that
// moves to the first entry whose STRETCHED length i = glen + nins(dlc, glen) reaches the stretched length j of the current
// df entry; both sequences are non-decreasing, so that pointer is simply the lower bound of j in {i}:
//        g = sum over d (in list order) of  cf.freq[ first c with i_c >= j_d ] * df.freq[d]       (0 once cf is exhausted)
// Every term is independent of the others; only the ORDER of the additions is part of the contract.  On a 64-wide wave the
// two-pointer walk is a divergent loop with a dependent LDS read per step (v2: 93 VALU + 125 SALU wave instructions per cell);
// here every loop runs over a list index that is the same in all lanes:
//   * row side (a): lane t owns row m0 + t of a 64-row strip; the row's three static lists live in REGISTERS (never change
//     inside a strip), loops over them are fully unrolled and leave at the wave's longest list;
//   * column side (b): the columns' static lists stream through a small LDS RING indexed by pool position (lists of
//     consecutive columns are contiguous in the profile pools), refilled 16 columns at a time; loops over them are rolled
//     and read entry k of every lane's own column;
//   * "first c with i_c >= j" is a select chain walked from the last entry down (the lowest hit is written last), no
//     per-lane pointer, no branch;
//   * dynamic lists are read through register heads that are sanitised behind their terminator.
// Strips of a DP run as a pipeline on progress counters exactly as in g2g_kernels_v3.hip (sweep mode); record scalars travel
// down the lanes by DPP; the strip boundary and the boundary chains use the v2 record image in HBM.
//
// Round 3: LDS decides how many strips a CU holds, and the rings of dynamic lists were most of it (a slot per list of
// hetero + 1 dwords: 31-45 KB per strip).  Now a list keeps only its INLINE part in LDS and the rest in an HBM twin (LS6 below):
// 26-30 KB per strip, so EVERY _pf DP of a sweep runs at four strips per CU -- those that ran three per CU or fell back to the
// 8-lanes-per-cell kernel before included.  (A second form of the cell -- row lists in LDS, rolled loops, 253 registers, two
// waves per SIMD -- was built and measured: slower per cell at equal occupancy, and LDS would not let the second wave in; it is
// in the history of this file.)  The ring refill no longer picks a descriptor field by a per-lane index, which had put the
// register copy of the whole descriptor into scratch memory (1.1 KB per lane): the kernel has no private segment now.
#include <hip/hip_runtime.h>

// HBM pointers of the sweep carry their address space: a generic pointer compiles to flat_load/flat_store, which also
// occupy the LDS counter (every wait for an LDS read would then wait for the global loads in flight as well)
#define GLB __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const GLB T *glb(const T *p) { return (const GLB T *) p; }
template <class T> __device__ __forceinline__ GLB T *glbw(T *p) { return (GLB T *) p; }

// -DG2G_V6_STAMP: cycle shares of the phases of a step (diagnostics; s_memtime deltas summed per strip, then atomically)
#ifdef G2G_V6_STAMP
__device__ unsigned long long g2g_v6_stamp_acc[16];
#define V6_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; }
#define V6_STAMP_ARGS , unsigned long long (&st_acc)[12], unsigned long long &st_t
#define V6_STAMP_PASS , st_acc, st_t
#else
#define V6_STAMP(k)
#define V6_STAMP_ARGS
#define V6_STAMP_PASS
#endif

// byte offsets; ringk / ringf / rs: per view (s, t, r) the key and freq arrays of the column-list ring and its entries (power of 2)
struct V6Lds { int rows, black, stsc, svals, sink, total; int ringk[3], ringf[3], rs[3]; };

#define V6_FEED 16                      // columns per ring refill
#define V6_AHEAD 32                     // a refill reaches this many columns beyond lane 0's
#define V6_WINDOW (64 + V6_AHEAD + 4)   // columns whose lists must fit the ring together
#define V6_RHCOLS 128                   // columns of the per-column array of r-list heads (a power of 2 >= V6_WINDOW)

// ---- where the dynamic lists of a strip live ---------------------------------------------------------------
// A record's list may be as long as its side has gap states (hetero + 1 dwords), but 99.98 % of the lists of a refinement sweep
// have seven entries or fewer.  LDS therefore holds only the first I dwords of a list (its INLINE part: I = 8, or 4 when the
// side's capacity is 4 -- then nothing else exists); the rest lives in a per-workgroup image in HBM (the list's TWIN: slot
// offset x 2, so an inline part of I dwords owns 2 I dwords there; capacities above 16 take I = 16, 32).  Only the code that
// already handled "beyond the register head" ever gets there -- the scans, the stores of a long newdelta result, the strip
// hand-over of a long list -- always behind a wave-uniform test.  A wave reads back only what it wrote itself, after a
// s_waitcnt vmcnt(0).  LDS per strip drops from 31-45 KB (and more) to 26-30 KB.
G2G_HD inline int v6_inline_dw(int cap4) { return cap4 <= 4 ? 4 : cap4 <= 16 ? 8 : cap4 <= 32 ? 16 : cap4 <= 64 ? 32 : cap4; }
struct LS6 { const lu32 *rows0; GLB unsigned *tw; int ia, ib; };          // first dword of the rows, the twin image, inline dwords per side
__device__ __forceinline__ GLB unsigned *ls6_twin(const LS6 &W, const lu32 *p, const int k) { return W.tw + 2 * (int) (p - W.rows0) + k; }
__device__ __forceinline__ unsigned ls6_rd(const LS6 &W, const lu32 *p, const int k, const int I) { return k < I ? p[k] : *ls6_twin(W, p, k); }
__device__ __forceinline__ void ls6_wr(const LS6 &W, lu32 *p, const int k, const int I, const unsigned v)
{
    if (k < I) p[k] = v;
    else { *ls6_twin(W, p, k) = v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}

// ---- register heads of dynamic lists ---------------------------------------------------------------------
// {glen,nins} packed 16+16, ascending in both, terminator 0xFFFF0000; the first entry is never the terminator.  Behind the
// terminator memory is stale: the head replaces it by terminators, so a lookup is a chain of unsigned compares of the
// packed key (g << 16 | 0xFFFF) with no validity tests.  NE leading entries are held (5: ds_read_b128 + b64; 8: two b128);
// x = entry NE, sanitised the same way: x < T says the list is longer than the head.
// The cell exists in three instances (v6_cell_pf): NE = 5 and NE = 8 WITHOUT any per-lookup test -- they leave at once
// when some lane's list is longer than the head (0.012 % of the lists have six entries or more, 4e-6 nine or more) -- and
// NE = 8 with an inline scan of LDS behind the head (SCAN), which is correct for any length.
#define V6_UNROLL _Pragma("clang loop unroll(full)")
template <int NE> struct DH { unsigned e[NE]; unsigned x; const lu32 *p; };
// (I: inline dwords of the list's side.  An eight-entry head of a list whose inline part is eight dwords finds entry 8 in the
// twin -- looked at only when entry 7 is not the terminator, behind a wave-uniform test)
template <int NE>
__device__ __forceinline__ DH<NE> dh_load6(const lu32 *p, const LS6 &W, const int I)
{
    const unsigned T = DL_END << 16;
    DH<NE> h;
    const v4u32 v = *(const LDS v4u32 *) p;
    h.e[0] = v.x; h.e[1] = v.y; h.e[2] = v.z; h.e[3] = v.w;
    if (NE == 8) {
        const v4u32 w = *(const LDS v4u32 *) (p + 4); h.e[4] = w.x; h.e[5] = w.y; h.e[6] = w.z; h.e[7] = w.w;
        h.x = T;
        if (I > 8) h.x = p[8];
        else { const bool lg = w.w < T && w.z < T; if (__ballot(lg)) { if (lg) h.x = ls6_rd(W, p, 8, I); } }
    }
    else if (NE == 5) { const unsigned long long w = *(const LDS unsigned long long *) (p + 4); h.e[4] = (unsigned) w; h.x = (unsigned) (w >> 32); }
    else h.x = p[NE];
    V6_UNROLL
    for (int k = 2; k < NE; ++k) h.e[k] = (h.e[k - 1] >= T) ? T : h.e[k];
    h.x = (h.e[NE - 1] >= T) ? T : h.x;
    h.p = p;
    return h;
}
// (a terminator's glen = -1 is looked up as 0: its result is never used)
__device__ __forceinline__ unsigned v6_key(const int g) { return ((unsigned) (g < 0 ? 0 : g) << 16) | 0xFFFFu; }
// the entry that governs static gap length g (GapLenSD, gfreq.h:67): the largest entry <= key
template <int NE, bool SCAN>
__device__ __forceinline__ unsigned dh_ent(const unsigned key, const DH<NE> &h, const LS6 &W, const int I)
{
    unsigned e = h.e[0];
    V6_UNROLL
    for (int k = 1; k < NE; ++k) e = key >= h.e[k] ? h.e[k] : e;
    if (SCAN) {
        const bool more = key >= h.x;
        if (__ballot(more)) {                               // more than NE entries at or below g
            if (more) {
                int k = NE;
                e = h.x;
                for (;;) {                                  // e = entry k (at or below the key); stop when entry k + 1 is above it
                    if (k + 1 >= DL_GUARD) break;
                    const unsigned nx = ls6_rd(W, h.p, k + 1, I);
                    if (key < nx) break;
                    e = nx; ++k;
                }
            }
        }
    }
    return e;
}
// Stretched lengths are compared, never used as numbers: they stay in key form, ((glen + nins) << 16) | 0xFFFF, which is
// one shift-add on the matched entry (no overflow: glen + nins < 65536 on this path)
template <int NE, bool SCAN>
__device__ __forceinline__ unsigned dh_stretch(const unsigned key, const DH<NE> &h, const LS6 &W, const int I) { return (dh_ent<NE, SCAN>(key, h, W, I) << 16) + key; }
template <int NE>
__device__ __forceinline__ DH<NE> dh_sel6(const bool c, const DH<NE> &x, const DH<NE> &y)
{
    DH<NE> h;
    V6_UNROLL
    for (int k = 0; k < NE; ++k) h.e[k] = c ? x.e[k] : y.e[k];
    h.x = c ? x.x : y.x; h.p = c ? x.p : y.p;
    return h;
}
// incdelta(dlt, dln, 1), gfreq.cc:598-605, from a head: the leading entries go out as 16-byte stores (what lands behind
// the terminator is stale by definition); d2: second destination or the lane's 16-byte sink
template <int NE, bool SCAN>
__device__ __forceinline__ void v6_incdelta(const bool on, const DH<NE> &h, lu32 *d1, lu32 *d2, lu32 *sink16, const LS6 &W, const int I)
{
    const unsigned T = DL_END << 16;
    v4u32 w;
    w.x = h.e[0] + 1; w.y = h.e[1] >= T ? T : h.e[1] + 1; w.z = h.e[2] >= T ? T : h.e[2] + 1; w.w = h.e[3] >= T ? T : h.e[3] + 1;
    *(LDS v4u32 *) (on ? d1 : sink16) = w;
    *(LDS v4u32 *) ((on && d2) ? d2 : sink16) = w;
    if (NE == 5) {                                          // entry 4 and the terminator behind it (a list of this instance has five entries at most)
        const bool m4 = on && h.e[3] < T;
        if (__ballot(m4)) {
            const unsigned long long w45 = ((unsigned long long) T << 32) | (h.e[4] >= T ? T : h.e[4] + 1);
            *(LDS unsigned long long *) (m4 ? d1 + 4 : sink16) = w45;
            *(LDS unsigned long long *) ((m4 && d2) ? d2 + 4 : sink16) = w45;
        }
    }
    if (NE == 6) {                                          // entries 4 and 5 (the terminator is one of them when the list has four or five
        const bool m4 = on && h.e[3] < T;                   // entries; behind a sixth entry the scan below goes on)
        if (__ballot(m4)) {
            const unsigned long long w45 = ((unsigned long long) (h.e[5] >= T ? T : h.e[5] + 1) << 32) | (h.e[4] >= T ? T : h.e[4] + 1);
            *(LDS unsigned long long *) (m4 ? d1 + 4 : sink16) = w45;
            *(LDS unsigned long long *) ((m4 && d2) ? d2 + 4 : sink16) = w45;
        }
    }
    if (NE == 8) {                                          // (entry 4 is only looked at when entry 3 is not the terminator)
        const bool m4 = on && h.e[3] < T;
        if (__ballot(m4)) {
            w.x = h.e[4] >= T ? T : h.e[4] + 1; w.y = h.e[5] >= T ? T : h.e[5] + 1; w.z = h.e[6] >= T ? T : h.e[6] + 1; w.w = h.e[7] >= T ? T : h.e[7] + 1;
            *(LDS v4u32 *) (m4 ? d1 + 4 : sink16) = w;
            *(LDS v4u32 *) ((m4 && d2) ? d2 + 4 : sink16) = w;
        }
        if (!SCAN) {                                        // eight entries: the terminator sits behind the head
            const bool m8 = on && h.e[7] < T;
            if (__ballot(m8)) { if (m8) { ls6_wr(W, d1, 8, I, T); if (d2) ls6_wr(W, d2, 8, I, T); } }
        }
    }
    if (SCAN) {
        const bool more = on && h.e[NE - 1] < T;
        if (__ballot(more)) {
            if (more) {
                for (int k = NE; k < DL_GUARD; ++k) {
                    unsigned e = ls6_rd(W, h.p, k, I);
                    if (e < T) e += 1;
                    ls6_wr(W, d1, k, I, e); if (d2) ls6_wr(W, d2, k, I, e);
                    if (e >= T) break;
                }
            }
        }
    }
}
// newdelta (gfreq.cc:570-587) as a step function over the static entries of a t list
struct ND6 { int kd; unsigned tg, tn; bool on; };
// (LONG: some lane of the wave is about to store at or beyond the inline part)
template <int NE, bool SCAN, bool LONG>
__device__ __forceinline__ void nd6_step(ND6 &s, const DH<NE> &h, const int g, const unsigned key, lu32 *d1, lu32 *d2, lu32 *sink, const LS6 &W, const int I)
{
    const unsigned sn = dh_ent<NE, SCAN>(key, h, W, I) & 0xFFFFu;
    const bool emit = s.on && g >= 0 && sn > s.tn;
    const unsigned e = (s.tg << 16) | s.tn;
    if (LONG) {
        if (emit) { ls6_wr(W, d1, s.kd, I, e); if (d2) ls6_wr(W, d2, s.kd, I, e); }
    } else {
        *(emit ? d1 + s.kd : sink) = e;
        *((emit && d2) ? d2 + s.kd : sink) = e;
    }
    s.kd += emit ? 1 : 0;
    s.tn = emit ? sn : s.tn;
    s.tg = emit ? (unsigned) (g + 1) : s.tg;
    s.on = s.on && g >= 0;
}
template <int NE>
__device__ __forceinline__ void nd6_fin(const ND6 &s, const bool was_on, lu32 *d1, lu32 *d2, lu32 *sink, const LS6 &W, const int I)
{
    const unsigned e = (s.tg << 16) | s.tn;
    if (NE > 5 && __ballot(was_on && s.kd + 1 >= I)) {                // the last entry or the terminator of some lane lands beyond the inline part
        if (was_on) {
            ls6_wr(W, d1, s.kd, I, e); ls6_wr(W, d1, s.kd + 1, I, DL_END << 16);
            if (d2) { ls6_wr(W, d2, s.kd, I, e); ls6_wr(W, d2, s.kd + 1, I, DL_END << 16); }
        }
        return;
    }
    *(was_on ? d1 + s.kd : sink) = e;
    *(was_on ? d1 + s.kd + 1 : sink) = DL_END << 16;
    *((was_on && d2) ? d2 + s.kd : sink) = e;
    *((was_on && d2) ? d2 + s.kd + 1 : sink) = DL_END << 16;
}

// ---- the ring of b's static lists ------------------------------------------------------------------------
// Per view two arrays: lookup keys (u32, (glen << 16) | 0xFFFF) and freq (f64) -- 12 bytes per entry, terminators left out.
// An entry's ring index is its COMPACT pool position & mask: pool position minus the terminators in front of it, i.e.
// off[v][n + 1] - (n + 1) + k for entry k of column n (every position contributes exactly one terminator).  A lane knows its
// column's lists by (start, length) taken from the offset tables.
struct SE6 { int g; unsigned key; double f; };
struct Ring6 { const lu32 *k; const lf64 *f; int mask; };
__device__ __forceinline__ SE6 se6_read(const Ring6 &r, const int idx)
{
    const int i = idx & r.mask;
    SE6 e; e.key = r.k[i]; e.f = r.f[i]; e.g = (int) (e.key >> 16); return e;
}

// the row's static lists in registers: glen as lookup key ((g << 16) | 0xFFFF; slots behind the list: key 0xFFFF, freq 0) and freq;
// ls / lt / lr: the lane's list lengths
// The r view is not held: r = [head {glen 0, freq rhf}, if present] + the t list with glen + 1 (DevSide::r_from_t).
template <int N> struct A6 { const unsigned (&sk)[N]; const double (&sf)[N]; const unsigned (&tk)[N]; const double (&tf)[N]; double rhf; int ls, lt; };
// the lane's column: rings of its s and t lists, list starts (compact), lengths.  The r view has no ring: r = [head {glen 0, freq rhf}, if the
// column has one (lenr > lent)] + the t entries with glen + 1 (DevSide::r_from_t, checked by the host); rhf comes from a per-column array
struct B6 { Ring6 rs, rt; int os, ot, lens, lent, lenr; double rhf; };

// one "X" merge: cf = the column's s list (ring, stretched by dlb of the record), df = a row list in registers (stretched by
// dla): newgap(b.s, dlb, a.t|a.r, dla).  RV: df is the r view = an optional head entry {glen 0, freq hf} followed by the t
// entries with glen + 1 (hf = 0: no head; its term is then +0).  lmax: the wave's longest s list.  Slots behind the row's
// list hold key 0xFFFF / freq 0: their terms are +0 as well, so validity needs no test.
template <int N, int NE, bool SCAN, bool RV>
__device__ __forceinline__ double v6_xmerge(const DH<NE> &ha, const DH<NE> &hb, const unsigned (&dk)[N], const double (&dfq)[N], const double hf,
                                            const int TA, const B6 &B, const int lmax, const LS6 &W)
{
    unsigned j[N], jh = 0;
    double S[N], Sh = 0;
    if (RV) jh = dh_stretch<NE, SCAN>(0xFFFFu, ha, W, W.ia);
    V6_UNROLL
    for (int d = 0; d < N; ++d) {
        if (d < TA) {
            j[d] = dh_stretch<NE, SCAN>(dk[d] + (RV ? 0x10000u : 0u), ha, W, W.ia);
            S[d] = 0;
        }
    }
    for (int kk = lmax - 1; kk >= 0; --kk) {
        const bool valid = kk < B.lens;
        const SE6 e = se6_read(B.rs, B.os + (valid ? kk : 0));
        const unsigned i = valid ? dh_stretch<NE, SCAN>(e.key, hb, W, W.ib) : 0u;
        if (RV) Sh = i >= jh ? e.f : Sh;
        V6_UNROLL
        for (int d = 0; d < N; ++d)
            if (d < TA) S[d] = i >= j[d] ? e.f : S[d];
    }
    double g = 0;
    if (RV) g += Sh * hf;
    V6_UNROLL
    for (int d = 0; d < N; ++d)
        if (d < TA) g += S[d] * dfq[d];
    return g;
}

// ---- one cell by one lane ----------------------------------------------------------------------------------
// Returns false -- before anything is stored -- when !SCAN and some lane's list is longer than the heads of this instance.
template <bool NOLL3, int N, int NE, bool SCAN>
__device__ __forceinline__ bool v6_cell_pf(const DevProb &P, const LS6 &W, const A6<N> &A, const int TAs, const int TAt,
    const B6 &B, lu32 *sink, lu32 *sink16,
    const RS &hd, const lu32 *hdl, const RS &hu, const lu32 *hul, const RS &gu, const lu32 *gul,
    const RS &g2u, const lu32 *g2ul, const RS &hl, const lu32 *hll, const RS &fl, const lu32 *fll,
    const RS &f2l, const lu32 *f2ll,
    lu32 *dh, lu32 *dg, lu32 *dg2, lu32 *df, lu32 *df2,
    const bool do_vert, const bool do_hori, const double dab, const double pua, const double pub,
    RS &oH, RS &oG, RS &oG2, RS &oF, RS &oF2, int &trb V6_STAMP_ARGS)
{
    typedef DH<NE> DHn;
    const int ca4 = W.ia;                                   // a record's b-side list follows its a-side list's inline part
    // heads of the dynamic lists of the five (seven) records this cell reads: a side and b side
    const DHn a_hd = dh_load6<NE>(hdl, W, W.ia), b_hd = dh_load6<NE>(hdl + ca4, W, W.ib);
    const DHn a_gu = dh_load6<NE>(gul, W, W.ia), b_gu = dh_load6<NE>(gul + ca4, W, W.ib);
    const DHn a_hu = dh_load6<NE>(hul, W, W.ia), b_hu = dh_load6<NE>(hul + ca4, W, W.ib);
    const DHn a_fl = dh_load6<NE>(fll, W, W.ia), b_fl = dh_load6<NE>(fll + ca4, W, W.ib);
    const DHn a_hl = dh_load6<NE>(hll, W, W.ia), b_hl = dh_load6<NE>(hll + ca4, W, W.ib);
    const DHn a_g2 = dh_load6<NE>(NOLL3 ? g2ul : gul, W, W.ia), b_g2 = dh_load6<NE>((NOLL3 ? g2ul : gul) + ca4, W, W.ib);
    const DHn a_f2 = dh_load6<NE>(NOLL3 ? f2ll : fll, W, W.ia), b_f2 = dh_load6<NE>((NOLL3 ? f2ll : fll) + ca4, W, W.ib);
    if (!SCAN) {
        const unsigned T = DL_END << 16;
        unsigned mn = a_hd.x < b_hd.x ? a_hd.x : b_hd.x;
        mn = a_gu.x < mn ? a_gu.x : mn; mn = b_gu.x < mn ? b_gu.x : mn; mn = a_hu.x < mn ? a_hu.x : mn; mn = b_hu.x < mn ? b_hu.x : mn;
        mn = a_fl.x < mn ? a_fl.x : mn; mn = b_fl.x < mn ? b_fl.x : mn; mn = a_hl.x < mn ? a_hl.x : mn; mn = b_hl.x < mn ? b_hl.x : mn;
        if (NOLL3) { mn = a_g2.x < mn ? a_g2.x : mn; mn = b_g2.x < mn ? b_g2.x : mn; mn = a_f2.x < mn ? a_f2.x : mn; mn = b_f2.x < mn ? b_f2.x : mn; }
        if (__ballot(mn < T)) { V6_STAMP(8) return false; }
    }
    V6_STAMP(1)
    Costs c;
    c.d0 = c.d1 = c.gnpv = c.gopv = c.gnph = c.goph = c.gnpv2 = c.gnph2 = 0;
#ifndef V6_SKIP_Y
    // ---- "Y" merges: cf = the row's s list (registers), df = a column list (ring): diagonal part 0 (b.t, record hd),
    // vertical gnp / gop (b.r, records gu / hu), vertical2 (b.r, g2u) -- gfreq.cc:507-521 in rank form
    {
        unsigned i_hd[N], i_gu[N], i_hu[N], i_g2[N];          // stretched keys
        unsigned m_hd = 0, m_gu = 0, m_hu = 0, m_g2 = 0;
        V6_UNROLL
        for (int k = 0; k < N; ++k) {
            if (k < TAs) {
                // (slots behind the list: key 0xFFFF, freq 0 -- a hit there selects freq 0, which is what an exhausted cf adds)
                const unsigned key = A.sk[k];
                i_hd[k] = dh_stretch<NE, SCAN>(key, a_hd, W, W.ia);
                i_gu[k] = dh_stretch<NE, SCAN>(key, a_gu, W, W.ia);
                i_hu[k] = dh_stretch<NE, SCAN>(key, a_hu, W, W.ia);
                if (NOLL3) i_g2[k] = dh_stretch<NE, SCAN>(key, a_g2, W, W.ia);
                m_hd = i_hd[k] > m_hd ? i_hd[k] : m_hd;
                m_gu = i_gu[k] > m_gu ? i_gu[k] : m_gu;
                m_hu = i_hu[k] > m_hu ? i_hu[k] : m_hu;
                if (NOLL3) m_g2 = i_g2[k] > m_g2 ? i_g2[k] : m_g2;
            }
        }
        double g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        bool l0 = true, l1 = do_vert, l2 = do_vert, l3 = do_vert && NOLL3;
        for (int d = 0; d < DL_GUARD; ++d) {
            if (wave_none(l0 || l1 || l2 || l3)) break;
            const int dr = d - (B.lenr > B.lent ? 1 : 0);                          // entry d of the r view: the head, or t entry d - 1 / d
            const SE6 et = se6_read(B.rt, B.ot + d), ert = se6_read(B.rt, B.ot + (dr < 0 ? 0 : dr));
            SE6 er;
            er.key = dr < 0 ? 0xFFFFu : ert.key + 0x10000u; er.f = dr < 0 ? B.rhf : ert.f; er.g = (int) (er.key >> 16);
            l0 = l0 && d < B.lent;
            const bool lv = d < B.lenr;
            l1 = l1 && lv; l2 = l2 && lv; l3 = l3 && lv;
            const unsigned j0 = dh_stretch<NE, SCAN>(et.key, b_hd, W, W.ib), j1 = dh_stretch<NE, SCAN>(er.key, b_gu, W, W.ib), j2 = dh_stretch<NE, SCAN>(er.key, b_hu, W, W.ib);
            const unsigned j3 = NOLL3 ? dh_stretch<NE, SCAN>(er.key, b_g2, W, W.ib) : 0u;
            l0 = l0 && m_hd >= j0; l1 = l1 && m_gu >= j1; l2 = l2 && m_hu >= j2; l3 = l3 && m_g2 >= j3;       // cf exhausted: break
            double S0 = 0, S1 = 0, S2 = 0, S3 = 0;
            // from the wave's last s entry down to entry 0: a fall-through switch (a guarded unrolled loop is turned into
            // selects over all N entries by the compiler; the jump is wave-uniform)
#define V6_YCH(k) if (N > (k)) { S0 = i_hd[k] >= j0 ? A.sf[k] : S0; S1 = i_gu[k] >= j1 ? A.sf[k] : S1; S2 = i_hu[k] >= j2 ? A.sf[k] : S2; \
                                 if (NOLL3) S3 = i_g2[k] >= j3 ? A.sf[k] : S3; }
            switch (TAs) {
            default: V6_YCH(15) case 15: V6_YCH(14) case 14: V6_YCH(13) case 13: V6_YCH(12) case 12: V6_YCH(11) case 11: V6_YCH(10)
            case 10: V6_YCH(9) case 9: V6_YCH(8) case 8: V6_YCH(7) case 7: V6_YCH(6) case 6: V6_YCH(5) case 5: V6_YCH(4)
            case 4: V6_YCH(3) case 3: V6_YCH(2) case 2: V6_YCH(1) case 1: V6_YCH(0) case 0: ;
            }
#undef V6_YCH
            g0 = l0 ? g0 + S0 * et.f : g0;
            g1 = l1 ? g1 + S1 * er.f : g1;
            g2 = l2 ? g2 + S2 * er.f : g2;
            if (NOLL3) g3 = l3 ? g3 + S3 * er.f : g3;
        }
        c.d0 = g0 * P.basic_gop; c.gnpv = g1 * P.basic_gop; c.gopv = g2 * P.basic_gop; c.gnpv2 = NOLL3 ? g3 * P.basic_gop : 0;
    }
#endif
    V6_STAMP(2)
#ifndef V6_SKIP_X
    // ---- "X" merges: cf = the column's s list, df = a row list: diagonal part 1 (a.t, hd), horizontal gnp / gop (a.r, fl / hl)
    {
        int lmax = 0;
        while (__ballot(B.lens > lmax)) ++lmax;
        c.d1 = v6_xmerge<N, NE, SCAN, false>(a_hd, b_hd, A.tk, A.tf, 0., TAt, B, lmax, W) * P.basic_gop;
        c.gnph = v6_xmerge<N, NE, SCAN, true>(a_fl, b_fl, A.tk, A.tf, A.rhf, TAt, B, lmax, W) * P.basic_gop;
        c.goph = v6_xmerge<N, NE, SCAN, true>(a_hl, b_hl, A.tk, A.tf, A.rhf, TAt, B, lmax, W) * P.basic_gop;
        c.gnph2 = NOLL3 ? v6_xmerge<N, NE, SCAN, true>(a_f2, b_f2, A.tk, A.tf, A.rhf, TAt, B, lmax, W) * P.basic_gop : 0;
    }
#endif
    V6_STAMP(3)
    const Dec d = v3_decide<2, NOLL3>(P, c, hd, hu, gu, g2u, hl, fl, f2l, do_vert, do_hori, dab, pua, pub);
    const int win = d.win;
    V6_STAMP(4)
    // ---- list updates (update(), fwd2c.cc:216-231); the winner's lists are also the new H's -------------------
    lu32 *const nul = (lu32 *) 0;
#ifndef V6_SKIP_ND
    {   // a side: newdelta over a.t for G (G2) and a diagonal H; incdelta for F (F2)
        const DHn h_gs = dh_sel6<NE>(d.g_from_h, a_hu, a_gu), h_gs2 = dh_sel6<NE>(d.g2_from_h, a_hu, a_g2);
        ND6 n_g = {0, 0, 0, do_vert}, n_h = {0, 0, 0, win == 0}, n_g2 = {0, 0, 0, do_vert && NOLL3};
        lu32 *const g_d2 = win == 1 ? dh : nul, *const g2_d2 = win == 2 ? dh : nul;
        // (an instance with five-entry heads only runs on lists of five entries at most: nothing leaves the inline parts there)
        V6_UNROLL
        for (int k = 0; k < N; ++k) {
            if (k < TAt) {
                const int g = k < A.lt ? (int) (A.tk[k] >> 16) : -1;
                if (NE > 5 && __ballot(n_g.kd >= W.ia || n_h.kd >= W.ia || (NOLL3 && n_g2.kd >= W.ia))) {      // a result outgrows its inline part
                    nd6_step<NE, SCAN, true>(n_g, h_gs, g, A.tk[k], dg, g_d2, sink, W, W.ia);
                    nd6_step<NE, SCAN, true>(n_h, a_hd, g, A.tk[k], dh, nul, sink, W, W.ia);
                    if (NOLL3) nd6_step<NE, SCAN, true>(n_g2, h_gs2, g, A.tk[k], dg2, g2_d2, sink, W, W.ia);
                } else {
                    nd6_step<NE, SCAN, false>(n_g, h_gs, g, A.tk[k], dg, g_d2, sink, W, W.ia);
                    nd6_step<NE, SCAN, false>(n_h, a_hd, g, A.tk[k], dh, nul, sink, W, W.ia);
                    if (NOLL3) nd6_step<NE, SCAN, false>(n_g2, h_gs2, g, A.tk[k], dg2, g2_d2, sink, W, W.ia);
                }
            }
        }
        nd6_fin<NE>(n_g, do_vert, dg, g_d2, sink, W, W.ia);
        nd6_fin<NE>(n_h, win == 0, dh, nul, sink, W, W.ia);
        if (NOLL3) nd6_fin<NE>(n_g2, do_vert, dg2, g2_d2, sink, W, W.ia);
        v6_incdelta<NE, SCAN>(do_hori, dh_sel6<NE>(d.f_from_h, a_hl, a_fl), df, win == 3 ? dh : nul, sink16, W, W.ia);
        if (NOLL3) v6_incdelta<NE, SCAN>(do_hori, dh_sel6<NE>(d.f2_from_h, a_hl, a_f2), df2, win == 4 ? dh : nul, sink16, W, W.ia);
    }
    V6_STAMP(5)
    {   // b side: newdelta over b.t for F (F2) and a diagonal H; incdelta for G (G2)
        const DHn h_fs = dh_sel6<NE>(d.f_from_h, b_hl, b_fl), h_fs2 = dh_sel6<NE>(d.f2_from_h, b_hl, b_f2);
        ND6 n_f = {0, 0, 0, do_hori}, n_h = {0, 0, 0, win == 0}, n_f2 = {0, 0, 0, do_hori && NOLL3};
        lu32 *const f_d2 = win == 3 ? dh + ca4 : nul, *const f2_d2 = win == 4 ? dh + ca4 : nul;
        for (int k = 0; k < DL_GUARD; ++k) {
            if (wave_none(n_f.on || n_h.on || (NOLL3 && n_f2.on))) break;
            const SE6 e = se6_read(B.rt, B.ot + k);
            const int g = k < B.lent ? e.g : -1;
            if (NE > 5 && __ballot(n_f.kd >= W.ib || n_h.kd >= W.ib || (NOLL3 && n_f2.kd >= W.ib))) {
                nd6_step<NE, SCAN, true>(n_f, h_fs, g, e.key, df + ca4, f_d2, sink, W, W.ib);
                nd6_step<NE, SCAN, true>(n_h, b_hd, g, e.key, dh + ca4, nul, sink, W, W.ib);
                if (NOLL3) nd6_step<NE, SCAN, true>(n_f2, h_fs2, g, e.key, df2 + ca4, f2_d2, sink, W, W.ib);
            } else {
                nd6_step<NE, SCAN, false>(n_f, h_fs, g, e.key, df + ca4, f_d2, sink, W, W.ib);
                nd6_step<NE, SCAN, false>(n_h, b_hd, g, e.key, dh + ca4, nul, sink, W, W.ib);
                if (NOLL3) nd6_step<NE, SCAN, false>(n_f2, h_fs2, g, e.key, df2 + ca4, f2_d2, sink, W, W.ib);
            }
        }
        nd6_fin<NE>(n_f, do_hori, df + ca4, f_d2, sink, W, W.ib);
        nd6_fin<NE>(n_h, win == 0, dh + ca4, nul, sink, W, W.ib);
        if (NOLL3) nd6_fin<NE>(n_f2, do_hori, df2 + ca4, f2_d2, sink, W, W.ib);
        v6_incdelta<NE, SCAN>(do_vert, dh_sel6<NE>(d.g_from_h, b_hu, b_gu), dg + ca4, win == 1 ? dh + ca4 : nul, sink16, W, W.ib);
        if (NOLL3) v6_incdelta<NE, SCAN>(do_vert, dh_sel6<NE>(d.g2_from_h, b_hu, b_g2), dg2 + ca4, win == 2 ? dh + ca4 : nul, sink16, W, W.ib);
    }
#endif
    V6_STAMP(6)
    v3_outputs<2, NOLL3>(d, 0, 0, do_vert, do_hori, oH, oG, oG2, oF, oF2, trb);
    V6_STAMP(7)
    return true;
}

// ---- one STRIP (64 rows x all columns) by one wave, pipelined behind the strip above on progress counters -----------
// Strip-boundary records cross XCDs (each XCD has its own L2).  With G2G_V6_NOFENCE they are written with agent-scope
// (write-through) stores and read with agent-scope loads, so that publishing / consuming progress needs no buffer_wbl2 /
// buffer_inv of the whole L2; without it they are plain accesses ordered by release / acquire fences.
#ifdef G2G_V6_NOFENCE
#define V6_XLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define V6_XST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define V6_ACQUIRE()
#define V6_RELEASE()
#else
#define V6_XLD(p) (*(p))
#define V6_XST(p, v) (*(p) = (v))
#define V6_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define V6_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#endif
template <bool NOLL3, int NA>
__device__ __forceinline__ void v6_strip(const DevProb &Pmem, lchar *lds, const V6Lds LO, const int ti, const int nsteps,
                                         const int *prog_up, int *prog_self, int *dbg, const int pgen, const int pint, const int *prog_left,
                                         double *simscr, int *failp, unsigned *twin)
{
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int lane = threadIdx.x;                          // blockDim.x == 64
    const int capa = P.capa, capb = P.capb;
    const int ca4 = v6_inline_dw((capa + 3) & ~3), cb4 = v6_inline_dw((capb + 3) & ~3), lsz = ca4 + cb4;      // INLINE dwords per side (LS6)
    const int nslot = NOLL3 ? 9 : 6;
    const int pitch = v3_pitch(nslot, lsz);
    const int ndw = ((16 + 4 * (capa + capb) + 15) & ~15) / 4;
    lu32 *const rows = (lu32 *) (lds + LO.rows);           // row 0: staging (the strip above), row t+1: lane t
    const LS6 W = {rows, glbw(twin), ca4, cb4};
    const unsigned TERM = DL_END << 16;
    lu32 *const blk = (lu32 *) (lds + LO.black);
    lu32 *const stsc = (lu32 *) (lds + LO.stsc);           // staging scalars: H ring 0-2, G 3-4, G2 5-6
#define V6_L(r, slot) (rows + (r) * pitch + (slot) * lsz)
    const size_t rbuf = (size_t) P.v2_rowstride * ndw;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const GLB unsigned *rowHp = glb((const unsigned *) P.v2_rowH + bprev * rbuf), *rowGp = glb((const unsigned *) P.v2_rowG + bprev * rbuf);
    const GLB unsigned *rowG2p = NOLL3 ? glb((const unsigned *) P.v2_rowG2 + bprev * rbuf) : 0;
    GLB unsigned *rowHc = glbw((unsigned *) P.v2_rowH + bcur * rbuf), *rowGc = glbw((unsigned *) P.v2_rowG + bcur * rbuf);
    GLB unsigned *rowG2c = NOLL3 ? glbw((unsigned *) P.v2_rowG2 + bcur * rbuf) : 0;
    const GLB unsigned *colH = glb((const unsigned *) P.v2_colH);
    const GLB int *boff0 = glb(b.off[0]), *boff1 = glb(b.off[1]), *boff2 = glb(b.off[2]);
    const GLB double *bthk = glb(b.thk);
    GLB uint8_t *const trace = glbw(P.trace);
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const int m0 = a.left + ti * 64, m = m0 + lane;
    if (prog_left) {                                       // the left boundary chain runs beside the strips (v2_chain_tile)
        const int rows_ = m0 + 64 - a.left;
        const int wantl = ((pgen & 0x7FF) << 20) | (rows_ < 0xFFFFF ? rows_ : 0xFFFFF);
        (void) g2g_wait_ge(prog_left, wantl, dbg, failp, ti);
        V6_ACQUIRE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int mend = (m0 + 64 < a.right) ? m0 + 64 : a.right;
    const int llast = mend - 1 - m0;                       // lane of the strip's last row
    const int c0 = b.left, c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo, hi = nhi;
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right;            // lane 0's hi
    const bool vert0 = m0 > a.left;                        // the strip has a row above

    // ---- LDS init: black list, every ring slot black (reset(f1), reset(f2), fwd2c.h:385-386; G of the DP's first row is
    // never written and is read as black by the row below, fwd2c.h:401)
    if (lane < 8) blk[lane] = (lane == 1) ? (DL_END << 16) : 0;
    if (lane < 8) blk[ca4 + lane] = (lane == 1) ? (DL_END << 16) : 0;
    for (int sl = 0; sl < nslot; ++sl) {
        lu32 *p = V6_L(lane + 1, sl);
        p[0] = 0; p[1] = DL_END << 16;
        p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    if (lane < nslot) {
        lu32 *p = V6_L(0, lane);
        p[0] = 0; p[1] = DL_END << 16;
        p[ca4] = 0; p[ca4 + 1] = DL_END << 16;
    }
    // ---- the row's static lists -> registers; the wave's longest list per view bounds the unrolled loops
    unsigned a_sk[NA], a_tk[NA];
    double a_sf[NA], a_tf[NA], a_rhf = 0;
    int ls = 0, lt = 0;
    {
        int g[NA];
        rl_load(g, a_sf, a, 0, m, row_ok);
        V6_UNROLL for (int k = 0; k < NA; ++k) { a_sk[k] = v6_key(g[k]); ls += g[k] >= 0; }
        rl_load(g, a_tf, a, 1, m, row_ok);
        V6_UNROLL for (int k = 0; k < NA; ++k) { a_tk[k] = v6_key(g[k]); lt += g[k] >= 0; }
        if (row_ok) {                                      // the r view's head entry, if any (glen 0; the t entries follow with glen + 1)
            const int o = a.off[2][m + 1];
            if (a.off[2][m + 2] - o - 1 > lt) a_rhf = a.freq[2][o];
        }
    }
    const A6<NA> A = {a_sk, a_sf, a_tk, a_tf, a_rhf, ls, lt};
    int TAs = 0, TAt = 0;
    while (__ballot(ls > TAs)) ++TAs;
    while (__ballot(lt > TAt)) ++TAt;
    // ---- the ring of the columns' static lists
    Ring6 ring[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) { ring[v].k = (const lu32 *) (lds + LO.ringk[v]); ring[v].f = (const lf64 *) (lds + LO.ringf[v]); ring[v].mask = LO.rs[v] - 1; }
    int fedcol = cbase - 1;                                // columns <= fedcol are in the ring
    lf64 *const rhring = (lf64 *) (lds + LO.ringf[2]);     // head freq of the r list of column c at [c & (V6_RHCOLS - 1)] (-1: no head)
    auto refill = [&](int upto) {                          // wave-uniform; lane <-> column, one view after the other (the view is a
        if (upto > b.right - 1) upto = b.right - 1;        // compile-time index: a per-lane choice among the descriptor's fields
#pragma unroll                                             // turns its register copy into an indexed object in scratch memory)
        for (int v = 0; v < 2; ++v) {
            const GLB int *bo = v == 0 ? boff0 : boff1;
            const GLB int *bgl = glb(b.glen[v]);
            const GLB double *bfr = glb(b.freq[v]);
            lu32 *rk = (lu32 *) (lds + LO.ringk[v]);
            lf64 *rf = (lf64 *) (lds + LO.ringf[v]);
            const int mask = LO.rs[v] - 1;
            for (int c0_ = fedcol + 1; c0_ <= upto; c0_ += 64) {
                const int col = c0_ + lane;
                if (col <= upto) {
                    const int k0 = bo[col + 1], len = bo[col + 2] - k0 - 1, c = k0 - (col + 1);
                    for (int e = 0; e < len; ++e) { rk[(c + e) & mask] = v6_key(bgl[k0 + e]); rf[(c + e) & mask] = bfr[k0 + e]; }
                }
            }
        }
        {
            const GLB double *bfr2 = glb(b.freq[2]);
            for (int c0_ = fedcol + 1; c0_ <= upto; c0_ += 64) {
                const int col = c0_ + lane;
                if (col <= upto) {
                    const int k2 = boff2[col + 1], lr_ = boff2[col + 2] - k2 - 1, lt_ = boff1[col + 2] - boff1[col + 1] - 1;
                    rhring[col & (V6_RHCOLS - 1)] = lr_ > lt_ ? bfr2[k2] : -1.;          // (-1: the column's r view has no head entry)
                }
            }
        }
        if (upto > fedcol) fedcol = upto;
    };
    // ---- the records this row starts from ------------------------------------------------------------
    RS oH = rs_black(), oG = rs_black(), oG2 = rs_black(), oF = rs_black(), oF2 = rs_black();
    if (row_ok && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left) {      // left boundary corner (m+1, b.left)
        lu32 *p = V6_L(lane + 1, SLOT_H(c0));
        const GLB unsigned *src = colH + (size_t) (m + 1 - a.left) * ndw;
        oH.val = __hiloint2double((int) V6_XLD(src + 1), (int) V6_XLD(src)); oH.dir = (int) V6_XLD(src + 2); oH.glb = (int) V6_XLD(src + 3);
        for (int k = 0; k < capa; ++k) { const unsigned v = V6_XLD(src + 4 + k); ls6_wr(W, p, k, ca4, v); if (v >= TERM) break; }
        for (int k = 0; k < capb; ++k) { const unsigned v = V6_XLD(src + 4 + capa + k); ls6_wr(W, p + ca4, k, cb4, v); if (v >= TERM) break; }
    }
    // ---- staging row: records of the strip above for lane 0's columns, one dword per lane --------------
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < ndw) {
            const GLB unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * ndw : rowHp + (size_t) col * ndw;
            rh = V6_XLD(s + lane);
            if (wantG) { rg = V6_XLD(rowGp + (size_t) col * ndw + lane); if (NOLL3) rg2 = V6_XLD(rowG2p + (size_t) col * ndw + lane); }
        }
    };
    // one record dword per lane: 0-3 the scalars, then the a-side list, then the b-side list.  Dwords beyond a list's inline
    // part go to its twin -- only while they are in front of (or are) the list's terminator: what follows it is stale
    auto stage_put = [&](int slot, int sid, unsigned v) {
        const int j = lane - 4;
        lu32 *const p = V6_L(0, slot);
        if (lane < 4) stsc[sid * 4 + lane] = v;
        else if (j < ca4 && j < capa) p[j] = v;
        else if (j >= capa && j - capa < cb4 && j < capa + capb) p[ca4 + j - capa] = v;
        if (capa > ca4 || capb > cb4) {                    // (wave-uniform: this DP's lists can outgrow their inline parts)
            const unsigned long long tb = __ballot(v >= TERM) >> 4;                              // bit j: dword j of the lists is a terminator
            const int ta = tb ? (int) __builtin_ctzll(tb) : 63;                                  // the a list's terminator (it has one)
            const unsigned long long tbb = capa < 60 ? tb >> capa : 0ull;
            const int tbp = tbb ? (int) __builtin_ctzll(tbb) : 63;                               // the b list's, relative to its start
            if (j >= ca4 && j < capa && j <= ta) ls6_wr(W, p, j, ca4, v);
            if (j >= capa + cb4 && j < capa + capb && j - capa <= tbp) ls6_wr(W, p + ca4, j - capa, cb4, v);
        }
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < ndw) {
            stage_put(SLOT_H(col), SLOT_H(col), rh);
            if (wantG) { stage_put(SLOT_G(col), 3 + (col & 1), rg); if (NOLL3) stage_put(SLOT_G2(col), 5 + (col & 1), rg2); }
        }
    };
    int avail = prog_up ? 0 : 0x7fffffff;                  // corner columns of the strip above known to be final
    const int penc = (pgen & 0x7FF) << 20;
    auto need = [&](const int col) {                       // wave-uniform: every lane polls, nobody branches alone
        const int want = penc | (col < 0xFFFFF ? col : 0xFFFFF);
        if (prog_up && want > avail) {
            avail = g2g_wait_ge(prog_up, want, dbg, failp, ti);
            V6_ACQUIRE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
// diagnostic heartbeat (build with -DG2G_V6_HEARTBEAT): step counter and a marker of the place in the step, stored next to the
// strip's progress word; the report of a time-out reads them (g2g_wait_ge).  Off by default: ten 4-byte write-through stores
// per step are ~30 GB/s of fabric traffic for nothing.
#if defined(G2G_V6_HEARTBEAT) || defined(G2G_HEARTBEAT)
#define V6_MARK(k) { if (prog_self) __hip_atomic_store(prog_self + G2G_DIAG + 2, (k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define V6_BEAT(s) { if (prog_self) __hip_atomic_store(prog_self + G2G_DIAG + 1, (s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
#define V6_MARK(k)
#define V6_BEAT(s)
#endif
    auto publish = [&](const int col) {                    // corners <= col of this strip's last row are in HBM
        if (prog_self) {
            V6_MARK(9)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            V6_MARK(10)
            V6_RELEASE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            V6_MARK(11)
            G2G_POST(prog_self, penc | (col < 0 ? 0 : col < 0xFFFFF ? col : 0xFFFFF));
            V6_MARK(12)
        }
    };
    if (lane < 28) stsc[lane] = 0;
    team_sync();
    refill(cbase + V6_AHEAD);
    need(cbase + 1 <= c1 ? cbase + 1 : cbase);
    {
        unsigned rh = 0, rg = 0, rg2 = 0;
        stage_load(cbase, false, rh, rg, rg2);
        stage_store(cbase, false, rh, rg, rg2);
        if (cbase + 1 <= c1) {
            stage_load(cbase + 1, vert0, rh, rg, rg2);
            stage_store(cbase + 1, vert0, rh, rg, rg2);
        }
    }
    // per-row constants and one-step-ahead register pipelines (column score, b's column thickness, list offsets)
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    SimBlk SB; SB.buf = (GLBV3 double *) simscr; SB.cbase = cbase;      // strip-local column scores (g2g_kernels_v3.hip)
    simblk_fill(P, SB, 0, m0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double sim_cur = 0, bc_cur = 0;
    int os_cur = 0, oe_cur = 0, ot_cur = 0, te_cur = 0;       // starts / ends (pool positions) of the column's s and t lists
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    team_sync();
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;                                  // staging registers hold column n0 + 1
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;    // the previous step's trace byte
    const int ull = __builtin_amdgcn_readfirstlane(llast);
    int lhi = m0 + llast + P.up + 1; if (lhi > b.right) lhi = b.right;
    int llo = m0 + llast + P.lw; if (llo < b.left) llo = b.left;
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
                    const lu32 *p = V6_L(llast + 1, slot);
                    v = (j < capa) ? (j < ca4 ? p[j] : TERM) : (j < capa + capb) ? (j - capa < cb4 ? p[ca4 + j - capa] : TERM) : 0;
                    // a list that runs beyond its inline part (its last inline dword is an entry): the rest comes from the twin
                    if (capa > ca4 && p[ca4 - 1] < TERM && j >= ca4 && j < capa) v = ls6_rd(W, p, j, ca4);
                    if (capb > cb4 && p[ca4 + cb4 - 1] < TERM && j >= capa + cb4 && j < capa + capb) v = ls6_rd(W, p + ca4, j - capa, cb4);
                }
                GLB unsigned *dst = (x == 0) ? rowHc : (x == 1) ? rowGc : rowG2c;
                if (lane < ndw) V6_XST(dst + (size_t) col * ndw + lane, v);
            }
        }
    };
#ifdef G2G_V6_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#endif
    if (prog_self) {                                       // where this strip runs (read by the report of a time-out, g2g_wait_ge)
        __hip_atomic_store(prog_self + G2G_DIAG + 3, (int) __builtin_amdgcn_s_getreg((31 << 11) | 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(prog_self + G2G_DIAG + 4, 0x100 | ((int) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int s = 0; s < nsteps; ++s) {
        V6_STAMP(9)
        V6_BEAT(s)
        V6_MARK(1)
        const int n = cbase + s - lane;
        const int n0 = cbase + s;                          // lane 0's column
        const bool active = row_ok && n >= lo && n < hi;
        // -- top of the step: consume last step's loads, issue last step's stores
        if (st_prev) stage_store(n0 + 1, vert0, st_h, st_g, st_g2);
        if (p_act) trace[p_tri] = (uint8_t) p_trb;
        if (wr_rows && s > 0) flush_rows(n0 - 1 - llast);
        if (prog_self && s > 0 && (s & (pint - 1)) == 0) publish(n0 - llast);
        if ((s & (V6_FEED - 1)) == 0) { V6_MARK(2) refill(n0 + V6_AHEAD); team_sync(); }
        if ((s & 63) == 0) { V6_MARK(3) simblk_fill(P, SB, (s >> 6) + 1, m0, lane); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        // -- hand-over from the row above
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
        // -- loads for the next step: next column's score / thickness / list offsets; the strip above's records two columns ahead
        double sim_nx = 0, bc_nx = 0;
        int os_nx = 0, oe_nx = 0, ot_nx = 0, te_nx = 0;
        if (active) {
            if (!have) {
                sim_cur = *simblk_at(SB, lane, n); bc_cur = bthk[(size_t) (n + 1) * 3];
                os_cur = boff0[n + 1]; oe_cur = boff0[n + 2]; ot_cur = boff1[n + 1]; te_cur = boff1[n + 2];
            }
            if (n + 1 < hi) {
                sim_nx = *simblk_at(SB, lane, n + 1); bc_nx = bthk[(size_t) (n + 2) * 3];
                os_nx = oe_cur; oe_nx = boff0[n + 3]; ot_nx = te_cur; te_nx = boff1[n + 3];
            }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) { V6_MARK(4) need(n0 + 2); stage_load(n0 + 2, vert0, st_h, st_g, st_g2); }
        V6_MARK(5)
        RS myH = oH, myG = oG, myG2 = oG2;                 // (the produced records of this step)
        V6_STAMP(0)
        if (active) {                                      // (loops inside are uniform over the ACTIVE lanes: ballots see only them)
            const bool do_hori = n > b.left;
            B6 B;
            B.rs = ring[0]; B.rt = ring[1];
            B.os = os_cur - (n + 1); B.ot = ot_cur - (n + 1);                                     // compact positions (see Ring6)
            B.lens = oe_cur - os_cur - 1; B.lent = te_cur - ot_cur - 1;
            { const double hv = rhring[n & (V6_RHCOLS - 1)]; B.rhf = hv >= 0 ? hv : 0.; B.lenr = B.lent + (hv >= 0 ? 1 : 0); }
            const bool up_in = do_vert && (n - (m - 1) <= P.up);          // cell (m-1, n) exists
            const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
            const RS bk = rs_black();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const lu32 *hdl = V6_L(lane, SLOT_H(n));
            const lu32 *hul = up_in ? V6_L(lane, SLOT_H(n + 1)) : blk;
            const lu32 *gul = up_in ? V6_L(lane, SLOT_G(n + 1)) : blk;
            const lu32 *g2ul = (NOLL3 && up_in) ? V6_L(lane, SLOT_G2(n + 1)) : blk;
            const lu32 *hll = left_in ? V6_L(lane + 1, SLOT_H(n)) : blk;
            const lu32 *fll = left_in ? V6_L(lane + 1, SLOT_F) : blk;
            const lu32 *f2ll = (NOLL3 && left_in) ? V6_L(lane + 1, SLOT_F2) : blk;
            lu32 *const sink = (lu32 *) (lds + LO.sink) + lane;
            lu32 *const sink16 = (lu32 *) (lds + LO.sink) + 64 + 4 * lane;
            lu32 *dh = V6_L(lane + 1, SLOT_H(n + 1));
            lu32 *dg = V6_L(lane + 1, SLOT_G(n + 1));
            lu32 *dg2 = V6_L(lane + 1, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
            lu32 *df = V6_L(lane + 1, SLOT_F);
            lu32 *df2 = V6_L(lane + 1, NOLL3 ? SLOT_F2 : SLOT_F);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            int trb = 0;
// three instances of the cell: heads of 5 entries, of 8, and 8 + scan (see dh_load6); the first that applies runs
            if (!v6_cell_pf<NOLL3, NA, 5, false>(P, W, A, TAs, TAt, B, sink, sink16, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                                  dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb V6_STAMP_PASS))
                if (!v6_cell_pf<NOLL3, NA, 8, false>(P, W, A, TAs, TAt, B, sink, sink16, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                                  dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb V6_STAMP_PASS))
                    v6_cell_pf<NOLL3, NA, 8, true>(P, W, A, TAs, TAt, B, sink, sink16, hd, hdl, s_hu, hul, s_gu, gul, s_g2u, g2ul, s_hl, hll, s_fl, fll, s_f2l, f2ll,
                                  dh, dg, dg2, df, df2, do_vert, do_hori, sim_cur, pua, pub, myH, myG, myG2, oF, oF2, trb V6_STAMP_PASS);
            const int d = m + n;
            int mlo, mhi;
            diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (d - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; have = (n + 1 < hi);
            os_cur = os_nx; oe_cur = oe_nx; ot_cur = ot_nx; te_cur = te_nx;
            if (m == a.right - 1 && n == b.right - 1) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        V6_MARK(6)
        team_sync();
    }
    V6_MARK(7)
#ifdef G2G_V6_STAMP
    if (lane == 0) { for (int k = 0; k < 12; ++k) atomicAdd(&g2g_v6_stamp_acc[k], st_acc[k]); atomicAdd(&g2g_v6_stamp_acc[12], (unsigned long long) nsteps); }
#endif
    if (p_act) trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - llast);
    publish(0xFFFFF);
    V6_MARK(8)
#undef V6_L
}

#define V6_KERNEL(NAME, N3, NA, WPE)                                                                 \
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))      \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, V6Lds LO, int pint, int pro_off, double *simscr, unsigned *twin, int twin_dw) \
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
            v2_chain_tile<2>(probs[T.prob], (lchar *) g2g_lds, T.ti, done + T.self, gen, pro_off);  \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        int *failp = done + done[G2G_HDR + 2] + T.prob;                                                      \
        if (threadIdx.x == 0) s_vals[0] = g2g_dp_failed(failp) ? 1 : 0;      /* (one reader: the branch must be uniform) */ \
        __syncthreads();                                                                            \
        const int dp_dead = s_vals[0];                                                              \
        __syncthreads();                                                                            \
        if (dp_dead) {                    /* this DP lost a wait: its strips are skipped, dependents released */ \
            if (threadIdx.x == 0) G2G_POST(done + T.self, ((gen & 0x7FF) << 20) | 0xFFFFF); \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        const int *pl = T.dep_left >= 0 ? done + T.dep_left : (const int *) 0;                      \
        const int *pu = T.dep_up >= 0 ? done + T.dep_up : (const int *) 0;                          \
        __syncthreads();                                                                            \
        v6_strip<N3, NA>(probs[T.prob], (lchar *) g2g_lds, LO, T.ti, T.nsteps, pu, done + T.self, done + G2G_HDR, gen, pint, pl, \
                         simscr + (size_t) blockIdx.x * G2G_SIMBLK_STRIDE, failp, twin + (size_t) blockIdx.x * twin_dw); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
    }                                                                                               \
}
#ifndef G2G_V6_NA
#define G2G_V6_NA 16
#endif
#ifndef G2G_V6_NA3
#define G2G_V6_NA3 10                   /* Noll 3: two more records per cell -- with 16-entry row lists the kernel would spill 94 registers */
#endif
#ifdef G2G_TU_V6
V6_KERNEL(g2g_v6_pf2, false, G2G_V6_NA, 1)
V6_KERNEL(g2g_v6_pf3, true, G2G_V6_NA3, 1)
#else
#define V6_KERNEL_DECL(NAME) extern "C" __global__ void NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, V6Lds LO, int pint, int pro_off, double *simscr, unsigned *twin, int twin_dw);
V6_KERNEL_DECL(g2g_v6_pf2) V6_KERNEL_DECL(g2g_v6_pf3)
#endif
