// g2g_kernels_v2.hip -- 8-lanes-per-cell forward kernels for the gap-profile engines (_hf, _pf; Noll 2/3); the
// default for _pf (the _hf default is g2g_kernels_v3.hip).  Also: the list primitives every later generation uses,
// the boundary-chain prologue, the column-score kernel and the tile/strip scheduler conventions.
//
// Same recurrence, same arithmetic order as g2g_kernels.hip (Fwd2c::forwardB, reference
// src/fwd2c.h:359-482 + src/fwd2c.cc:152-251 + src/gfreq.cc:493-605), different machine mapping:
//
//  * ROW STRIPS.  A workgroup sweeps one DP in strips of R = blockDim/8 rows.  Inside a strip row t runs
//    one column behind row t-1 (skewed wavefront, one barrier per step), so every record a cell needs
//    was produced one or two steps earlier by the row itself or by the row above.
//  * STATE IN LDS.  Each row keeps small rings of its own DP records -- H: 3 slots (corners n-1, n, n+1),
//    G: 2, F: 1 (+G2: 2, F2: 1 for the double-affine penalty) -- in LDS, dynamic gap-state lists packed
//    as 16+16-bit {glen, nins}.  Nothing but the strip boundary (the last row's H/G records, one record
//    per column) and the trace bytes goes back to HBM; v1 streamed every record through HBM.
//  * A TEAM OF 8 LANES PER CELL.  The six gap-open costs of a cell (two diagonal, two vertical, two
//    horizontal list merges) and the column score are independent, as are the list updates of the
//    three produced records: lanes of a team take one each, exchange scalars with wave shuffles, and
//    all lanes replay the (cheap) scalar decision logic.  This cuts the dependent-access chain per cell
//    ~10x, which is what bounds a single DP's sweep time.
// The boundary chains (initB) run once in a prologue and are parked in HBM (top row: RowH, left column:
// ColH).  Trace bytes and the backtrack kernel are shared with v1.
#include <hip/hip_runtime.h>

#define TEAM 8
#ifdef G2G_V2_STAMP
static __device__ unsigned long long g2g_stamp_acc[16];      // (one copy per translation unit: g2g_tu_v2.hip reads its own, g2g_v2_stamps)
#define STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += t_ - stamp_t; stamp_t = t_; }
#else
#define STAMP(k)
#endif
#define DL_END 0xFFFFu                       // packed terminator glen (INT_MAX in the reference)
// ---- bounded waits on progress words of other resident workgroups -------------------------------------------------------
// Persistent workgroups poll flags / progress counters that other workgroups of the SAME launch write.  Dependencies sit
// earlier in the queue than their dependents, so a wait can only be long, not endless -- unless something outside the design
// happens (several processes oversubscribing the device were seen to stretch waits past an iteration-count bound).  The
// bound is therefore WALL CLOCK (s_memrealtime: 100 MHz, keeps running while a wave is descheduled), set per launch by the
// host (hdr[3], units of 65536 ticks = 0.655 ms), and a time-out costs ONE DP, not the batch: the DP is marked in the
// batch's fail array, every other wait of that DP gives up at its next check, its remaining strips are skipped, and
// g2g_batch_run re-runs the marked DPs (on the ordinary kernels once more, then on the non-polling g2g_forward_kernel).
// hdr = done + G2G_HDR: [0] time-outs, [1] strip index of the first, [2] offset of the fail array from `done`, [3] the limit,
// [4..47] what the first wave to give up saw (the host prints it under G2G_WARN; DESIGN.md 4.2 reads such reports).
#define G2G_HDR 24                   // d_flags: [0, 24) queue heads of the kernel variants, [24, 28) this header, [28, 72) snapshot of the first time-out, tile flags behind
// Two 128-byte lines per strip.  The FIRST holds the progress word and nothing else: it is the line other workgroups poll, and the
// only store that ever goes into it is the publish itself.  Everything the strip leaves for a time-out report (HW_ID, markers,
// per-wave columns, heartbeats) lives in the SECOND line, G2G_DIAG ints on, which nobody polls.  Round 4's reports showed what the
// stopped workgroups of DESIGN.md 4.2 were doing: the head strip of every pipeline of one launch sat at an s_waitcnt vmcnt(0)
// behind a write-through store INTO THE LINE ITS SUCCESSOR WAS POLLING (its queue marker, its per-wave column, the progress
// word itself), for as long as the polling went on -- a store starved by a stream of coherent loads from another XCD.  Hence
// also G2G_POLL / G2G_POST: with -DG2G_POLL_RMW the progress word is read and written with read-modify-write atomics, which
// execute at the memory side and leave no copy of the line in anybody's L2.  MEASURED (three whole refinements each way, round
// 4): the events went on with both remedies in place (1, 1 and 3 per run), and the RMW polls cost 4 % -- the hypothesis is refuted,
// the split lines stay (they cost nothing), the polls and publishes are plain agent-scope loads and stores again.
#define G2G_FSTRIDE 64               // ints between two tile flags / progress words
#define G2G_DIAG 32                  // offset of a strip's diagnostics line from its progress word
#ifdef G2G_POLL_RMW
#define G2G_POLL(p) __hip_atomic_fetch_or((int *) (p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define G2G_POST(p, v) ((void) __hip_atomic_exchange((int *) (p), (int) (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#else
#define G2G_POLL(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define G2G_POST(p, v) __hip_atomic_store((int *) (p), (int) (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#define G2G_DUMP_STRIPS 580           // strips of one DP the first time-out dumps for the host ...
#define G2G_DUMP_WORDS 11             // ... words each: progress word, HW_ID, three markers, two publish columns, (step, place) of two waves
#define G2G_GAP_TICKS 400000ull       // 4 ms of s_memrealtime: more than ten times what 64 polls take
#define G2G_HDRN 104                  // header words: 4 + the snapshot (want, seen, offset of the polled word, the words at and below it; [48, 64): per-wave heartbeats); G2G_HDR + G2G_HDRN is a multiple of G2G_FSTRIDE, so every progress line IS one 128-byte line
__device__ __forceinline__ int g2g_wait_ge(const int *p, const int want, int *hdr, int *failp, const int slot)
{
    int v = G2G_POLL(p);
    if (v >= want) return v;
    // The limit counts the time THIS wave was running: a gap of more than G2G_GAP_TICKS between two looks at the clock (64 polls:
    // 0.3 ms when the wave runs) means the wave itself was off the machine -- and with it, as a rule, the rest of its kernel, the
    // producer included -- so only G2G_GAP_TICKS of it count.  Gaps are counted in hdr[40] (longest, in units of 1024 ticks, in
    // hdr[41]): the host reports them whether or not a wait was lost.
    unsigned long long tl = __builtin_amdgcn_s_memrealtime(), run = 0;
    for (unsigned it = 1; ; ++it) {
        // back off: the first polls come 0.2 us apart, later ones 2-3 us with a phase that differs from wave to wave (hundreds of
        // waits per sweep last tens of ms -- strips pulled long before their producers get going: no point in hammering the fabric)
        const unsigned nap = it < 16 ? 1 : it < 64 ? 2 + (it & 1) : 8 + ((it * 5 + (unsigned) slot) & 7);
        for (unsigned j = 0; j < nap; ++j) __builtin_amdgcn_s_sleep(8);
        v = G2G_POLL(p);
        if (v >= want) return v;
        if ((it & 63) == 0) {
            if (__hip_atomic_load(failp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {                       // this DP is lost already
                // ... and this wave was waiting too: of all the waves released this way the one with the LOWEST strip index leaves what
                // it was waiting for and what it last saw (the blocker of its DP, if the blocker itself sat in a wait)
                const int key = 0x7fffffff - slot;
                if (key > atomicMax(hdr + 72, key)) {
                    hdr[73] = want; hdr[74] = v; hdr[75] = (int) (p - (hdr - G2G_HDR)); hdr[76] = (int) it; hdr[77] = (int) (run >> 16);
                    hdr[78] = atomicAdd((int *) p, 0);
                }
                return 0x7fffffff;
            }
            const unsigned long long tn = __builtin_amdgcn_s_memrealtime();
            unsigned long long d = tn - tl;
            tl = tn;
            if (d > G2G_GAP_TICKS) {
                atomicAdd(hdr + 40, 1);
                atomicMax(hdr + 41, (int) (d >> 10 > 0x7fffffffull ? 0x7fffffffull : d >> 10));
                d = G2G_GAP_TICKS;
            }
            run += d;
            if ((run >> 16) > (unsigned long long) (unsigned) __hip_atomic_load(hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(failp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                {   // where the waves that gave up sit, and where their producers sit (v6 strips leave HW_ID / XCC_ID next to their progress word)
                    const int my_xcc = (int) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
                    const int pr_xcc = __hip_atomic_load(p + G2G_DIAG + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicAdd(hdr + 24 + (my_xcc & 7), 1);
                    if ((pr_xcc & ~15) == 0x100) atomicAdd(hdr + 32 + (pr_xcc & 7), 1);
                }
                if (atomicAdd(hdr, 1) == 0) {
                    hdr[20] = __hip_atomic_load(p + G2G_DIAG + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hdr[21] = __hip_atomic_load(p + G2G_DIAG + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hdr[22] = (int) __builtin_amdgcn_s_getreg((31 << 11) | 4);
                    hdr[23] = (int) __builtin_amdgcn_s_getreg((31 << 11) | 20);             // the first one leaves a snapshot for the host's report
                    hdr[1] = slot; hdr[4] = want; hdr[5] = v;
                    const int off = (int) (p - (hdr - G2G_HDR));
                    hdr[6] = off; hdr[7] = (int) (failp - (hdr - G2G_HDR));
                    for (int k = 0; k < 8 && off - k * G2G_FSTRIDE >= G2G_HDR + G2G_HDRN; ++k) hdr[8 + k] = __hip_atomic_load(p - k * G2G_FSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // the producer's heartbeat (v6 strips: step counter and a marker of the place in the step, stored next to the
                    // progress word), read twice 50 us apart: is the producer running, and where?
                    hdr[16] = __hip_atomic_load(p + G2G_DIAG + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hdr[17] = __hip_atomic_load(p + G2G_DIAG + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int w = 0; w < 8; ++w) hdr[56 + w] = __hip_atomic_load(p + G2G_DIAG + 12 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int k = 0; k < 256; ++k) __builtin_amdgcn_s_sleep(8);
                    hdr[18] = __hip_atomic_load(p + G2G_DIAG + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hdr[19] = __hip_atomic_load(p + G2G_DIAG + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int w = 0; w < 4; ++w) hdr[42 + w] = __hip_atomic_load(p + G2G_DIAG + 8 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the producer's waves at their last publish (v2 / v3 strips)
                    // per-wave heartbeats of a v2 / v3 producer (builds with -DG2G_HEARTBEAT: (step, place) of each of its waves,
                    // G2G_HB below), read now and once more behind the 50 us above: which wave stands still, and where
                    for (int w = 0; w < 8; ++w) hdr[48 + w] = hdr[56 + w];
                    for (int w = 0; w < 8; ++w) hdr[56 + w] = __hip_atomic_load(p + G2G_DIAG + 12 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    {   // the BLOCKER: walk down the strips of this DP from the polled word while they have not published anything in this
                        // generation; the lowest such strip has a finished strip (or the top chain) above it -- it waits for nobody's
                        // progress, so where IT stands is the question.  Its markers (v2 strips): +5 taken from the queue by workgroup
                        // (0x20000 | id), +6 past the wait for the left chain, +7 past the first look at the strip above.
                        int kb = 0;
                        const int gen_now = want & ~0xFFFFF;
                        for (int k = 1; k < 64 && off - k * G2G_FSTRIDE >= G2G_HDR + G2G_HDRN; ++k) {
                            const int w = __hip_atomic_load(p - k * G2G_FSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (w >= (gen_now | 1)) break;            // published something in this generation
                            kb = k;
                        }
                        const int *q = p - kb * G2G_FSTRIDE;
                        hdr[64] = kb;
                        hdr[65] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int w = 0; w < 5; ++w) hdr[66 + w] = __hip_atomic_load(q + G2G_DIAG + 3 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hdr[71] = __hip_atomic_load(q + G2G_DIAG + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // sweep mode with the chains in the queue: the two chain words of this DP sit right below its first strip's word
                    if (off - (slot + 1) * G2G_FSTRIDE >= G2G_HDR + G2G_HDRN) {
                        hdr[80] = __hip_atomic_load(p - slot * G2G_FSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // left chain
                        hdr[81] = __hip_atomic_load(p - (slot + 1) * G2G_FSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // top chain
                    }
                    {   // the whole pipeline above this waiter (the strips' lines are contiguous, the one above p at p - G2G_FSTRIDE): word,
                        // HW_ID, the three markers, the two waves' last publish -- the host finds the strips that wait for nobody in it
                        const int doff = __hip_atomic_load(hdr + 82, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (doff > 0) {
                            int *dump = (hdr - G2G_HDR) + doff;
                            int nd = 0;
                            for (int k = 0; k < G2G_DUMP_STRIPS && k < slot && off - k * G2G_FSTRIDE >= G2G_HDR + G2G_HDRN; ++k) {
                                const int *q = p - k * G2G_FSTRIDE;
                                int *d = dump + 2 + G2G_DUMP_WORDS * k;
                                d[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[1] = __hip_atomic_load(q + G2G_DIAG + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[2] = __hip_atomic_load(q + G2G_DIAG + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[3] = __hip_atomic_load(q + G2G_DIAG + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[4] = __hip_atomic_load(q + G2G_DIAG + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[5] = __hip_atomic_load(q + G2G_DIAG + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                d[6] = __hip_atomic_load(q + G2G_DIAG + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                for (int w = 0; w < 4; ++w) d[7 + w] = __hip_atomic_load(q + G2G_DIAG + 12 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (-DG2G_HEARTBEAT: step / place of the strip's two waves)
                                nd = k + 1;
                            }
                            dump[0] = nd; dump[1] = slot;
                        }
                    }
                    hdr[8 + 38] = atomicAdd((int *) p, 0);          // the same word through a read-modify-write (executes at the coherent point)
                    hdr[8 + 39] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return 0x7fffffff;
            }
        }
    }
}
__device__ __forceinline__ bool g2g_dp_failed(const int *failp) { return __hip_atomic_load(failp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
// Strip-boundary records cross workgroups (and XCDs: each has its own L2).  Plain accesses ordered by agent-scope release /
// acquire fences cost a write-back of the whole L2 (buffer_wbl2) per publish and an invalidate (buffer_inv) per consumed
// publish -- from hundreds of workgroups, every few steps when a handful of DPs publish every 4 steps (g2g_refine's windows).
// The records themselves are therefore written with agent-scope (write-through) stores and read with agent-scope loads, and
// publishing is: wait for the stores, store the progress word -- no cache-maintenance operation in the step loop.  Measured
// speed-neutral; adopted because the producers that were seen to stop stopped where a wave
// can only be waiting for its own memory operations.  -DG2G_FENCE restores the fenced form.
#ifndef G2G_FENCE
#define G2G_NOFENCE 1
#endif
#ifdef G2G_NOFENCE
#define G2G_XLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define G2G_XST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define G2G_ACQUIRE()
#define G2G_RELEASE()
#else
#define G2G_XLD(p) (*(p))
#define G2G_XST(p, v) (*(p) = (v))
#define G2G_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define G2G_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#endif
// Passive producer-side heartbeat of the v2 / v3 strips (build with G2G_EXTRA_FLAGS=-DG2G_HEARTBEAT, which also turns on the v6
// strips' G2G_V6_HEARTBEAT): every wave of a strip keeps (step, place in the step) in words 12 + 2 w / 13 + 2 w of the strip's
// progress line; whoever times out waiting for the strip copies them into its report twice, 50 us apart (g2g_wait_ge,
// hdr[48..63]) -- the report then names the wave that stands still and the place it stands at.  Off by default: a handful of
// write-through stores per step and wave.  Places (v2): 1 top of the step, 2 publish entered, 3 behind the publish barrier,
// 4 waiting for the strip above, 5 behind the column-score block, 6 sources staged, 7 cell done, 8 at the step's barrier.
#ifdef G2G_HEARTBEAT
#define G2G_HB_STEP(pself, w, s) { if (pself) __hip_atomic_store((pself) + G2G_DIAG + 12 + 2 * (w), (int) (s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define G2G_HB(pself, w, k) { if (pself) __hip_atomic_store((pself) + G2G_DIAG + 13 + 2 * (w), (int) (k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
#define G2G_HB_STEP(pself, w, s)
#define G2G_HB(pself, w, k)
#endif
#define DL_GUARD 128                         // no list is this long: a corrupted one must not hang the wave

// Everything below lives in LDS and says so in its pointer types (address space 3): generic pointers
// would compile to flat_load/flat_store instead of ds_read/ds_write.
#define LDS __attribute__((address_space(3)))
typedef LDS char lchar;
typedef LDS unsigned lu32;
typedef LDS int li32;
typedef LDS double lf64;

// ---- LDS record: {f64 val; i32 dir; i32 glb; u32 dla[capa]; u32 dlb[capb]} -----------------------
struct LRec { lchar *p; };
__device__ __forceinline__ lf64 &lval(LRec r) { return *(lf64 *) r.p; }
__device__ __forceinline__ li32 &ldir(LRec r) { return *(li32 *) (r.p + 8); }
__device__ __forceinline__ li32 &lglb(LRec r) { return *(li32 *) (r.p + 12); }
__device__ __forceinline__ lu32 *ldla(LRec r) { return (lu32 *) (r.p + 16); }
__device__ __forceinline__ lu32 *ldlb(LRec r, int capa) { return (lu32 *) (r.p + 16) + capa; }
// a static GFREQ list cached in LDS
struct LList { const li32 *glen; const lf64 *freq; };
// v2 keeps the gap lengths of its LDS list caches as 16 bit (all lengths < 65000 on this path, terminator -1): 2.5 KB per
// workgroup, which is what lets a sixth 2-wave workgroup fit on a CU
typedef LDS short li16;
struct LList16 { const li16 *glen; const lf64 *freq; };

__device__ __forceinline__ void team_sync()
{   // lanes of a team live in one wave: ordering LDS traffic between phases is a compiler matter only
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- register-cached list heads ---------------------------------------------------------------------
// Lists are short (1-4 entries is the common case), and a dependent LDS read costs more than a dozen ALU
// ops, so the first four entries of every list a merge touches are read up front (independent reads, one
// wait) and indexed with selects; longer lists fall through to LDS for index >= 4.
struct DHead { unsigned e0, e1, e2, e3; const lu32 *p; };          // dynamic {glen,nins} list
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
// AL: the list starts on a 16-byte boundary (v3 layout) -- one ds_read_b128
template <bool AL = false>
__device__ __forceinline__ DHead dh_load(const lu32 *p)
{
    DHead h;
    if (AL) { const v4u32 v = *(const LDS v4u32 *) p; h.e0 = v.x; h.e1 = v.y; h.e2 = v.z; h.e3 = v.w; }
    else { h.e0 = p[0]; h.e1 = p[1]; h.e2 = p[2]; h.e3 = p[3]; }
    h.p = p; return h;
}
// nins of the entry that governs static gap length g: last k with glen[k] <= g (GapLenSD, gfreq.h:67).
// Entries behind the terminator are garbage: the AND chain stops at the first failed comparison.
__device__ __forceinline__ int dh_nins(int g, const DHead &h)
{
    const bool c1 = g >= (int) (h.e1 >> 16);
    const bool c2 = c1 && g >= (int) (h.e2 >> 16);
    const bool c3 = c2 && g >= (int) (h.e3 >> 16);
    unsigned e = c2 ? h.e2 : (c1 ? h.e1 : h.e0);
    if (c3) {                                                       // rare: more than three entries below g
        int k = 3;
        while (g >= (int) (h.p[k + 1] >> 16) && k < DL_GUARD) ++k;
        e = h.p[k];
    }
    return (int) (e & 0xFFFFu);
}
template <class CL> struct SHead { int g0, g1, g2, g3; CL l; };    // static GFREQ list (glen part)
template <class CL>
__device__ __forceinline__ SHead<CL> sh_load(const CL l)
{
    SHead<CL> h; h.g0 = l.glen[0]; h.g1 = l.glen[1]; h.g2 = l.glen[2]; h.g3 = l.glen[3]; h.l = l; return h;
}
// (the load behind index 3 sits behind a wave-uniform test: a conditional load inside the select chain is compiled
// into a tree of divergent branches)
template <class CL>
__device__ __forceinline__ int sh_glen(const SHead<CL> &h, int i)
{
    int g = i == 0 ? h.g0 : i == 1 ? h.g1 : i == 2 ? h.g2 : h.g3;
    if (__ballot(i > 3)) { const int t = h.l.glen[i > 3 ? i : 0]; g = i > 3 ? t : g; }
    return g;
}

// GapLenSD, gfreq.h:67, on a packed list (plain form, used by the boundary chains' helpers)
__device__ __forceinline__ int p_gaplen(int g, const lu32 *dl)
{
    int k = 0;
    while (g >= (int) (dl[k + 1] >> 16) && k < DL_GUARD) ++k;
    return g + (int) (dl[k] & 0xFFFFu);
}
// newgap(cf, dlc, df, dld), gfreq.cc:507-521.  The reference re-evaluates GapLenSD(cf) of the current cf
// at the start of every df step; the value cannot change, so it is carried in `gi`.
template <class CL, class DL, bool AL = false>
__device__ double p_newgap4(const CL cf, const lu32 *dlc, const DL df, const lu32 *dld)
{
    const DHead hc = dh_load<AL>(dlc), hd = dh_load<AL>(dld);
    const SHead<CL> sc = sh_load(cf);
    const SHead<DL> sd = sh_load(df);
    double g = 0;
    int ci = 0;
    int cg = sc.g0;
    int gi = cg >= 0 ? cg + dh_nins(cg, hc) : 0;
    for (int di = 0; ; ++di) {
        const int dg = sh_glen(sd, di);
        if (dg < 0) break;
        const int j = dg + dh_nins(dg, hd);
        while (cg >= 0 && gi < j) {
            ++ci;
            cg = sh_glen(sc, ci);
            gi = cg >= 0 ? cg + dh_nins(cg, hc) : 0;
        }
        if (cg < 0) break;
        g += cf.freq[ci] * df.freq[di];
    }
    return g;
}
// newgap1 / newgap2, maln.h:296-308 (+ newgapc/newgapd :280-291)
template <class CL, bool AL = false>
__device__ double p_newgap1(const DevProb &P, const CL acf, const lu32 *dla, int glb)
{
    const SHead<CL> sc = sh_load(acf);
    if (sc.g0 < 0) return 0;
    const DHead h = dh_load<AL>(dla);
    if (sc.g1 >= 0) {
        for (int ci = 0; ; ++ci) {                          // newgap(cf, dlc, j), gfreq.cc:523-532
            const int cg = sh_glen(sc, ci);
            if (cg < 0) break;
            if (cg + dh_nins(cg, h) >= glb) return P.weighted_gop * acf.freq[ci];
        }
        return P.weighted_gop * 0.;
    }
    return ((int) (h.e0 & 0xFFFFu) + sc.g0 >= glb) ? (P.weighted_gop * acf.freq[0]) : 0;
}
template <class CL, bool AL = false>
__device__ double p_newgap2(const DevProb &P, const CL adf, int glb, const lu32 *dla)
{
    const SHead<CL> sd = sh_load(adf);
    if (sd.g0 < 0) return 0;
    const DHead h = dh_load<AL>(dla);
    if (sd.g1 >= 0) {                                       // newgap(df, i, dld), gfreq.cc:534-545
        double g = 0;
        for (int di = 0; ; ++di) {
            const int dg = sh_glen(sd, di);
            if (dg < 0) break;
            if (glb < dg + dh_nins(dg, h)) break;
            g += adf.freq[di];
        }
        return P.weighted_gop * g;
    }
    return (glb >= (int) (h.e0 & 0xFFFFu) + sd.g0) ? (P.weighted_gop * adf.freq[0]) : 0;
}
// newdelta(dlt, df, dln, 1), gfreq.cc:570-587; up to two destinations (the record itself and, when that
// record wins the cell, the new H).  The source list is read through its cached head, i.e. before any
// store: when dst aliases src (F updated in place) this is the out-of-place result, which equals the
// reference's in-place one (an entry is overwritten only after the scan has moved past it, and the first
// entry's glen is always 0).
template <class CL, bool AL = false>
__device__ void p_newdelta(lu32 *dst, lu32 *dst2, const CL df, const lu32 *src)
{
    const DHead h = dh_load<AL>(src);
    const SHead<CL> sd = sh_load(df);
    int kd = 0;
    unsigned tg = 0, tn = 0;
    for (int di = 0; ; ++di) {
        const int g = sh_glen(sd, di);
        if (g < 0) break;
        const unsigned sn = (unsigned) dh_nins(g, h);
        if (sn > tn) {
            const unsigned e = (tg << 16) | tn;
            dst[kd] = e; if (dst2) dst2[kd] = e;
            ++kd;
            tn = sn;
            tg = (unsigned) (g + 1);
        }
    }
    const unsigned e = (tg << 16) | tn;
    dst[kd] = e; dst[kd + 1] = DL_END << 16;
    if (dst2) { dst2[kd] = e; dst2[kd + 1] = DL_END << 16; }
}
// incdelta(dlt, dln, 1), gfreq.cc:598-605
__device__ void p_incdelta(lu32 *dst, lu32 *dst2, const lu32 *src)
{
    int k = 0;
    for ( ; k < DL_GUARD; ++k) {
        unsigned e = src[k];
        if ((e >> 16) == DL_END) { dst[k] = e; if (dst2) dst2[k] = e; break; }
        e += 1;                                             // nins + 1
        dst[k] = e; if (dst2) dst2[k] = e;
    }
}
__device__ void p_copylist(lu32 *dst, const lu32 *src)
{
    for (int k = 0; ; ++k) { const unsigned e = src[k]; dst[k] = e; if ((e >> 16) == DL_END) break; }
}
__device__ __forceinline__ void p_clearlist(lu32 *d) { d[0] = 0; d[1] = DL_END << 16; }

// global <-> LDS record moves by one team (lane j moves dwords j, j+8, ...)
__device__ __forceinline__ void rec_g2l(LRec dst, const unsigned *src, int ndw, int lane)
{
    lu32 *d = (lu32 *) dst.p;
    for (int k = lane; k < ndw; k += TEAM) d[k] = src[k];
}
__device__ __forceinline__ void rec_l2g(unsigned *dst, LRec src, int ndw, int lane)
{
    const lu32 *s = (const lu32 *) src.p;
    for (int k = lane; k < ndw; k += TEAM) dst[k] = s[k];
}
// ... of records that cross strips (G2G_XLD / G2G_XST)
__device__ __forceinline__ void rec_g2l_x(LRec dst, const unsigned *src, int ndw, int lane)
{
    lu32 *d = (lu32 *) dst.p;
    for (int k = lane; k < ndw; k += TEAM) d[k] = G2G_XLD(src + k);
}
__device__ __forceinline__ void rec_l2g_x(unsigned *dst, LRec src, int ndw, int lane)
{
    const lu32 *q = (const lu32 *) src.p;
    for (int k = lane; k < ndw; k += TEAM) G2G_XST(dst + k, q[k]);
}

struct V2Geom {
    int capa, capb, recsz, ndw;       // list capacities, record bytes / dwords
    int nslot;                        // ring slots per row: 6 (Noll 2) or 9 (Noll 3)
    int R;                            // rows per strip
    lchar *lds;
    // slot ids inside a row
    // +16 B per row: the same field of the 8 rows of a wave must not share an LDS bank
    __device__ __forceinline__ LRec row(int t, int slot) const { LRec r; r.p = lds + t * (nslot * recsz + 16) + slot * recsz; return r; }
    __device__ __forceinline__ LRec extra(int k) const { LRec r; r.p = lds + R * (nslot * recsz + 16) + k * recsz; return r; }
};
// ring slots: H corner c -> c mod 3 (0..2); G corner c -> 3 + (c & 1); F -> 5; G2 -> 6 + (c & 1); F2 -> 8
__device__ __forceinline__ int mod3(int c) { return ((c % 3) + 3) % 3; }
#define SLOT_H(c) (mod3(c))
#define SLOT_G(c) (3 + ((c) & 1))
#define SLOT_F 5
#define SLOT_G2(c) (6 + ((c) & 1))
#define SLOT_F2 8
// extras: 0 black; staging of the strip's first row, whose upper neighbours live in HBM: 1,2 H corners
// (corner c in slot c & 1: this step's Hup is the next step's Hdiag), 3 Gup, 4 G2up
enum { EX_BLACK = 0, EX_H0 = 1, EX_H1 = 2, EX_GU = 3, EX_G2U = 4, EX_N = 5 };

template <int KIND>
__device__ __forceinline__ void lrec_black(LRec r, int capa)
{
    lval(r) = NEVSEL; ldir(r) = 0; lglb(r) = 0;
    p_clearlist(ldla(r));
    if (KIND == 2) p_clearlist(ldlb(r, capa));
}

// The six static lists a cell touches: a's s/t/r views at row m, b's at column n (LDS cache in the
// sweep, HBM in the boundary chains)
template <class SL> struct CellLists { SL as, at, ar, bs, bt, br; };

// Fwd2c<_hf/_pf>::gapopen (fwd2c.cc:152-160, 203-212), one list merge per call
template <int KIND, class SL>
__device__ __forceinline__ double gap_diag(const DevProb &P, const CellLists<SL> &L, LRec rc, int capa, int part)
{
    if (KIND == 1) return p_newgap2(P, L.at, lglb(rc), ldla(rc));
    if (part == 0) return p_newgap4(L.as, ldla(rc), L.bt, ldlb(rc, capa)) * P.basic_gop;
    return p_newgap4(L.bs, ldlb(rc, capa), L.at, ldla(rc)) * P.basic_gop;
}
template <int KIND, class SL>
__device__ __forceinline__ double gap_vert(const DevProb &P, const CellLists<SL> &L, LRec rc, int capa)
{
    if (KIND == 1) return p_newgap1(P, L.as, ldla(rc), lglb(rc));
    return p_newgap4(L.as, ldla(rc), L.br, ldlb(rc, capa)) * P.basic_gop;
}
template <int KIND, class SL>
__device__ __forceinline__ double gap_hori(const DevProb &P, const CellLists<SL> &L, LRec rc, int capa)
{
    if (KIND == 1) return p_newgap2(P, L.ar, lglb(rc), ldla(rc));
    return p_newgap4(L.bs, ldlb(rc, capa), L.ar, ldla(rc)) * P.basic_gop;
}

__device__ __forceinline__ void rec_store_all(unsigned *dst, LRec src, int ndw)
{
    const lu32 *sp = (const lu32 *) src.p;
    for (int k = 0; k < ndw; ++k) dst[k] = sp[k];
}

// progress of a boundary chain that runs as a queue entry of a persistent kernel (sweep mode): same counter format as a
// strip's (g2g_kernels_v3.hip); called by the walking lane after its stores
__device__ __forceinline__ void chain_publish(int *prog, int penc, int v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2G_POST(prog, penc | (v < 0xFFFFF ? v : 0xFFFFF));
}
// ---- prologue: the boundary chains of initB (fwd2c.h:138-176), one lane each, lists straight from HBM
template <int KIND>
__device__ void v2_chain_top(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *rowH, int *prog = 0, int penc = 0)
{
    const DevSide &a = P.a, &b = P.b;
    int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;
    const int nlast = a.left + rrt, ai = a.left - 1;
    lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG;          // origin
    rec_store_all(rowH + (size_t) b.left * G.ndw, s0, G.ndw);
    LRec prv = s0, cur = s1;
    CellLists<SList> L;
    L.as = gfq_at(a, 0, ai); L.at = gfq_at(a, 1, ai); L.ar = gfq_at(a, 2, ai);
    L.bs = L.bt = L.br = L.as;
    for (int n = b.left + 1; n <= nlast; ++n) {
        const int bi = n - 1;
        if (KIND == 2) { L.bs = gfq_at(b, 0, bi); L.bt = gfq_at(b, 1, bi); L.br = gfq_at(b, 2, bi); }
        const double pub = unpb(P, bi, ai);
        double gnp = gap_hori<KIND>(P, L, prv, G.capa);
        gnp = (n - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
        // update(h, h-1, ..., -1): fwd2c.cc:175-178 / 226-229
        ldir(cur) = isvert(ldir(prv)) ? D_NEWH : D_HORI;
        if (KIND == 2) p_newdelta(ldlb(cur, G.capa), (lu32 *) 0, L.bt, ldlb(prv, G.capa));
        p_incdelta(ldla(cur), (lu32 *) 0, ldla(prv));
        lglb(cur) = 0;
        lval(cur) = lval(prv) + gnp;
        rec_store_all(rowH + (size_t) n * G.ndw, cur, G.ndw);
        if (prog && ((n - b.left) & 31) == 0) chain_publish(prog, penc, n);
        LRec t = prv; prv = cur; cur = t;
    }
}
template <int KIND>
__device__ void v2_chain_left(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *colH, int *prog = 0, int penc = 0)
{
    const DevSide &a = P.a, &b = P.b;
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int mlast = b.left - rrl, bi = b.left - 1;
    lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG;
    LRec prv = s0, cur = s1;
    CellLists<SList> L;
    L.as = gfq_at(a, 0, a.left); L.at = L.ar = L.bs = L.bt = L.br = L.as;
    if (KIND == 2) { L.bs = gfq_at(b, 0, bi); L.bt = gfq_at(b, 1, bi); L.br = gfq_at(b, 2, bi); }
    for (int m = a.left + 1; m <= mlast; ++m) {
        const int ai = m - 1;
        L.as = gfq_at(a, 0, ai); L.at = gfq_at(a, 1, ai); L.ar = gfq_at(a, 2, ai);
        const double pua = unpa(P, ai, bi);
        double gnp = gap_vert<KIND>(P, L, prv, G.capa);
        gnp = (m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
        // update(h, h+1, ..., 1): fwd2c.cc:170-173 / 222-225
        ldir(cur) = ishori(ldir(prv)) ? D_NEWV : D_VERT;
        const int g1 = lglb(prv) + 1;
        p_newdelta(ldla(cur), (lu32 *) 0, L.at, ldla(prv));
        if (KIND == 1) lglb(cur) = g1;
        else { p_incdelta(ldlb(cur, G.capa), (lu32 *) 0, ldlb(prv, G.capa)); lglb(cur) = 0; }
        lval(cur) = lval(prv) + gnp;
        rec_store_all(colH + (size_t) (m - a.left) * G.ndw, cur, G.ndw);
        if (prog && ((m - a.left) & 31) == 0) chain_publish(prog, penc, m - a.left);
        LRec t = prv; prv = cur; cur = t;
    }
}

// ---- the same chains with their static lists staged through LDS ---------------------------------------
// One lane walks a chain, and with the lists in HBM every step of it is a string of dependent global loads
// (offset -> gap lengths -> frequencies, three times): ~8.6 us per step, 18 ms for 2049 columns -- 10 % of a DP
// sweep once the sweep is sharded over 8 GPUs.  The lists of consecutive positions are contiguous in the profile,
// so the whole wave copies them a block of PRO_B positions at a time (one latency per block instead of several per
// step) and the walking lane reads LDS only.  Arithmetic and order are those of v2_chain_top/left above.
#define PRO_B 32                       // positions per staged block
#define PRO_POOL 1024                  // entries of the two staged views of a block (shrinks the block if short)
#define PRO_CONST 128                  // entries of the three views of the chain's fixed position
struct ProLds {
    lf64 *pf, *cf, *pu; li32 *pg, *cg, *po;
    __device__ __forceinline__ void carve(lchar *base) {
        pf = (lf64 *) base; cf = pf + PRO_POOL; pu = cf + PRO_CONST;
        pg = (li32 *) (pu + PRO_B); cg = pg + PRO_POOL; po = cg + PRO_CONST;
    }
};
static const size_t PRO_LDS_BYTES = (8 * (PRO_POOL + PRO_CONST + PRO_B) + 4 * (PRO_POOL + PRO_CONST + 2 * (PRO_B + 1)) + 15) & ~(size_t) 15;

__device__ __forceinline__ void pro_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// the three views of `side` at position pos -> cg/cf; offsets of the views in co[3]
__device__ __forceinline__ void pro_stage_const(const DevSide &side, int pos, const ProLds &S, int lane, int (&co)[3])
{
    int o = 0;
    for (int v = 0; v < 3; ++v) {
        const int g0 = side.off[v][pos + 1], len = side.off[v][pos + 2] - g0;
        co[v] = o;
        for (int k = lane; k < len; k += 64) { S.cg[o + k] = side.glen[v][g0 + k]; S.cf[o + k] = side.freq[v][g0 + k]; }
        o += len;
    }
}
// views 0 (s) and 1 (t) of positions [p0, p0 + count) -> pg/pf, per-position offsets -> po[v * (PRO_B + 1) + j];
// returns the count that fits the pool
__device__ __forceinline__ int pro_stage_block(const DevSide &side, int p0, int count, const ProLds &S, int lane)
{
    int g0[2], len[2];
    for (;;) {
        for (int v = 0; v < 2; ++v) { g0[v] = side.off[v][p0 + 1]; len[v] = side.off[v][p0 + count + 1] - g0[v]; }
        if (len[0] + len[1] <= PRO_POOL || count == 1) break;
        count >>= 1;
    }
    for (int v = 0; v < 2; ++v) {
        const int base = v ? len[0] : 0;
        for (int k = lane; k < len[v]; k += 64) { S.pg[base + k] = side.glen[v][g0[v] + k]; S.pf[base + k] = side.freq[v][g0[v] + k]; }
        if (lane < count) S.po[v * (PRO_B + 1) + lane] = base + side.off[v][p0 + lane + 1] - g0[v];
    }
    return count;
}
__device__ __forceinline__ LList pro_list(const li32 *g, const lf64 *f, int o) { LList l; l.glen = g + o; l.freq = f + o; return l; }

template <int KIND>
__device__ void v2_chain_top_staged(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *rowH, const ProLds &S, const int lane, int *prog = 0, int penc = 0)
{
    const DevSide &a = P.a, &b = P.b;
    int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;
    const int nlast = a.left + rrt, ai = a.left - 1;
    int co[3];
    pro_stage_const(a, ai, S, lane, co);
    if (lane == 0) {
        lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG;          // origin
        rec_store_all(rowH + (size_t) b.left * G.ndw, s0, G.ndw);
    }
    LRec prv = s0, cur = s1;
    CellLists<LList> L;
    L.as = pro_list(S.cg, S.cf, co[0]); L.at = pro_list(S.cg, S.cf, co[1]); L.ar = pro_list(S.cg, S.cf, co[2]);
    L.bs = L.bt = L.br = L.as;
    for (int n0 = b.left + 1; n0 <= nlast; ) {
        int count = nlast - n0 + 1; if (count > PRO_B) count = PRO_B;
        pro_wave_sync();                                  // the walking lane is done with the previous block
        if (KIND == 2) count = pro_stage_block(b, n0 - 1, count, S, lane);
        if (lane < count) S.pu[lane] = unpb(P, n0 - 1 + lane, ai);
        pro_wave_sync();
        if (lane == 0) {
            for (int j = 0; j < count; ++j) {
                const int n = n0 + j;
                if (KIND == 2) { L.bs = pro_list(S.pg, S.pf, S.po[j]); L.bt = pro_list(S.pg, S.pf, S.po[PRO_B + 1 + j]); }
                const double pub = S.pu[j];
                double gnp = gap_hori<KIND>(P, L, prv, G.capa);
                gnp = (n - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
                ldir(cur) = isvert(ldir(prv)) ? D_NEWH : D_HORI;
                if (KIND == 2) p_newdelta(ldlb(cur, G.capa), (lu32 *) 0, L.bt, ldlb(prv, G.capa));
                p_incdelta(ldla(cur), (lu32 *) 0, ldla(prv));
                lglb(cur) = 0;
                lval(cur) = lval(prv) + gnp;
                rec_store_all(rowH + (size_t) n * G.ndw, cur, G.ndw);
                LRec t = prv; prv = cur; cur = t;
            }
            if (prog) chain_publish(prog, penc, n0 + count - 1);
        }
        n0 += count;
    }
}
template <int KIND>
__device__ void v2_chain_left_staged(const DevProb &P, const V2Geom &G, LRec s0, LRec s1, unsigned *colH, const ProLds &S, const int lane, int *prog = 0, int penc = 0)
{
    const DevSide &a = P.a, &b = P.b;
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int mlast = b.left - rrl, bi = b.left - 1;
    int co[3] = {0, 0, 0};
    if (KIND == 2) pro_stage_const(b, bi, S, lane, co);
    if (lane == 0) { lrec_black<KIND>(s0, G.capa); lval(s0) = 0; ldir(s0) = D_DIAG; }
    LRec prv = s0, cur = s1;
    CellLists<LList> L;
    L.as = pro_list(S.cg, S.cf, 0); L.at = L.ar = L.bs = L.bt = L.br = L.as;
    if (KIND == 2) { L.bs = pro_list(S.cg, S.cf, co[0]); L.bt = pro_list(S.cg, S.cf, co[1]); L.br = pro_list(S.cg, S.cf, co[2]); }
    for (int m0 = a.left + 1; m0 <= mlast; ) {
        int count = mlast - m0 + 1; if (count > PRO_B) count = PRO_B;
        pro_wave_sync();
        count = pro_stage_block(a, m0 - 1, count, S, lane);
        if (lane < count) S.pu[lane] = unpa(P, m0 - 1 + lane, bi);
        pro_wave_sync();
        if (lane == 0) {
            for (int j = 0; j < count; ++j) {
                const int m = m0 + j;
                L.as = pro_list(S.pg, S.pf, S.po[j]); L.at = pro_list(S.pg, S.pf, S.po[PRO_B + 1 + j]);
                const double pua = S.pu[j];
                double gnp = gap_vert<KIND>(P, L, prv, G.capa);
                gnp = (m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
                ldir(cur) = ishori(ldir(prv)) ? D_NEWV : D_VERT;
                const int g1 = lglb(prv) + 1;
                p_newdelta(ldla(cur), (lu32 *) 0, L.at, ldla(prv));
                if (KIND == 1) lglb(cur) = g1;
                else { p_incdelta(ldlb(cur, G.capa), (lu32 *) 0, ldlb(prv, G.capa)); lglb(cur) = 0; }
                lval(cur) = lval(prv) + gnp;
                rec_store_all(colH + (size_t) (m - a.left) * G.ndw, cur, G.ndw);
                LRec t = prv; prv = cur; cur = t;
            }
            if (prog) chain_publish(prog, penc, m0 + count - 1 - a.left);
        }
        m0 += count;
    }
}

// ---- one cell by one team ------------------------------------------------------------------------
struct CellSrc { LRec hd, hu, gu, g2u, hl, fl, f2l; };   // records the cell reads
struct CellDst { LRec h, g, g2, f, f2; };                 // records it writes

template <int KIND, bool NOLL3>
__device__ void v2_cell(const DevProb &P, const V2Geom &G, const CellLists<LList16> &L_, int m, int n, int lane,
                        const CellSrc &S_, const CellDst &D_, bool do_vert, bool do_hori, uint8_t *tr,
                        double dab, double pua, double pub
#ifdef G2G_V2_STAMP
                        , unsigned long long *stamp_acc, unsigned long long &stamp_t
#endif
                        )
{
    const DevSide &a = P.a, &b = P.b;
    const int capa = G.capa;
    // private VALUE copies: the per-lane operand selects below pick among these; selecting among members of a struct
    // that lives behind a reference keeps the whole struct in scratch memory (it was: 378 GB of scratch writes per sweep)
    CellLists<LList16> L; L.as = L_.as; L.at = L_.at; L.ar = L_.ar; L.bs = L_.bs; L.bt = L_.bt; L.br = L_.br;
    CellSrc S; S.hd = S_.hd; S.hu = S_.hu; S.gu = S_.gu; S.g2u = S_.g2u; S.hl = S_.hl; S.fl = S_.fl; S.f2l = S_.f2l;
    CellDst D; D.h = D_.h; D.g = D_.g; D.g2 = D_.g2; D.f = D_.f; D.f2 = D_.f2;
    // ---- phase A: one independent cost per lane.  Lanes of a wave that take different branches are
    // serialised, so the jobs are expressed as ONE call with per-lane operands (select, then call):
    //   lane 0/1 diagonal (two merges for _pf), 2 vertical from G, 3 vertical from H, 4 horizontal from F,
    //   5 horizontal from H, 6 horizontal2 from F2, 7 vertical2 from G2 (Noll 3); then lane 6: sim2.
    const int jk = (lane == 0) ? 0 : (lane == 1) ? 1 : (lane == 2 || lane == 3 || lane == 7) ? 2 : 3;
    const LRec rc = (lane < 2) ? S.hd : (lane == 2) ? S.gu : (lane == 3) ? S.hu : (lane == 4) ? S.fl
                  : (lane == 5) ? S.hl : (lane == 6) ? S.f2l : S.g2u;
    bool on = (jk == 0) || (jk == 1 && KIND == 2) || (jk == 2 && do_vert && (lane != 7 || NOLL3))
              || (jk == 3 && do_hori && (lane != 6 || NOLL3));
    double r = 0;
    if (KIND == 2) {
        const bool aside = (jk == 0 || jk == 2);        // merge a's s-list against b's t/r list, or the mirror
        const LList16 cf = aside ? L.as : L.bs;
        const LList16 df = (jk == 0) ? L.bt : (jk == 1) ? L.at : (jk == 2) ? L.br : L.ar;
        const lu32 *dlc = aside ? ldla(rc) : ldlb(rc, capa);
        const lu32 *dld = aside ? ldlb(rc, capa) : ldla(rc);
        if (on) r = p_newgap4(cf, dlc, df, dld) * P.basic_gop;
    } else {
        if (on && jk == 2) r = p_newgap1(P, L.as, ldla(rc), lglb(rc));
        if (on && jk != 2) r = p_newgap2(P, jk == 0 ? L.at : L.ar, lglb(rc), ldla(rc));
    }
    const double r8 = r;                                 // lane 6's horizontal2 cost (Noll 3)
    STAMP(2)
    const double c_d0 = __shfl(r, 0, TEAM), c_d1 = __shfl(r, 1, TEAM);
    const double c_gnpv = __shfl(r, 2, TEAM), c_gopv = __shfl(r, 3, TEAM);
    const double c_gnph = __shfl(r, 4, TEAM), c_goph = __shfl(r, 5, TEAM);
    const double c_gnpv2 = NOLL3 ? __shfl(r, 7, TEAM) : 0;
    const double c_gnph2 = NOLL3 ? __shfl(r8, 6, TEAM) : 0;
    // ---- scalar decisions, replayed by every lane (fwd2c.h:395-453) ------------------------------
    double gop = (KIND == 2) ? c_d0 + c_d1 : c_d0;
    const double hval = lval(S.hd) + (dab + gop);
    const int hdir = isdiag(ldir(S.hd)) ? D_DIAG : D_NEWD;
    int bits = 0, win = 0;                          // win: 0 diag, 1 G, 2 G2, 3 F, 4 F2
    double mxval = NEVSEL;                          // mx = g: at the first row G is a black record
    double gval = 0, g2val = 0, fval = 0, f2val = 0;
    int gdir = 0, g2dir = 0, fdir = 0, f2dir = 0;
    bool g_from_h = false, g2_from_h = false, f_from_h = false, f2_from_h = false;
    if (do_vert) {
        const double gnp = c_gnpv;
        gop = c_gopv;
        const bool hu_nv = !isvert(ldir(S.hu));
        g_from_h = hu_nv && (lval(S.hu) + gop > lval(S.gu) + gnp);
        const LRec gs = g_from_h ? S.hu : S.gu;
        gdir = ishori(ldir(gs)) ? D_NEWV : D_VERT;
        gval = lval(gs) + (g_from_h ? gop : gnp);
        gval += pua;
        if (!g_from_h) bits |= T_GEXT;
        mxval = gval; win = 1;
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c_gnpv2;
            gop = P.v2divv1 * gop;
            g2_from_h = hu_nv && (lval(S.hu) + gop > lval(S.g2u) + gnp2);
            const LRec gs2 = g2_from_h ? S.hu : S.g2u;
            g2dir = ishori(ldir(gs2)) ? D_NEWV : D_VERT;
            g2val = lval(gs2) + (g2_from_h ? gop : gnp2);
            g2val += P.u2divu1 * pua;
            if (!g2_from_h) bits |= T_G2EXT;
            if (g2val > mxval) { mxval = g2val; win = 2; }
        }
    } else {
        mxval = NEVSEL; win = 1;                     // first row: mx starts as the untouched black G
    }
    if (do_hori) {
        const double gnp = c_gnph;
        gop = c_goph;
        const bool hl_nh = !ishori(ldir(S.hl));
        f_from_h = hl_nh && (lval(S.hl) + gop > lval(S.fl) + gnp);
        const LRec fs = f_from_h ? S.hl : S.fl;
        fdir = isvert(ldir(fs)) ? D_NEWH : D_HORI;
        fval = lval(fs) + (f_from_h ? gop : gnp);
        fval += pub;
        if (!f_from_h) bits |= T_FEXT;
        if (fval >= mxval) { mxval = fval; win = 3; }
        if (NOLL3) {
            const double gnp2 = P.v2divv1 * c_gnph2;
            gop = P.v2divv1 * gop;
            f2_from_h = hl_nh && (lval(S.hl) + gop > lval(S.f2l) + gnp2);
            const LRec fs2 = f2_from_h ? S.hl : S.f2l;
            f2dir = isvert(ldir(fs2)) ? D_NEWH : D_HORI;
            f2val = lval(fs2) + (f2_from_h ? gop : gnp2);
            f2val += P.u2divu1 * pub;
            if (!f2_from_h) bits |= T_F2EXT;
            if (f2val >= mxval) { mxval = f2val; win = 4; }
        }
    }
    if (!(mxval > hval)) win = 0;                      // diagonal wins ties (fwd2c.h:453)
    if (!do_vert && win == 1) win = 0;                 // (the black G can never win)
    STAMP(3)
    // ---- phase B: list updates, one per lane; the winner's lists are also written to the new H ----
    team_sync();
    const LRec gs = g_from_h ? S.hu : S.gu, gs2 = g2_from_h ? S.hu : S.g2u;
    const LRec fs = f_from_h ? S.hl : S.fl, fs2 = f2_from_h ? S.hl : S.f2l;
    lu32 *const nul = (lu32 *) 0;
    {   // newdelta jobs (one call, per-lane operands), then incdelta jobs (one call)
        lu32 *d1 = nul, *d2 = nul; const lu32 *sp = nul; LList16 df = L.at; bool go = false;
        if (KIND == 2) {
            switch (lane) {
            case 0: go = win == 0; d1 = ldla(D.h); sp = ldla(S.hd); break;
            case 1: go = win == 0; d1 = ldlb(D.h, capa); sp = ldlb(S.hd, capa); df = L.bt; break;
            case 2: go = do_vert; d1 = ldla(D.g); d2 = win == 1 ? ldla(D.h) : nul; sp = ldla(gs); break;
            case 4: go = do_hori; d1 = ldlb(D.f, capa); d2 = win == 3 ? ldlb(D.h, capa) : nul; sp = ldlb(fs, capa); df = L.bt; break;
            case 6: go = NOLL3 && do_vert; d1 = ldla(D.g2); d2 = win == 2 ? ldla(D.h) : nul; sp = ldla(gs2); break;
            case 7: go = NOLL3 && do_hori; d1 = ldlb(D.f2, capa); d2 = win == 4 ? ldlb(D.h, capa) : nul; sp = ldlb(fs2, capa); df = L.bt; break;
            default: break;
            }
        } else {
            switch (lane) {
            case 0: go = win == 0; d1 = ldla(D.h); sp = ldla(S.hd); break;
            case 2: go = do_vert; d1 = ldla(D.g); d2 = win == 1 ? ldla(D.h) : nul; sp = ldla(gs); break;
            case 6: go = NOLL3 && do_vert; d1 = ldla(D.g2); d2 = win == 2 ? ldla(D.h) : nul; sp = ldla(gs2); break;
            default: break;
            }
        }
        if (go) p_newdelta(d1, d2, df, sp);
        go = false; d2 = nul;
        if (KIND == 2) {
            switch (lane) {
            case 3: go = do_vert; d1 = ldlb(D.g, capa); d2 = win == 1 ? ldlb(D.h, capa) : nul; sp = ldlb(gs, capa); break;
            case 5: go = do_hori; d1 = ldla(D.f); d2 = win == 3 ? ldla(D.h) : nul; sp = ldla(fs); break;
            case 0: go = NOLL3 && do_vert; d1 = ldlb(D.g2, capa); d2 = win == 2 ? ldlb(D.h, capa) : nul; sp = ldlb(gs2, capa); break;
            case 1: go = NOLL3 && do_hori; d1 = ldla(D.f2); d2 = win == 4 ? ldla(D.h) : nul; sp = ldla(fs2); break;
            default: break;
            }
        } else {
            switch (lane) {
            case 4: go = do_hori; d1 = ldla(D.f); d2 = win == 3 ? ldla(D.h) : nul; sp = ldla(fs); break;
            case 7: go = NOLL3 && do_hori; d1 = ldla(D.f2); d2 = win == 4 ? ldla(D.h) : nul; sp = ldla(fs2); break;
            default: break;
            }
        }
        if (go) p_incdelta(d1, d2, sp);
    }
    STAMP(4)
    // scalars of the produced records (lane 0 writes; glb per fwd2c.cc:169,173,177)
    const int glb_g = (KIND == 1 && do_vert) ? lglb(gs) + 1 : 0;
    const int glb_g2 = (KIND == 1 && NOLL3 && do_vert) ? lglb(gs2) + 1 : 0;
    team_sync();
    if (lane == 0) {
        if (do_vert) { lval(D.g) = gval; ldir(D.g) = gdir; lglb(D.g) = glb_g; }
        if (NOLL3 && do_vert) { lval(D.g2) = g2val; ldir(D.g2) = g2dir; lglb(D.g2) = glb_g2; }
        if (do_hori) { lval(D.f) = fval; ldir(D.f) = fdir; lglb(D.f) = 0; }
        if (NOLL3 && do_hori) { lval(D.f2) = f2val; ldir(D.f2) = f2dir; lglb(D.f2) = 0; }
        double hv = hval; int hd = hdir, hg = 0;
        if (win == 1) { hv = gval; hd = gdir; hg = glb_g; }
        else if (win == 2) { hv = g2val; hd = g2dir; hg = glb_g2; }
        else if (win == 3) { hv = fval; hd = fdir; }
        else if (win == 4) { hv = f2val; hd = f2dir; }
        lval(D.h) = hv; ldir(D.h) = hd; lglb(D.h) = hg;
        if (win == 2 || win == 4) bits |= T_SEL2;
        *tr = (uint8_t) (bits | dir2code(hd));
    }
    team_sync();
}

// copy one static list (with its terminator) from the HBM pool into an LDS slot, `nl` lanes sharing it
__device__ __forceinline__ void list_g2l(li16 *gl, lf64 *fr, const DevSide &s, int view, int pos, int lane, int nl)
{
    const int o = s.off[view][pos + 1], e = s.off[view][pos + 2];
    const int *sg = s.glen[view] + o;
    const double *sf = s.freq[view] + o;
    for (int k = lane; k < e - o; k += nl) { gl[k] = (short) sg[k]; fr[k] = sf[k]; }
}

// ---- a wave-uniform private copy of the DP descriptor -------------------------------------------------
// The descriptor lives in HBM next to buffers the kernel writes, so the compiler must assume every store may
// change it: left alone, the sweep re-reads its fields with vector loads -- and a full vmcnt(0) wait -- in the
// middle of every step.  The fields the sweep uses are therefore copied once per tile into scalar registers.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uni(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
template <class T> __device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v >> 32));
    return (T *) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ void uni_side(DevSide &d, const DevSide &s)
{
    d.many = uni(s.many); d.len = uni(s.len); d.left = uni(s.left); d.right = uni(s.right); d.nils = uni(s.nils);
    d.nelm = uni(s.nelm); d.felm = uni(s.felm); d.hetero = uni(s.hetero); d.maxlist = uni(s.maxlist);
    d.seq = uni(s.seq); d.weight = uni(s.weight); d.pseq = uni(s.pseq); d.thk = uni(s.thk);
    for (int v = 0; v < 3; ++v) { d.off[v] = uni(s.off[v]); d.glen[v] = uni(s.glen[v]); d.freq[v] = uni(s.freq[v]); }
    d.gapdens = uni(s.gapdens); d.postgapdens = uni(s.postgapdens);
}
__device__ __forceinline__ void uni_prob(DevProb &d, const DevProb &s)
{
    d.kind = uni(s.kind); d.noll = uni(s.noll); d.sim2_kind = uni(s.sim2_kind); d.crg2_kind = uni(s.crg2_kind);
    d.codonk1 = uni(s.codonk1); d.lw = uni(s.lw); d.up = uni(s.up); d.width = uni(s.width);
    d.basic_gop = uni(s.basic_gop); d.weighted_gop = uni(s.weighted_gop); d.u = uni(s.u);
    d.u2divu1 = uni(s.u2divu1); d.v2divv1 = uni(s.v2divv1);
    d.simmtx = uni(s.simmtx); d.simdim = uni(s.simdim); d.capa = uni(s.capa); d.capb = uni(s.capb);
    uni_side(d.a, s.a); uni_side(d.b, s.b);
    d.v2_rowH = uni(s.v2_rowH); d.v2_rowG = uni(s.v2_rowG); d.v2_rowG2 = uni(s.v2_rowG2); d.v2_colH = uni(s.v2_colH);
    d.v2_cbH = uni(s.v2_cbH); d.v2_cbF = uni(s.v2_cbF); d.v2_cbF2 = uni(s.v2_cbF2);
    d.v2_rowstride = uni(s.v2_rowstride); d.v2_sim = uni(s.v2_sim); d.v2_rowoff = uni(s.v2_rowoff);
    d.trace = uni(s.trace); d.tstride = uni(s.tstride); d.d0 = uni(s.d0); d.d1 = uni(s.d1);
    d.score = uni(s.score);
}

// ---- strip-local column scores (sweep mode) ----------------------------------------------------------------------
// PwdM::sim2 of a cell does not depend on the recurrence.  Tile mode reads it from a matrix a separate kernel fills ahead
// of the sweep (8 B per cell of HBM, written once and read once).  A strip in sweep mode makes its own instead, block by
// block: every 64 steps it computes the (rows of the strip) x 64 columns its first row enters 64 steps later -- thread <->
// column, rows in a loop, so a row's profile vector is a wave-uniform read and the stores are coalesced -- into one of
// THREE 32 KB buffers of a per-workgroup scratch area (the rows of a skewed strip straddle two blocks while the third is
// filled).  The scratch area is reused by every strip the workgroup runs: ~100 MB per launch, cache resident, instead of
// 8 B per cell of the sweep.
#define GLBV3 __attribute__((address_space(1)))
#ifndef G2G_SIMBLK_STRIDE
#define G2G_SIMBLK_STRIDE (3 * 4096)          // doubles of column-score scratch per workgroup: three blocks of 64 x 64
#endif
struct SimBlk { GLBV3 double *buf; int cbase; };
// (tid / nthr / rows: the filling threads -- one wave for the one-lane-per-cell kernels, the workgroup for v2 -- and the
// rows of a strip; thread <-> column tid & 63, rows tid >> 6, tid >> 6 + nthr / 64, ...)
__device__ __forceinline__ void simblk_fill(const DevProb &P, const SimBlk &S, const int bk, const int m0, const int tid,
                                            const int nthr = 64, const int rows = 64)
{
    const int n = S.cbase + bk * 64 + (tid & 63);
    if (bk < 0 || n >= P.b.right) return;
    GLBV3 double *dst = S.buf + (size_t) (bk % 3) * 4096 + (tid & 63);
    for (int r = tid >> 6; r < rows; r += nthr >> 6) {
        const int m = m0 + r;
        if (m >= P.a.right) break;
        int nlo = m + P.lw; if (nlo < P.b.left) nlo = P.b.left;
        int nhi = m + P.up + 1; if (nhi > P.b.right) nhi = P.b.right;
        if (n >= nlo && n < nhi) dst[r * 64] = sim2(P, m, n);
    }
}
__device__ __forceinline__ const GLBV3 double *simblk_at(const SimBlk &S, const int row, const int n)
{
    const int k = n - S.cbase;
    return S.buf + (size_t) ((k >> 6) % 3) * 4096 + row * 64 + (k & 63);
}

// ---- one TILE = (strip of R rows) x (block of C columns) by one workgroup ------------
// Tiles of a DP depend on their upper, left and upper-left neighbours only, so all tiles with the same
// i + j (over every DP of the batch) run in one launch; a big DP is spread over many workgroups instead
// of bounding the sweep time.  What crosses tile borders lives in HBM:
//   rowH/rowG/rowG2[3][col]  the last row's corners of a strip (3 buffers: strip i writes i % 3, reads
//                            (i + 2) % 3; the top boundary chain is "strip -1" and writes buffer 2).
//                            Tile (i, j) overwrites what tile (i-2, j+1) still reads as its first diagonal
//                            corner, hence the extra write-after-read dependency dep_war.
//   cbH/cbF/cbF2[row]        each row's H corner and F records at the block's right edge
//   colH[row]                the left boundary chain
template <int KIND, bool NOLL3>
__device__ __forceinline__ void v2_tile(const DevProb &Pmem, lchar *lds, int ti, int tj, int nsteps, const int C,
                        const int *prog_up = 0, int *prog_self = 0, int *dbg = 0, const int pgen = 0, const int pint = 32,
                        const int *prog_left = 0, double *simscr = 0, int *failp = 0)
{
    // SWEEP MODE (prog_self != 0), as in g2g_kernels_v3.hip: the tile is a whole strip; the strip above publishes every
    // 32 steps up to which corner column its last row's records are in HBM, this strip waits only before its first
    // team reads beyond what it has seen published.  Everything is executed uniformly by all threads of the workgroup.
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int tid = threadIdx.x, lane = tid & (TEAM - 1), team = tid / TEAM;
    V2Geom G;
    G.capa = P.capa; G.capb = (KIND == 2) ? P.capb : 0;
    G.recsz = (16 + 4 * (G.capa + G.capb) + 15) & ~15; G.ndw = G.recsz / 4;
    G.nslot = NOLL3 ? 9 : 6;
    G.R = blockDim.x / TEAM;
    G.lds = lds;
    const int R = G.R, RC = R + 2;                         // RC: slots of the column ring
    const int mla = P.a.maxlist, mlb = (KIND == 2) ? P.b.maxlist : 0;
    lchar *q = lds + R * (G.nslot * G.recsz + 16) + EX_N * G.recsz;
    lf64 *afreq = (lf64 *) q;            q += sizeof(double) * (size_t) R * 3 * mla;
    lf64 *bfreq = (lf64 *) q;            q += sizeof(double) * (size_t) RC * 3 * mlb;
    li16 *aglen = (li16 *) q;            q += sizeof(short) * (size_t) R * 3 * mla;
    li16 *bglen = (li16 *) q;
    const size_t rbuf = (size_t) P.v2_rowstride * G.ndw;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const unsigned *rowHp = (const unsigned *) P.v2_rowH + bprev * rbuf, *rowGp = (const unsigned *) P.v2_rowG + bprev * rbuf;
    const unsigned *rowG2p = NOLL3 ? (const unsigned *) P.v2_rowG2 + bprev * rbuf : 0;
    unsigned *rowHc = (unsigned *) P.v2_rowH + bcur * rbuf, *rowGc = (unsigned *) P.v2_rowG + bcur * rbuf;
    unsigned *rowG2c = NOLL3 ? (unsigned *) P.v2_rowG2 + bcur * rbuf : 0;
    const unsigned *colH = (const unsigned *) P.v2_colH;
    unsigned *cbH = (unsigned *) P.v2_cbH, *cbF = (unsigned *) P.v2_cbF, *cbF2 = (unsigned *) P.v2_cbF2;
    if (tid == 0) lrec_black<KIND>(G.extra(EX_BLACK), G.capa);
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const LRec black = G.extra(EX_BLACK);
    const int m0 = a.left + ti * R, m = m0 + team;
    // markers for the time-out report of whoever ends up waiting for this strip (g2g_wait_ge): taken from the queue, by which workgroup
    if (prog_self) __hip_atomic_store(prog_self + G2G_DIAG + 5, ((pgen & 0x7FF) << 20) | 0x20000 | (int) (blockIdx.x & 0xFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prog_left) {                                       // sweep mode: the left boundary chain runs beside the strips (v2_chain_tile)
        const int rows_ = m0 + R - a.left;
        const int wantl = ((pgen & 0x7FF) << 20) | (rows_ < 0xFFFFF ? rows_ : 0xFFFFF);
        (void) g2g_wait_ge(prog_left, wantl, dbg, failp, ti);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (prog_self) __hip_atomic_store(prog_self + G2G_DIAG + 6, ((pgen & 0x7FF) << 20) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ... past the left chain
    G2G_HB_STEP(prog_self, tid >> 6, -1)
    G2G_HB(prog_self, tid >> 6, 10)                         // places 10-18: the strip's prologue
    const int c0 = b.left + tj * C;
    int c1 = c0 + C; if (c1 > b.right) c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo > c0 ? nlo : c0, hi = nhi < c1 ? nhi : c1;      // ... clipped to this block
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left; if (cbase < c0) cbase = c0;
    // every ring slot black (reset(f1), reset(f2), fwd2c.h:385-386; G of the DP's first row is never
    // written and is read as black by the row below, fwd2c.h:401)
    if (lane < G.nslot) lrec_black<KIND>(G.row(team, lane), G.capa);
    if (lane == 0 && G.nslot > TEAM) lrec_black<KIND>(G.row(team, 8), G.capa);
    team_sync();
    G2G_HB(prog_self, tid >> 6, 11)
    // rows that continue from the block on the left: their corner at the block edge and their F
    if (row_ok && c0 - 1 >= nlo && c0 - 1 < nhi) {
        rec_g2l(G.row(team, SLOT_H(c0)), cbH + (size_t) (m - a.left) * G.ndw, G.ndw, lane);
        if (lo < hi) {
            rec_g2l(G.row(team, SLOT_F), cbF + (size_t) (m - a.left) * G.ndw, G.ndw, lane);
            if (NOLL3) rec_g2l(G.row(team, SLOT_F2), cbF2 + (size_t) (m - a.left) * G.ndw, G.ndw, lane);
        }
    }
    G2G_HB(prog_self, tid >> 6, 12)
    // this row's static lists -> LDS (they serve every cell of the row)
    CellLists<LList16> L;
    {
        li16 *ag = aglen + (size_t) team * 3 * mla;
        lf64 *af = afreq + (size_t) team * 3 * mla;
        if (row_ok) for (int v = 0; v < 3; ++v) list_g2l(ag + v * mla, af + v * mla, a, v, m, lane, TEAM);
        L.as.glen = ag; L.as.freq = af;
        L.at.glen = ag + mla; L.at.freq = af + mla;
        L.ar.glen = ag + 2 * mla; L.ar.freq = af + 2 * mla;
        L.bs = L.bt = L.br = L.as;
    }
    G2G_HB(prog_self, tid >> 6, 13)
    // the view (s/t/r) of b this thread prefetches: chosen once with selects (a runtime index into the descriptor copy
    // would force the copy into scratch memory)
    // (the loader lanes sit in the LAST wave: the first wave already carries the staging of the strip above)
    const int pf_tid = tid - ((int) blockDim.x - 64);
    const int pf_view = pf_tid / TEAM;
    const int *const pf_off = pf_view == 0 ? b.off[0] : pf_view == 1 ? b.off[1] : b.off[2];
    const int *const pf_glen = pf_view == 0 ? b.glen[0] : pf_view == 1 ? b.glen[1] : b.glen[2];
    const double *const pf_freq = pf_view == 0 ? b.freq[0] : pf_view == 1 ? b.freq[1] : b.freq[2];
    if (KIND == 2 && pf_tid >= 0 && pf_tid < 3 * TEAM) {   // column ring: column cbase for step 0
        const int v = pf_view;
        const int o = pf_off[cbase + 1], e = pf_off[cbase + 2];
        for (int k = lane; k < e - o; k += TEAM) { bglen[(size_t) v * mlb + k] = (short) pf_glen[o + k]; bfreq[(size_t) v * mlb + k] = pf_freq[o + k]; }
    }
    G2G_HB(prog_self, tid >> 6, 14)
    // per-row constants and one-step-ahead register pipelines (column score, b's column thickness,
    // and -- for the strip's first row -- the upper neighbours' records): nothing that comes from HBM is
    // waited for inside the step that uses it
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    int nf0 = m + P.lw; if (nf0 < b.left) nf0 = b.left;
    const double pua_row = row_ok ? unpa(P, m, nf0) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    G2G_HB(prog_self, tid >> 6, 15)
    const bool own_sim = prog_self != 0 && simscr != 0;    // sweep mode: the strip makes its column scores block by block (SimBlk)
    const double *simrow = (row_ok && !own_sim) ? P.v2_sim + P.v2_rowoff[m - a.left] - nlo : 0;
    SimBlk SB; SB.buf = (GLBV3 double *) simscr; SB.cbase = cbase;
    if (own_sim) { simblk_fill(P, SB, 0, m0, tid, blockDim.x, R); G2G_HB(prog_self, tid >> 6, 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    G2G_HB(prog_self, tid >> 6, 17)
    double sim_cur = 0, bc_cur = 0;
    bool have = false;
    int rslot = (RC - team % RC) % RC;                     // ring slot of column cbase + s - team
    int wslot = 1 % RC;                                    // ring slot of column cbase + s + 1
    const bool stage_regs = G.ndw <= 64;                   // the records of the strip above are fetched a step ahead by the first wave, one dword per lane
    int avail = prog_up ? 0 : 0x7fffffff;
    const int penc = (pgen & 0x7FF) << 20;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right; if (hi0 > c1) hi0 = c1;    // the first team's hi
    const int tl = ((m0 + R < a.right) ? m0 + R : a.right) - 1 - m0;                          // team of the strip's last row
    auto need = [&](int col) {
        if (col > hi0 + 1) col = hi0 + 1;
        const int want = penc | (col < 0 ? 0 : col < 0xFFFFF ? col : 0xFFFFF);
        if (prog_up && want > avail) {
            avail = g2g_wait_ge(prog_up, want, dbg, failp, ti);
            G2G_ACQUIRE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto publish = [&](const int col) {                    // (a barrier inside: call it uniformly)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G2G_RELEASE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // every wave leaves the column it publishes at before it enters the barrier (all lanes, same word, same value): if this
        // workgroup ever stops here, the report of whoever waits for it shows which wave did not arrive (g2g_wait_ge, hdr[42..45])
        __hip_atomic_store(prog_self + G2G_DIAG + 8 + (tid >> 6), col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        G2G_HB(prog_self, tid >> 6, 2)
        __syncthreads();
        G2G_HB(prog_self, tid >> 6, 3)
        G2G_POST(prog_self, penc | (col < 0 ? 0 : col < 0xFFFFF ? col : 0xFFFFF));
    };
    need(cbase + 2);
    G2G_HB(prog_self, tid >> 6, 18)
    if (prog_self) __hip_atomic_store(prog_self + G2G_DIAG + 7, ((pgen & 0x7FF) << 20) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ... past the first look at the strip above
    if (prog_self && tid < 64) {                           // (the whole first wave, same value: no one-lane branch) where this strip runs: for the time-out report of whoever waits for it (g2g_wait_ge)
        __hip_atomic_store(prog_self + G2G_DIAG + 3, (int) __builtin_amdgcn_s_getreg((31 << 11) | 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(prog_self + G2G_DIAG + 4, 0x100 | ((int) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef G2G_V2_STAMP
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        STAMP(7)
        const int n = cbase + s - team;
        const bool active = row_ok && n >= lo && n < hi;
        G2G_HB_STEP(prog_self, tid >> 6, s)
        G2G_HB(prog_self, tid >> 6, 1)
        if (prog_self && s > 0 && (s & (pint - 1)) == 0) publish(cbase + s - tl);
        G2G_HB(prog_self, tid >> 6, 4)
        need(cbase + s + 2);
        if (own_sim && (s & 63) == 0) {                    // the block the first row enters 64 steps from now; visible to the
            simblk_fill(P, SB, (s >> 6) + 1, m0, tid, blockDim.x, R);      // whole workgroup long before (a barrier every step)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G2G_HB(prog_self, tid >> 6, 5)
        double sim_nx = 0, bc_nx = 0;
        if (active) {
            if (!have) { sim_cur = own_sim ? (double) *simblk_at(SB, team, n) : simrow[n]; bc_cur = thk_at(b, n)[0]; }
            if (n + 1 < hi) { sim_nx = own_sim ? (double) *simblk_at(SB, team, n + 1) : simrow[n + 1]; bc_nx = thk_at(b, n + 1)[0]; }
        }
        // prefetch the column the first row reaches next step into the ring (3 teams, one view each): the
        // loads are issued here, the LDS stores wait until the cell work of this step is done
        int pf_g[4]; double pf_f[4]; int pf_n = 0, pf_base = 0;
        const bool pf_on = KIND == 2 && pf_tid >= 0 && pf_tid < 3 * TEAM && cbase + s + 1 < c1;
        if (pf_on) {
            const int v = pf_view, pos = cbase + s + 1;
            const int o = pf_off[pos + 1];
            pf_n = pf_off[pos + 2] - o;
            pf_base = (wslot * 3 + v) * mlb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + j * TEAM;
                if (k < pf_n) { pf_g[j] = pf_glen[o + k]; pf_f[j] = pf_freq[o + k]; }
            }
        }
        // the first row's upper neighbours for the column after next: fetched now by the whole first wave (lane k: dword k of each
        // record), parked in LDS after this step's cells -- nothing that comes from HBM is waited for in the step that uses it.
        // (Four dwords per lane of the first team only covered records of 128 bytes; with longer ones that row loaded three records
        //  synchronously in every step, and the other rows of the strip waited for it at the barrier.)
        const int n0 = cbase + s;                          // the first row's column in this step
        bool pf_h = false, pf_gu = false;
        unsigned pfh1 = 0, pfgu1 = 0, pfg2u1 = 0;
        if (stage_regs && m0 < a.right && n0 < hi0 && n0 + 1 < hi0) {
            const bool vert0 = m0 > a.left;
            int nhi0 = m0 + P.up + 1; if (nhi0 > b.right) nhi0 = b.right;
            const bool up_nx = vert0 && (n0 + 1 - (m0 - 1) <= P.up);
            pf_h = up_nx || (!vert0 && n0 + 2 < nhi0);
            pf_gu = up_nx;
            if (tid < 64 && tid < G.ndw) {
                if (pf_h) pfh1 = G2G_XLD(rowHp + (size_t) (n0 + 2) * G.ndw + tid);
                if (pf_gu) pfgu1 = G2G_XLD(rowGp + (size_t) (n0 + 2) * G.ndw + tid);
                if (NOLL3 && pf_gu) pfg2u1 = G2G_XLD(rowG2p + (size_t) (n0 + 2) * G.ndw + tid);
            }
        }
        if (active) {
            CellSrc S; CellDst D;
            const bool do_vert = m > a.left, do_hori = n > b.left;
            if (KIND == 2) {
                const int slot = rslot;
                const li16 *bg = bglen + (size_t) slot * 3 * mlb;
                const lf64 *bf = bfreq + (size_t) slot * 3 * mlb;
                L.bs.glen = bg; L.bs.freq = bf;
                L.bt.glen = bg + mlb; L.bt.freq = bf + mlb;
                L.br.glen = bg + 2 * mlb; L.br.freq = bf + 2 * mlb;
            }
            // -- sources -----------------------------------------------------------------------
            if (team == 0) {
                // the row above lives in HBM (previous strip's last row / the top boundary chain); its
                // records for the NEXT column are fetched into registers now and parked in LDS after the cell
                const LRec hs0 = G.extra(EX_H0), hs1 = G.extra(EX_H1), gu = G.extra(EX_GU), g2u = G.extra(EX_G2U);
                const LRec hcur = (n & 1) ? hs1 : hs0, hnxt = (n & 1) ? hs0 : hs1;
                const bool up_in = do_vert && (n - (m - 1) <= P.up);       // cell (m-1, n) exists
                if (n == lo || !stage_regs) {                               // first cell of the row in this block
                    if (n == lo) {
                        if (n == b.left && m > a.left) rec_g2l_x(hcur, colH + (size_t) (m - a.left) * G.ndw, G.ndw, lane);
                        else rec_g2l_x(hcur, rowHp + (size_t) n * G.ndw, G.ndw, lane);
                    }
                    if (up_in || (!do_vert && n + 1 < nhi)) rec_g2l_x(hnxt, rowHp + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                    if (up_in) {
                        rec_g2l_x(gu, rowGp + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                        if (NOLL3) rec_g2l_x(g2u, rowG2p + (size_t) (n + 1) * G.ndw, G.ndw, lane);
                    }
                }
                team_sync();
                S.hd = hcur; S.hu = up_in ? hnxt : black; S.gu = up_in ? gu : black; S.g2u = up_in ? g2u : black;
            } else {
                S.hd = G.row(team - 1, SLOT_H(n));
                const bool up_in = (n - (m - 1) <= P.up);
                S.hu = up_in ? G.row(team - 1, SLOT_H(n + 1)) : black;
                S.gu = up_in ? G.row(team - 1, SLOT_G(n + 1)) : black;
                S.g2u = (NOLL3 && up_in) ? G.row(team - 1, SLOT_G2(n + 1)) : black;
            }
            const bool left_in = (n - 1 - m >= P.lw);                      // cell (m, n-1) exists
            S.hl = left_in ? G.row(team, SLOT_H(n)) : black;
            S.fl = left_in ? G.row(team, SLOT_F) : black;
            S.f2l = (NOLL3 && left_in) ? G.row(team, SLOT_F2) : black;
            D.h = G.row(team, SLOT_H(n + 1));
            D.g = G.row(team, SLOT_G(n + 1));
            D.g2 = G.row(team, NOLL3 ? SLOT_G2(n + 1) : SLOT_G(n + 1));
            D.f = G.row(team, SLOT_F);
            D.f2 = G.row(team, NOLL3 ? SLOT_F2 : SLOT_F);
            int mlo, mhi;
            const int d = m + n;
            diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            uint8_t *tr = P.trace + (size_t) (d - P.d0) * P.tstride + (m - mlo);
            STAMP(1)
            G2G_HB(prog_self, tid >> 6, 6)
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            v2_cell<KIND, NOLL3>(P, G, L, m, n, lane, S, D, do_vert, do_hori, tr, sim_cur, pua, pub
#ifdef G2G_V2_STAMP
                                 , stamp_acc, stamp_t
#endif
                                 );
            STAMP(5)
            G2G_HB(prog_self, tid >> 6, 7)
            sim_cur = sim_nx; bc_cur = bc_nx; have = (n + 1 < hi);
            // the row below starts at b.left with the left-boundary corner (m+1, b.left) as its
            // diagonal source: park it in this row's H ring where that row will look for it
            if (n == b.left && team + 1 < R && m + 1 < a.right && m + 1 <= m_left_last && (m + 1 + P.lw) <= b.left) {
                rec_g2l_x(G.row(team, SLOT_H(b.left)), colH + (size_t) (m + 1 - a.left) * G.ndw, G.ndw, lane);
            }
            // strip boundary: the last row's corners go to HBM for the strip below
            if (team == R - 1 || m == a.right - 1) {
                rec_l2g_x(rowHc + (size_t) (n + 1) * G.ndw, D.h, G.ndw, lane);
                rec_l2g_x(rowGc + (size_t) (n + 1) * G.ndw, D.g, G.ndw, lane);
                if (NOLL3) rec_l2g_x(rowG2c + (size_t) (n + 1) * G.ndw, D.g2, G.ndw, lane);
            }
            // block boundary: this row's corner and F records for the block on the right
            if (n == c1 - 1 && c1 < b.right) {
                rec_l2g(cbH + (size_t) (m - a.left) * G.ndw, D.h, G.ndw, lane);
                rec_l2g(cbF + (size_t) (m - a.left) * G.ndw, D.f, G.ndw, lane);
                if (NOLL3) rec_l2g(cbF2 + (size_t) (m - a.left) * G.ndw, D.f2, G.ndw, lane);
            }
            if (m == a.right - 1 && n == b.right - 1 && lane == 0) *P.score = lval(D.h);
        }
        if ((pf_h || pf_gu) && tid < 64 && tid < G.ndw) {      // park the prefetched neighbours (corner n0 + 2 shares corner n0's slot)
            const LRec hslot = (n0 & 1) ? G.extra(EX_H1) : G.extra(EX_H0);
            if (pf_h) ((lu32 *) hslot.p)[tid] = pfh1;
            if (pf_gu) ((lu32 *) G.extra(EX_GU).p)[tid] = pfgu1;
            if (NOLL3 && pf_gu) ((lu32 *) G.extra(EX_G2U).p)[tid] = pfg2u1;
        }
        if (pf_on) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + j * TEAM;
                if (k < pf_n) { bglen[pf_base + k] = (short) pf_g[j]; bfreq[pf_base + k] = pf_f[j]; }
            }
            if (pf_n > 4 * TEAM) {                          // (lists longer than 32 entries: straight copy)
                const int pos = cbase + s + 1, o = pf_off[pos + 1];
                for (int k = lane + 4 * TEAM; k < pf_n; k += TEAM) { bglen[pf_base + k] = (short) pf_glen[o + k]; bfreq[pf_base + k] = pf_freq[o + k]; }
            }
        }
        if (++rslot == RC) rslot = 0;
        if (++wslot == RC) wslot = 0;
        STAMP(6)
        G2G_HB(prog_self, tid >> 6, 8)
        __syncthreads();
    }
    if (prog_self) publish(0xFFFFF);
#ifdef G2G_V2_STAMP
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g2g_stamp_acc[k + (threadIdx.x == 0 ? 0 : 8)], stamp_acc[k]);
#endif
}

// boundary chains of every DP: one small workgroup each (wave 0: top row, wave 1: left column); they are
// latency-bound single-lane chains and run on their own stream beside the (fully parallel) score kernel
template <int KIND>
__device__ void v2_prologue(const DevProb &Pmem, lchar *lds, const int pro_off)
{
    DevProb P;
    uni_prob(P, Pmem);                                   // descriptor in scalar registers (see uni_prob)
    V2Geom G;
    G.capa = P.capa; G.capb = (KIND == 2) ? P.capb : 0;
    G.recsz = (16 + 4 * (G.capa + G.capb) + 15) & ~15; G.ndw = G.recsz / 4;
    G.nslot = 1; G.R = 4; G.lds = lds;
    unsigned *rowH2 = (unsigned *) P.v2_rowH + 2 * (size_t) P.v2_rowstride * G.ndw;
    const int mlx = P.a.maxlist > P.b.maxlist ? P.a.maxlist : P.b.maxlist;
    if (pro_off && 3 * mlx <= PRO_CONST && 2 * mlx <= PRO_POOL) {         // lists staged through LDS, a wave per chain
        ProLds S; S.carve(lds + pro_off + (threadIdx.x >> 6) * PRO_LDS_BYTES);
        if (threadIdx.x < 64) v2_chain_top_staged<KIND>(P, G, G.row(0, 0), G.row(1, 0), rowH2, S, threadIdx.x);
        else v2_chain_left_staged<KIND>(P, G, G.row(2, 0), G.row(3, 0), (unsigned *) P.v2_colH, S, threadIdx.x - 64);
        return;
    }
    if (threadIdx.x == 0) v2_chain_top<KIND>(P, G, G.row(0, 0), G.row(1, 0), rowH2);
    if (threadIdx.x == 64) v2_chain_left<KIND>(P, G, G.row(2, 0), G.row(3, 0), (unsigned *) P.v2_colH);
}


// A boundary chain as an entry of a persistent kernel's queue (sweep mode): which = -1 top row, -2 left column, walked by
// the first wave of the workgroup.  Chain entries head the queue and wait for nothing, so the strips that poll their
// progress counters (strip 0: the top chain as its "strip above"; every strip: the left chain, before it starts) cannot
// deadlock -- and the chains run beside the strips instead of in a kernel of their own before them (16 ms of a 166 ms
// sweep at 1/8 of the bench batch).
template <int KIND>
__device__ __forceinline__ void v2_chain_tile(const DevProb &Pmem, lchar *lds, const int which, int *prog, const int pgen, const int pro_off)
{
    if (threadIdx.x >= 64) return;
    DevProb P;
    uni_prob(P, Pmem);
    V2Geom G;
    G.capa = P.capa; G.capb = (KIND == 2) ? P.capb : 0;
    G.recsz = (16 + 4 * (G.capa + G.capb) + 15) & ~15; G.ndw = G.recsz / 4;
    G.nslot = 1; G.R = 4; G.lds = lds;
    unsigned *rowH2 = (unsigned *) P.v2_rowH + 2 * (size_t) P.v2_rowstride * G.ndw;
    const int penc = (pgen & 0x7FF) << 20;
    const int mlx = P.a.maxlist > P.b.maxlist ? P.a.maxlist : P.b.maxlist;
    if (pro_off && 3 * mlx <= PRO_CONST && 2 * mlx <= PRO_POOL) {
        ProLds S; S.carve(lds + pro_off);
        if (which == -1) v2_chain_top_staged<KIND>(P, G, G.row(0, 0), G.row(1, 0), rowH2, S, threadIdx.x, prog, penc);
        else v2_chain_left_staged<KIND>(P, G, G.row(0, 0), G.row(1, 0), (unsigned *) P.v2_colH, S, threadIdx.x, prog, penc);
    } else if (threadIdx.x == 0) {
        if (which == -1) v2_chain_top<KIND>(P, G, G.row(0, 0), G.row(1, 0), rowH2, prog, penc);
        else v2_chain_left<KIND>(P, G, G.row(0, 0), G.row(1, 0), (unsigned *) P.v2_colH, prog, penc);
    }
    if (threadIdx.x == 0) chain_publish(prog, penc, 0xFFFFF);
}

// row offsets of the column-score matrix (needed by the score kernel, which then runs beside the chains)
#ifdef G2G_TU_V2
extern "C" __global__ void __launch_bounds__(64)
g2g_v2_rowoff_kernel(const DevProb *probs, const int *idx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevProb &P = probs[idx[i]];
    long long o = 0;
    for (int m = P.a.left; m < P.a.right; ++m) {
        int nlo = m + P.lw; if (nlo < P.b.left) nlo = P.b.left;
        int nhi = m + P.up + 1; if (nhi > P.b.right) nhi = P.b.right;
        P.v2_rowoff[m - P.a.left] = o;
        if (nhi > nlo) o += nhi - nlo;
    }
    P.v2_rowoff[P.a.right - P.a.left] = o;
}
#else
extern "C" __global__ void g2g_v2_rowoff_kernel(const DevProb *probs, const int *idx, int n);
#endif

#ifdef G2G_TU_V2
extern "C" __global__ void __launch_bounds__(128)
g2g_v2_prologue_kernel(const DevProb *probs, const int *idx, int pro_off)
{   // pro_off: byte offset of the two list-staging areas in dynamic LDS (0: lists straight from HBM)
    extern __shared__ __attribute__((aligned(16))) char g2g_lds[];
    lchar *lds = (lchar *) g2g_lds;
    const DevProb &P = probs[idx[blockIdx.x]];
    if (P.kind == 1) v2_prologue<1>(P, lds, pro_off); else if (P.kind == 2) v2_prologue<2>(P, lds, pro_off);
}
#else
extern "C" __global__ void g2g_v2_prologue_kernel(const DevProb *probs, const int *idx, int pro_off);
#endif

__host__ __device__ __forceinline__ bool sim_tiled_kind(int k) { return k == 31 || k == 320 || k == 321 || k == 33 || k == 330; }
// PwdM::sim2 for every in-band cell of the gap-profile DPs (maln.h:160-172, maln2.cc:534-623,1230-1296):
// independent of the recurrence, so it is computed ahead of it, fully parallel, row-major per DP
#ifdef G2G_TU_V2
extern "C" __global__ void __launch_bounds__(256)
g2g_v2_sim_kernel(const DevProb *probs, const int *idx, int tiled)
{
    const DevProb &P = probs[idx[blockIdx.y]];
    if (!P.v2_sim) return;                                   // (strips in sweep mode make their own scores: SimBlk, g2g_kernels_v3.hip)
    if (tiled && sim_tiled_kind(P.sim2_kind)) return;        // done by g2g_v2_sim_tile_kernel
    const int m = P.a.left + blockIdx.x;
    if (m >= P.a.right) return;
    int nlo = m + P.lw; if (nlo < P.b.left) nlo = P.b.left;
    int nhi = m + P.up + 1; if (nhi > P.b.right) nhi = P.b.right;
    double *out = P.v2_sim + P.v2_rowoff[m - P.a.left] - nlo;
    for (int n = nlo + threadIdx.x; n < nhi; n += blockDim.x) out[n] = sim2(P, m, n);
}
#else
extern "C" __global__ void g2g_v2_sim_kernel(const DevProb *probs, const int *idx, int tiled);
#endif

// The same for the scorers of the gap-profile engines -- sim31, sim32i/w (profile of a x residues of b) and sim33/33_n
// (profile of a . frequency vector of b) -- TILED: a block owns 32 rows x 128 columns, stages the 32 profile rows of a
// and the 128 columns of b (residues or frequency vectors) in LDS once and computes 4096 scores from there.  The
// row-by-row kernel above re-reads b's column data for every row: 1.6e9 cells x 184 B through L2 for sim33, which is
// what made it take 34 ms per sweep.  Expressions and summation order are those of sim2() (g2g_kernels.hip).
#define SIM_TR 64
#define SIM_TC 128
// sim33 / sim33_n rows of one thread's column: b's frequency vector sits in registers (NB: compile-time bound of felm),
// a's profile row is a wave-uniform LDS read.  Same products, same summation order as sim2().
template <int NB, bool DC>
__device__ __forceinline__ void sim_tile_vec(const DevProb &P, const lf64 *Ap, const int nap, const lf64 *vbl, const int felm,
                                             const int m0, const int m1, const int n, const int r0)
{
    const DevSide &a = P.a, &b = P.b;
    double vb[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) vb[j] = j < felm ? vbl[j] : 0;
    for (int r = r0; r < m1 - m0; r += 2) {
        const int m = m0 + r;
        int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;
        int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
        const lf64 *va = Ap + r * nap;
        double sc = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            constexpr int dc[6] = {0, 1, 2, 3, 5, 9};
            if (j < felm) sc += va[DC ? dc[j < 6 ? j : 0] : j] * vb[j];
        }
        if (n >= nlo && n < nhi) P.v2_sim[P.v2_rowoff[m - a.left] + (n - nlo)] = sc;
    }
}
#ifdef G2G_TU_V2
extern "C" __global__ void __launch_bounds__(256)
g2g_v2_sim_tile_kernel(const DevProb *probs, const int *idx)
{
    extern __shared__ __attribute__((aligned(16))) char g2g_lds[];
    const DevProb &P = probs[idx[blockIdx.z]];
    const int kind = P.sim2_kind;
    if (!P.v2_sim || !sim_tiled_kind(kind)) return;         // (no matrix wanted / those run in g2g_v2_sim_kernel)
    const DevSide &a = P.a, &b = P.b;
    const int m0 = a.left + blockIdx.y * SIM_TR, c0 = b.left + blockIdx.x * SIM_TC;
    if (m0 >= a.right || c0 >= b.right) return;
    const int m1 = m0 + SIM_TR < a.right ? m0 + SIM_TR : a.right, c1 = c0 + SIM_TC < b.right ? c0 + SIM_TC : b.right;
    if (c0 > (m1 - 1) + P.up || c1 - 1 < m0 + P.lw) return;  // tile outside the band
    const int tid = threadIdx.x;
    const int na = a.nelm - a.felm;                         // profile part of a's column vectors
    const bool vecb = kind == 33 || kind == 330;
    const int nb = vecb ? b.felm : 0;
    const int nap = (na + 1) & ~1;                          // row pitch of Ap (even: 16-byte aligned rows)
    lf64 *Ap = (lf64 *) (lchar *) g2g_lds;                  // [SIM_TR][nap]
    lf64 *Bf = Ap + SIM_TR * nap;                           // [SIM_TC][nb]
    LDS uint8_t *Br = (LDS uint8_t *) Bf;                   // [SIM_TC][b.many] (the kinds that read residues have no Bf)
    for (int k = tid; k < (m1 - m0) * na; k += 256) { const int r = k / na, j = k - r * na; Ap[r * nap + j] = vss_at(a, m0 + r)[a.felm + j]; }
    if (vecb) for (int k = tid; k < (c1 - c0) * nb; k += 256) { const int c = k / nb, j = k - c * nb; Bf[k] = vss_at(b, c0 + c)[j]; }
    else for (int k = tid; k < (c1 - c0) * b.many; k += 256) Br[k] = res_at(b, c0)[k];
    __syncthreads();
    const int c = tid & (SIM_TC - 1), n = c0 + c;
    if (n >= c1) return;
    if (kind == 330 && b.felm <= 6) { sim_tile_vec<6, true>(P, Ap, nap, Bf + (size_t) c * nb, b.felm, m0, m1, n, tid >> 7); return; }
    if (kind == 33 && b.felm <= 6) { sim_tile_vec<6, false>(P, Ap, nap, Bf + (size_t) c * nb, b.felm, m0, m1, n, tid >> 7); return; }
    if (kind == 33 && b.felm <= 24) { sim_tile_vec<24, false>(P, Ap, nap, Bf + (size_t) c * nb, b.felm, m0, m1, n, tid >> 7); return; }
    for (int r = tid >> 7; r < m1 - m0; r += 2) {
        const int m = m0 + r;
        int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;
        int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
        if (n < nlo || n >= nhi) continue;
        const lf64 *va = Ap + r * nap;
        double sc = 0;
        if (kind == 31) sc = va[Br[(size_t) c * b.many]];
        else if (kind == 320) { const LDS uint8_t *br = Br + (size_t) c * b.many; for (int j = 0; j < b.many; ++j) sc += va[br[j]]; }
        else if (kind == 321) { const LDS uint8_t *br = Br + (size_t) c * b.many; for (int j = 0; j < b.many; ++j) sc += va[br[j]] * b.weight[j]; }
        else if (kind == 33) { const lf64 *vb = Bf + (size_t) c * nb; for (int j = 0; j < b.felm; ++j) sc += va[j] * vb[j]; }
        else { const lf64 *vb = Bf + (size_t) c * nb; const int dc[6] = {0, 1, 2, 3, 5, 9}; for (int j = 0; j < b.felm; ++j) sc += va[dc[j]] * vb[j]; }
        P.v2_sim[P.v2_rowoff[m - a.left] + (n - nlo)] = sc;
    }
}
#else
extern "C" __global__ void g2g_v2_sim_tile_kernel(const DevProb *probs, const int *idx);
#endif

// self / dep_*: indices into the batch's tile-completion flags (-1: no such neighbour)
struct V2Tile { int prob, ti, tj, nsteps, self, dep_up, dep_left, dep_diag, dep_war; };

// One PERSISTENT kernel per (record type, Noll) -- the register budget of a combined kernel would be its
// worst variant's.  Workgroups pull tiles from a queue ordered by tile wavefront (i + j) and wait on
// completion flags of the three tiles they depend on, so a tile starts as soon as ITS neighbours are done
// instead of when a whole wavefront launch has drained.  Dependencies always sit earlier in the queue,
// i.e. are finished or being worked on by a resident workgroup: waiting cannot deadlock for any grid size.
// Cross-workgroup visibility follows the agent-scope release/acquire recipe of the CDNA guide (G16).
// Every spin is bounded (a kernel that never ends can take the whole node down): on a time-out the wait
// gives up, the incident is counted in dbg[0] and the host reports the batch as failed.
__device__ __forceinline__ void v2_wait_flag(const int *flag, int gen, int *dbg, int tile, int *failp)
{
    (void) g2g_wait_ge(flag, gen, dbg, failp, tile);
}
#ifdef G2G_V2_STAMP
__device__ unsigned long long g2g_wait_acc[4];
#define V2_WAIT_T0 const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#define V2_WAIT_T1 const unsigned long long w1_ = __builtin_amdgcn_s_memtime();
#define V2_WAIT_T2 if (threadIdx.x == 0) { const unsigned long long w2_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g2g_wait_acc[0], w1_ - w0_); atomicAdd(&g2g_wait_acc[1], w2_ - w1_); }
#else
#define V2_WAIT_T0
#define V2_WAIT_T1
#define V2_WAIT_T2
#endif
// NOTE on control flow: nothing in this loop is done by "thread 0 only".  With one-lane branches next to
// the barriers the structurizer rotates the loop so that lane 0 leaves it to run its blocks while the
// other lanes of its wave go round again -- they then re-read a stale tile index for ever.  So the queue
// pop, the polling, the fences and the flag store are all executed uniformly (redundantly) by every lane:
// each thread adds (tid == 0) to the queue head, parks its result in LDS, and slot 0 is the tile.
#define V2_KERNEL(NAME, KIND, N3)                                                                   \
extern "C" __global__ void __launch_bounds__(G2G_V2_THREADS, G2G_V2_MINWAVES)                         \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int lds_tile_off, int C, int sweep, int pro_off, double *simscr) \
{                                                                                                   \
    extern __shared__ __attribute__((aligned(16))) char g2g_lds[];                                  \
    li32 *s_vals = (li32 *) ((lchar *) g2g_lds + lds_tile_off);   /* tail of the dynamic LDS */      \
    for (;;) {                                                                                      \
        s_vals[threadIdx.x] = atomicAdd(qhead, threadIdx.x == 0 ? 1 : 0);                           \
        __syncthreads();                                                                            \
        const int t = __builtin_amdgcn_readfirstlane(s_vals[0]);                                    \
        __syncthreads();                                                                            \
        if (t >= ntiles) break;                                                                     \
        const V2Tile T = tiles[t];                                                                  \
        if (T.ti < 0) {           /* a boundary chain (see v2_chain_tile) */                        \
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
        /* one call site of the tile function (see g2g_kernels_v3.hip) */                           \
        const int *pu = (sweep && T.dep_up >= 0) ? done + T.dep_up : (const int *) 0;                \
        const int *pl = (sweep && T.dep_left >= 0) ? done + T.dep_left : (const int *) 0;            \
        int *ps = sweep ? done + T.self : (int *) 0;                                                \
        V2_WAIT_T0                                                                                  \
        if (!sweep) {                                                                               \
            if (T.dep_up >= 0) v2_wait_flag(done + T.dep_up, gen, done + G2G_HDR, t, failp);             \
            if (T.dep_left >= 0) v2_wait_flag(done + T.dep_left, gen, done + G2G_HDR, t, failp);         \
            if (T.dep_diag >= 0) v2_wait_flag(done + T.dep_diag, gen, done + G2G_HDR, t, failp);         \
            if (T.dep_war >= 0) v2_wait_flag(done + T.dep_war, gen, done + G2G_HDR, t, failp);           \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
        }                                                                                           \
        __syncthreads();                                                                            \
        V2_WAIT_T1                                                                                  \
        v2_tile<KIND, N3>(probs[T.prob], (lchar *) g2g_lds, T.ti, sweep ? 0 : T.tj, T.nsteps, C, pu, ps, done + G2G_HDR, gen, sweep, pl, \
                          (sweep && simscr) ? simscr + (size_t) blockIdx.x * G2G_SIMBLK_STRIDE : (double *) 0, failp); \
        V2_WAIT_T2                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
        if (!sweep) {                                                                               \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
            G2G_POST(done + T.self, gen);     \
        }                                                                                           \
    }                                                                                               \
}
#ifdef G2G_TU_V2
V2_KERNEL(g2g_v2_hf2, 1, false)
V2_KERNEL(g2g_v2_hf3, 1, true)
V2_KERNEL(g2g_v2_pf2, 2, false)
V2_KERNEL(g2g_v2_pf3, 2, true)
#else
#define V2_KERNEL_DECL(NAME) extern "C" __global__ void NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int lds_tile_off, int C, int sweep, int pro_off, double *simscr);
V2_KERNEL_DECL(g2g_v2_hf2) V2_KERNEL_DECL(g2g_v2_hf3) V2_KERNEL_DECL(g2g_v2_pf2) V2_KERNEL_DECL(g2g_v2_pf3)
#endif
