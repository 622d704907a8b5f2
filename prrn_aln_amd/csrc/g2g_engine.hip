// g2g_engine.hip -- host side of level 0 (include/g2g.h): packs g2g_problem batches into one HBM arena,
// launches the forward + backtrack kernels (g2g_kernels.hip) on the context's stream and fetches results.
// No CPU fallback lives here: without a usable HIP device every entry point fails with G2G_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <map>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include <memory>
#include <algorithm>
#include <chrono>
#ifndef G2G_TU_ALL
#define G2G_TU_V1 1                 // this unit emits the v1 / traceback / calcSpScore kernels and the f3 / pairsum kernels; the strip
#endif                              // kernels come from the g2g_tu_*.hip units (g2g_device.h)
#include "../../include/g2g.h"
#include "g2g_device.h"
#include "g2g_internal.h"

#include "g2g_kernels.hip"          // one translation unit: kernels + launcher (no -fgpu-rdc needed)
#include "g2g_kernels_v2.hip"
#include "g2g_kernels_v3.hip"
#include "g2g_kernels_v6.hip"
#include "g2g_kernels_v7.hip"
#include "g2g_kernels_v8.hip"

static thread_local std::string g_err;
// process-wide totals over every context (g2g_process_counters): runs, waits that gave up, DPs re-run, of those on v1, gaps seen
// by waiting waves, and the part of [1] / [2] that belongs to batches prepared with the INJECT_STALL test hook
static std::atomic<long long> g_tot[8];
static std::mutex g_report_mx;
static std::string g_last_report;
void g2g_set_error(const char *fmt, const char *a)
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a);
    g_err = buf;
}
extern "C" const char *g2g_last_error(void) { return g_err.c_str(); }
extern "C" int g2g_abi_version(void) { return G2G_ABI_VERSION; }

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    g2g_set_error("HIP error: %s", hipGetErrorString(e_)); return G2G_ERR_DEVICE; } } while (0)

#define G2G_NVS 8                   // (HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: bench.py asks for 8)
struct g2g_ctx {
    int device;
    int ok;
    hipStream_t stream;
    hipStream_t vstream[G2G_NVS];   // the persistent launches of a run are independent of each other: each takes the next of these streams
                                    // (round 2 gave some pairs of launches one stream by table, and a sweep was the SUM of their times)
    hipEvent_t ev[4];
    hipEvent_t vev[G2G_NVS + 1];    // join events of those streams; [G2G_NVS]: the fork event
    char *stage; size_t stage_cap;  // pinned host staging buffer of g2g_batch_prepare, kept between calls
    void *sp_slots; size_t sp_slots_cap;   // g2g_batch_spscore: the slots of the streamed walks (kept between calls, grows)
    long long n_gaps; double max_gap_ms;   // waiters that found themselves off the machine for more than 4 ms between two looks at the clock (g2g_wait_ge), longest such gap
    double rt_ticks_per_ms;         // rate of s_memrealtime on this device, measured at g2g_create (the waits' time limit is wall clock)
    std::map<std::string, std::pair<bool, std::string>> opt;   // g2g_set_option: name -> (present, value); see g2g_opt
    // Device memory belongs to the CONTEXT: every buffer a batch needs (arena, tile table, flags, the strips' column-score
    // scratch, list twins, the walks' workspace) is a block of this pool, handed back when the batch is freed and reused by the
    // next batch of similar size.  No hipMalloc / hipFree happens between the first and the last persistent launch of a run
    // (g2g_batch_run sizes and takes everything in a dry pass before it launches anything), and a refinement window -- one
    // batch per window -- allocates nothing once the pool holds its sizes.  Contents are as undefined as a fresh allocation's.
    struct DevBlock { char *p; size_t cap; };
    std::vector<DevBlock> pool;     // free blocks
    hipStream_t sp_stream; hipEvent_t sp_ev; void *sp_slots2; size_t sp_slots2_cap;    // the walks that run beside the DPs (g2g_batch_spscore_begin)
    long long n_dev_malloc, n_dev_free, n_pool_hits;   // hipMalloc / hipFree calls made for the pool, requests served from it
    int ncu;                        // compute units of the device
    std::shared_ptr<int> alive;     // 1 while the context exists: device slabs that outlive it (twins held by g2g_group objects) free themselves
    long long n_runs, n_timeouts, n_recovered, n_v1;   // g2g_ctx_counters: batch runs, waits that ran into the limit, DPs re-run, of those on v1
    struct MStream { int lo, n; hipStream_t s; unsigned long long used; };
    std::vector<MStream> mstream;   // streams confined to a share of the CUs (units lo .. lo + n - 1 of 32; see cu_share_stream)
    unsigned long long mstamp;
    long long n_mstreams;           // CU-mask streams created so far
    std::string last_timeout;       // report of the last recovered time-out (g2g_ctx_last_timeout)
};

// Tuning and diagnostic switches belong to a context (g2g_set_option / g2g_get_option); a name a context has not set
// reads the process environment variable G2G_<NAME>, so the environment only supplies defaults.  A switch is "on" when it
// has a value (any, also the empty string) and "off" when g2g_set_option was given NULL for it.
static const char *g2g_opt(const g2g_ctx *c, const char *name)
{
    if (c) {
        auto it = c->opt.find(name);
        if (it != c->opt.end()) return it->second.first ? it->second.second.c_str() : 0;
    }
    char env[96];
    snprintf(env, sizeof env, "G2G_%s", name);
    return getenv(env);
}
extern "C" int g2g_set_option(g2g_ctx *c, const char *name, const char *value)
{
    if (!c || !name || !*name || strlen(name) > 64) { g2g_set_error("g2g_set_option: %s", "bad argument"); return G2G_ERR_ARG; }
    if (!strncmp(name, "G2G_", 4)) name += 4;
    c->opt[name] = std::make_pair(value != 0, std::string(value ? value : ""));
    return G2G_OK;
}
extern "C" void g2g_reset_options(g2g_ctx *c) { if (c) c->opt.clear(); }
extern "C" const char *g2g_get_option(const g2g_ctx *c, const char *name)
{
    if (!name) return 0;
    if (!strncmp(name, "G2G_", 4)) name += 4;
    return g2g_opt(c, name);
}

extern "C" __global__ void g2g_clock_kernel(unsigned long long *out) { *out = __builtin_amdgcn_s_memrealtime(); }

// The first HIP call of a process initialises the runtime, and that initialisation draws from / reseeds glibc's random()
// state (measured: srand(1); rand(); <HIP init>; rand() no longer gives 846930886).  The reference seeds its own generator
// from rand() (McRand, src/randiv.cc:41-50) and breaks ties with rand() elsewhere, so a drop-in library must leave that
// state alone: HIP initialises against a private state array, the caller's is put back.  (Later HIP calls do not touch it.)
struct RandStateGuard {
    char buf[256]; char *prev;
    RandStateGuard() { prev = initstate(1u, buf, sizeof buf); }
    ~RandStateGuard() { if (prev) setstate(prev); }
};

extern "C" g2g_ctx *g2g_create(int device)
{
    RandStateGuard keep_host_rand;
    // one hardware queue per concurrent persistent launch: HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES queues (4 by
    // default) and reads the variable when it initialises -- which this call does if nothing in the process did before
    // (setenv is not safe against a concurrent getenv in another thread: a multithreaded host either sets GPU_MAX_HW_QUEUES
    //  itself before it starts its threads -- the Python loader does -- or calls g2g_create before it starts them; the variable
    //  is only written here when it is unset)
    if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g2g_set_error("%s", "no HIP device visible");
        return NULL;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) { g2g_set_error("%s", "device index out of range"); return NULL; }
    if (hipSetDevice(device) != hipSuccess) { g2g_set_error("%s", "hipSetDevice failed"); return NULL; }
    g2g_ctx *c = new g2g_ctx();
    c->device = device;
    c->ok = 0;
    c->stage = 0; c->stage_cap = 0;
    c->n_dev_malloc = c->n_dev_free = c->n_pool_hits = 0; c->ncu = 256;
    c->alive = std::make_shared<int>(1);
    c->n_runs = c->n_timeouts = c->n_recovered = c->n_v1 = 0;
    c->n_gaps = 0; c->max_gap_ms = 0; c->n_mstreams = 0;
    c->sp_slots = 0; c->sp_slots_cap = 0; c->sp_stream = 0; c->sp_ev = 0; c->sp_slots2 = 0; c->sp_slots2_cap = 0;
    c->mstamp = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; g2g_set_error("%s", "stream"); return NULL; }
    for (int i = 0; i < 4; ++i) hipEventCreate(&c->ev[i]);
    for (int i = 0; i < G2G_NVS; ++i) hipStreamCreateWithFlags(&c->vstream[i], hipStreamNonBlocking);
    for (int i = 0; i < G2G_NVS + 1; ++i) hipEventCreateWithFlags(&c->vev[i], hipEventDisableTiming);
    // the code object must contain an image for this GPU (the library is built for gfx950 only)
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, (const void *) g2g_forward_kernel);
    if (e != hipSuccess) {
        g2g_set_error("no gfx950 kernel image usable on this device: %s", hipGetErrorString(e));
        (void) hipGetLastError();
    } else c->ok = 1;
    if (c->ok) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) c->ncu = pr.multiProcessorCount;
        // every strip kernel may use the whole LDS of a CU: said once, here, so that no run changes a function attribute
        // between two persistent launches
        const void *fns[] = {(const void *) g2g_v2_hf2, (const void *) g2g_v2_hf3, (const void *) g2g_v2_pf2, (const void *) g2g_v2_pf3,
                             (const void *) g2g_v3_hf2, (const void *) g2g_v3_hf3, (const void *) g2g_v3r_hf2, (const void *) g2g_v3r_hf3,
                             (const void *) g2g_v6_pf2, (const void *) g2g_v6_pf3};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (160 * 1024)) != hipSuccess) (void) hipGetLastError();
    }
    c->rt_ticks_per_ms = 1.0e5;                              // nominal 100 MHz; measured below
    if (c->ok) {
        unsigned long long *d = 0, t[2] = {0, 0};
        if (hipMalloc((void **) &d, sizeof(unsigned long long)) == hipSuccess) {
            std::chrono::steady_clock::time_point h[2];
            for (int k = 0; k < 2; ++k) {
                hipLaunchKernelGGL(g2g_clock_kernel, dim3(1), dim3(1), 0, c->stream, d);
                hipStreamSynchronize(c->stream);
                h[k] = std::chrono::steady_clock::now();
                hipMemcpy(&t[k], d, sizeof t[k], hipMemcpyDeviceToHost);
                if (k == 0) std::this_thread::sleep_for(std::chrono::milliseconds(25));
            }
            const double ms = std::chrono::duration<double, std::milli>(h[1] - h[0]).count();
            if (t[1] > t[0] && ms > 1) c->rt_ticks_per_ms = (double) (t[1] - t[0]) / ms;
            hipFree(d);
        }
        (void) hipGetLastError();
    }
    return c;
}

extern "C" void g2g_destroy(g2g_ctx *c)
{
    if (!c) return;
    // G2G_LOG_FILE (plain environment): the phases of the tear-down with their times, appended to that file (a test runner captures
    // stderr; a tear-down that does not come back must still say where it stands)
    FILE *lf = getenv("G2G_LOG_FILE") ? fopen(getenv("G2G_LOG_FILE"), "a") : 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!lf) return;
        fprintf(lf, "[g2g destroy %p] %-22s %8.1f ms\n", (void *) c, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        fflush(lf);
    };
    hipSetDevice(c->device);
    lap("begin");
    // Order: everything idle first; then the MEMORY goes while every stream it was ever used on still exists (hipFree waits for the
    // device: round 4 saw tear-downs that never came back when some fifty pool blocks were freed behind destroyed CU-mask streams);
    // the streams and events last.
    hipDeviceSynchronize();
    lap("device idle");
    *c->alive = 0;
    for (auto &bk : c->pool) hipFree(bk.p);
    c->pool.clear();
    if (c->sp_slots) hipFree(c->sp_slots);
    if (c->sp_slots2) hipFree(c->sp_slots2);
    lap("device pool");
    if (c->stage) hipHostFree(c->stage);
    lap("pinned staging");
    for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev[i]);
    for (int i = 0; i < G2G_NVS + 1; ++i) hipEventDestroy(c->vev[i]);
    if (c->sp_ev) hipEventDestroy(c->sp_ev);
    lap("events");
    for (auto &m : c->mstream) hipStreamDestroy(m.s);
    lap("CU-share streams");
    for (int i = 0; i < G2G_NVS; ++i) hipStreamDestroy(c->vstream[i]);
    if (c->sp_stream) hipStreamDestroy(c->sp_stream);
    lap("launch streams");
    hipStreamDestroy(c->stream);
    lap("main stream");
    if (lf) fclose(lf);
    delete c;
}

extern "C" int g2g_device_ok(g2g_ctx *c) { return c && c->ok; }
extern "C" void g2g_free(void *p) { free(p); }

// ---- the context's device-memory pool ----------------------------------------------------------------------------------
// Requests of 64 MB and more (arenas, the strips' scratch) take the smallest free block that fits, whatever its size -- batch
// sizes of a refinement loop vary call by call --; smaller ones only blocks of at most four times the request (a tile table
// must not sit on an arena).  New blocks are rounded up (1/8 for large ones) so that the next batch, a little larger, still
// fits.  The pool is trimmed when it holds more than 48 free blocks or when the device runs out of memory.
static const size_t POOL_BIG = (size_t) 64 << 20;
static void pool_trim(g2g_ctx *c, size_t keep_blocks)
{
    while (c->pool.size() > keep_blocks) {                   // smallest first (the big ones are the expensive ones to get back)
        size_t k = 0;
        for (size_t i = 1; i < c->pool.size(); ++i) if (c->pool[i].cap < c->pool[k].cap) k = i;
        hipFree(c->pool[k].p); ++c->n_dev_free;
        c->pool.erase(c->pool.begin() + k);
    }
}
static void *pool_take(g2g_ctx *c, size_t bytes, size_t *cap)
{
    if (!bytes) bytes = 16;
    const bool cache = !g2g_opt(c, "NO_ARENA_CACHE");
    int best = -1;
    if (cache)
        for (size_t i = 0; i < c->pool.size(); ++i) {
            const size_t cp = c->pool[i].cap;
            if (cp < bytes) continue;
            if (bytes < POOL_BIG && cp > std::max(4 * bytes, (size_t) 1 << 16)) continue;
            if (best < 0 || cp < c->pool[best].cap) best = (int) i;
        }
    if (best >= 0) {
        void *r = c->pool[best].p;
        *cap = c->pool[best].cap;
        c->pool.erase(c->pool.begin() + best);
        ++c->n_pool_hits;
        return r;
    }
    size_t want = bytes >= POOL_BIG ? bytes + bytes / 8 : bytes >= 4096 ? bytes + bytes / 4 : 4096;
    want = (want + 255) & ~(size_t) 255;
    if (!cache) want = bytes;
    void *r = 0;
    if (want >= POOL_BIG && cache) {                         // a new large block: the free large ones were all too small -- give them back first
        for (size_t i = c->pool.size(); i-- > 0; )
            if (c->pool[i].cap >= POOL_BIG) { hipFree(c->pool[i].p); ++c->n_dev_free; c->pool.erase(c->pool.begin() + i); }
    }
    hipError_t e = hipMalloc(&r, want);
    if (e != hipSuccess && want > bytes) { (void) hipGetLastError(); want = bytes; e = hipMalloc(&r, want); }
    if (e != hipSuccess) { (void) hipGetLastError(); pool_trim(c, 0); e = hipMalloc(&r, want); }
    if (e != hipSuccess) { (void) hipGetLastError(); return 0; }
    ++c->n_dev_malloc;
    *cap = want;
    return r;
}
static void pool_give(g2g_ctx *c, void *p, size_t cap)
{
    if (!p) return;
    if (g2g_opt(c, "NO_ARENA_CACHE")) { hipFree(p); ++c->n_dev_free; return; }
    g2g_ctx::DevBlock bk; bk.p = (char *) p; bk.cap = cap;
    c->pool.push_back(bk);
    if (c->pool.size() > 48) pool_trim(c, 32);
}
extern "C" void g2g_ctx_mem_counters(const g2g_ctx *c, long long out[4])
{
    if (!out) return;
    size_t held = 0;
    if (c) for (auto &bk : c->pool) held += bk.cap;
    out[0] = c ? c->n_dev_malloc : 0; out[1] = c ? c->n_dev_free : 0; out[2] = c ? c->n_pool_hits : 0; out[3] = (long long) held;
}

// ---- batch -----------------------------------------------------------------------------------

struct Blob {                       // host image of the input part of the arena
    // (a raw realloc'ed buffer: a std::vector zero-fills and copies 1.2 GB of profiles again and again while it
    // grows -- that was 1.1 s of a 1.3 s prepare for the bench sweep)
    char *p; size_t sz, cap;
    struct Copy { size_t off; const void *src; size_t bytes; };
    std::vector<Copy> later;        // big copies are deferred and done by a few host threads at once (flush)
    g2g_ctx *owner;                 // the buffer is the context's pinned staging area: it outlives the Blob (no page
                                    // faults and a DMA-able source on every call after the first)
    int pack_threads;               // option PACK_THREADS (0: one per core, at most 16)
    explicit Blob(g2g_ctx *c) : p(c->stage), sz(0), cap(c->stage_cap), owner(c), pack_threads(0), oom(false) {}
    ~Blob() { owner->stage = p; owner->stage_cap = cap; }
    Blob(const Blob &) = delete;
    Blob &operator=(const Blob &) = delete;
    bool oom;
    void grow(size_t need)
    {
        if (need <= cap || oom) return;
        const size_t c = need + need / 8 + ((size_t) 1 << 20);
        char *q = 0;
        if (hipHostMalloc((void **) &q, c, hipHostMallocDefault) != hipSuccess || !q) { (void) hipGetLastError(); oom = true; return; }
        if (sz) memcpy(q, p, sz);
        if (p) hipHostFree(p);
        p = q; cap = c;
    }
    void extend(size_t newsize) { if (newsize <= sz) return; grow(newsize); if (oom) return; memset(p + sz, 0, newsize - sz); sz = newsize; }
    size_t put(const void *src, size_t bytes)
    {
        const size_t off = (sz + 15) & ~(size_t) 15;
        grow(off + bytes);
        if (oom) return 0;
        if (off > sz) memset(p + sz, 0, off - sz);
        if (bytes >= ((size_t) 64 << 10)) { Copy c = {off, src, bytes}; later.push_back(c); }
        else if (bytes) memcpy(p + off, src, bytes);
        sz = off + bytes;
        return off;
    }
    void flush()
    {
        if (later.empty() || oom) return;
        unsigned nthr = std::thread::hardware_concurrency();
        if (nthr > 16) nthr = 16;
        if (nthr < 1) nthr = 1;
        if (pack_threads >= 1 && pack_threads <= 64) nthr = (unsigned) pack_threads;
        if (nthr > later.size()) nthr = (unsigned) later.size();
        std::atomic<size_t> next(0);
        auto work = [&]() { for (size_t k; (k = next.fetch_add(1)) < later.size(); ) memcpy(p + later[k].off, later[k].src, later[k].bytes); };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthr; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        later.clear();
    }
    char *data() { return p; }
    size_t size() const { return sz; }
};

// the problem index of a batch sits in a grid y / z dimension of some launches (limit 65535)
#define G2G_MAX_BATCH 32768
extern "C" void g2g_batch_free(g2g_batch *b);
struct g2g_batch {
    g2g_ctx *ctx;
    int n;
    std::vector<DevProb> dp;        // host copy with device pointers
    std::vector<int> status;
    std::vector<long long> cells;
    std::vector<size_t> out_off;    // per problem: offset of {score, ntrace, otrace} block
    std::vector<int> tcap;
    size_t out_lo, out_hi;          // the output blocks occupy arena [out_lo, out_hi)
    std::vector<long long> rr1;     // b.left - a.left + b.right - a.right per problem
    char *d_arena;
    size_t arena_bytes, in_bytes;
    size_t arena_cap;               // bytes actually allocated behind d_arena (>= arena_bytes when the arena was reused)
    DevProb *d_probs;
    int *d_idx1, *d_idx2;           // problems run by the v1 / v2 forward kernel
    int *d_idxp; int np;            // problems whose boundary chains run in the prologue kernel (the others: as queue entries)
    int n1, n2;
    int nsimmat;                    // DPs that read a column-score matrix (tile-mode test configurations): only then the matrix kernels run
    size_t lds2;                    // dynamic LDS bytes of the v2 launch
    size_t lds2p;                   // ... of the v2 prologue launch
    int v2_maxrows;                 // longest a-range among the v2 problems
    int v2_maxcols;                 // longest b-range ...
    size_t simtile_lds;             // LDS of the tiled column-score kernel
    size_t tiles_cap, flags_cap;    // pool blocks behind d_tiles / d_flags
    V2Tile *d_tiles;                // tiles: per variant (v2: hf2, hf3, pf2, pf3; v3: the same four) a queue ordered by wavefront i + j
    int var_off[25];                // variant v owns tiles [var_off[v], var_off[v+1])
    long long var_cells[24];        // in-band cells of the DPs of variant v (the CU shares of a run are proportional to cells x cost)
    V3Lds v3lds[8];                 // LDS plan of the v3 variants
    V6Lds v6lds[6];                 // LDS plans of the v6 (_pf, one lane per cell, rank-form merges) launches: Noll 2, 3 x footprint class A / B / C
    int v2_cols;
    int v2_threads;                 // workgroup size of the v2 kernels: 256 (32-row strips) or 128 (16-row strips)
    int v2_sweep;                   // v2 (_pf): the same
    int v3_sweep;                   // v3r (_hf): strips as a pipeline with progress counters (one tile per strip)
    int v3_cols;                    // columns per v3 tile
    int *d_flags;                   // [0, G2G_HDR) queue heads, [G2G_HDR, +4) header of the waits, then tile flags / progress counters, then per-DP fail flags
    int nflags, gen;
    std::vector<int> flags0;        // initial contents of d_flags (re-uploaded when the 11-bit generation of the progress counters wraps)
    long long ntiles;
    float fwd_ms, tb_ms;
    double *simscr[24]; size_t simscr_cap[24];
    unsigned *twin[6]; size_t twin_cap[6];       // per v6 launch: HBM image of the dynamic lists' parts beyond their inline LDS slots
    bool v6_on;                     // this batch is large enough for v6 (else its _pf DPs go to v2: shorter critical path)
    int hdr_img[G2G_HDR + G2G_HDRN];       // host image of the queue heads + wait header of the current run
    std::vector<const g2g_problem *> src;        // the caller's problems (kept alive by the caller until the batch is freed): a DP
                                                 // that lost a wait is re-run from here on the non-polling kernel
    int fail_off;                                // offset of the per-DP fail flags in d_flags
    int dump_off;                                // offset of the time-out dump area in d_flags (0: none)
    int n_recovered;                             // DPs re-run after a time-out, over the life of the batch
    int last_timeouts, last_recovered;           // the same for the last g2g_batch_run: waits that gave up, DPs re-run
    std::vector<g2g_result> recovered;           // results of re-run DPs (trace owned by the batch until fetched)
    std::vector<char> was_recovered;
    bool injected;                               // prepared with the INJECT_STALL test hook: its time-outs are expected ones
    bool is_retry;                               // a batch of DPs that lost a wait, re-run on the ordinary kernels; ITS time-outs go to v1
    bool force_v1;                               // recovery batches: everything on g2g_forward_kernel   // per sweep-mode launch: strip-local column-score blocks (3 x 32 KB per workgroup)
};

// LDS footprint of g2g_forward_kernel_v2 for one problem (see V2Geom): (slots * R + extras) records
static size_t v2_lds_bytes(int kind, int noll, int capa, int capb, int mla, int mlb, int threads = G2G_V2_THREADS)
{
    const size_t recsz = (16 + 4 * (size_t) (capa + (kind == 2 ? capb : 0)) + 15) & ~(size_t) 15;
    const size_t R = threads / 8;
    const size_t lists = (size_t) 10 * 3 * (R * mla + (kind == 2 ? (R + 2) * mlb : 0)) + 8;   // glen i16 + freq f64
    return (((noll == 3 ? 9 : 6) * R + 5) * recsz + 16 * R + lists + 16 + 15) & ~(size_t) 15;
}
static const size_t V2_LDS_MAX = 160 * 1024;

// LDS plan of the v3 kernel (g2g_kernels_v3.hip) for one problem with C-column tiles: ring rows, the black
// list, staging scalars, and the static-list pools of a strip's rows / a block's columns
static V3Lds v3_layout(int rows_bytes, int ca4max, int apool, int bpool, int C)
{
    V3Lds L;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o = (o + bytes + 15) & ~15; return r; };
    L.rows = take(rows_bytes);
    L.black = take(4 * (ca4max + 4));
    L.stsc = take(4 * 28);
    L.aglen = take(4 * (apool + 4));
    L.afreq = take(8 * (apool + 4));
    L.boff = take(bpool ? 4 * 3 * C : 0);
    L.bglen = take(bpool ? 4 * (bpool + 4) : 0);
    L.bfreq = take(bpool ? 8 * (bpool + 4) : 0);
    L.svals = take(4 * 64);
    L.sink = take(4 * 64);
    L.total = o;
    return L;
}
// LDS plan of the v6 kernel (g2g_kernels_v6.hip): ring rows of dynamic lists, black lists, staging scalars, the ring of
// b's static lists (3 views x rs entries x 16 B), queue scratch, sinks
struct V6Ring { int rs[3]; };                  // ring entries per view; entries of the t lists of the strip that has the most
static V6Lds v6_layout(int rows_bytes, int ca4max, const V6Ring &R)
{
    V6Lds L;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o = (o + bytes + 15) & ~15; return r; };
    L.rows = take(rows_bytes);
    L.black = take(4 * (ca4max + 8));
    L.stsc = take(4 * 28);
    for (int v = 0; v < 3; ++v) { L.rs[v] = R.rs[v]; L.ringf[v] = take(8 * R.rs[v]); }
    for (int v = 0; v < 3; ++v) L.ringk[v] = take(v < 2 ? 4 * R.rs[v] : 0);       // (view 2: an array of head freqs per column, no keys)
    L.svals = take(4 * 64);
    L.sink = take(4 * 64 + 16 * 64);
    L.total = o;
    return L;
}
// A launch has ONE LDS plan, the largest of its DPs, and LDS decides how many strips a CU holds (the kernel takes a whole SIMD's
// registers: four per CU at most).  Since the dynamic lists keep only their inline parts in LDS (g2g_kernels_v6.hip, LS6) a strip
// of the bench sweep takes 26-30 KB; class A (up to 40 KB, four strips per CU) holds all of those.  The balanced divisions whose
// column lists need a 1024-entry ring (40-53 KB: three strips per CU) are FASTER on v2 (8 lanes per cell, 16-row strips): the
// bench sweep 714 -> 660 ms with them there (same box; limit 46 KB 709, 49 KB 738, 36 KB 663, 30 KB 694), so the default limit
// of v6 is class A's.  Classes B (up to V6_SMALL_KB) and C (V6_LARGE_KB) stay as options for measurements.
static const int V6_CLASS_A = 40 * 1024;
static int v6_small_lds(const g2g_ctx *c) { const char *e = g2g_opt(c, "V6_SMALL_KB"); return e ? atoi(e) * 1024 : 40 * 1024; }
#define V6_SMALL_LDS (v6_small_lds(ctx))
static int v6_large_lds(const g2g_ctx *c) { const char *e = g2g_opt(c, "V6_LARGE_KB"); return e ? atoi(e) * 1024 : 0; }
#define V6_LARGE_LDS (v6_large_lds(ctx))
static inline int v6_slot(int cls) { return cls < 4 ? 12 + cls : 16 + cls; }      // queue / variant slot of class index (footprint class x 2 + Noll 3)
static int v6_rows_bytes(const DevProb &d)
{
    const int lsz = v6_inline_dw((d.capa + 3) & ~3) + v6_inline_dw((d.capb + 3) & ~3);      // the INLINE parts of a record's two lists (LS6)
    return 65 * v3_pitch(d.noll == 3 ? 9 : 6, lsz) * 4 + 32;
}
// ring entries the v6 kernel needs per view for this problem: the most list entries (terminators not counted) any window
// of V6_WINDOW consecutive columns of b holds (a ring is indexed by compact pool position & (rs - 1))
static V6Ring v6_ring_need(const g2g_problem *p)
{
    V6Ring R;
    const int lo = p->b.left, hi = p->b.right;
    for (int v = 0; v < 3; ++v) {
        const int32_t *off = p->b.gfq.off[v];
        int need = 1;
        for (int c = lo; c < hi; ++c) {
            const int e = std::min(c + V6_WINDOW, hi);
            need = std::max(need, (off[e + 1] - (e + 1)) - (off[c + 1] - (c + 1)));
        }
        int rs = 32;
        while (rs < need) rs <<= 1;
        R.rs[v] = v < 2 ? rs : V6_RHCOLS;                    // (the r view is derived from the t ring: only its head freqs, one per column)
    }
    return R;
}
struct V3Need { int rows_bytes, ca4, apool, bpool, total; };
static V3Need v3_need(const DevProb &d, const g2g_problem *p, int C, bool areg = false)
{
    V3Need n;
    const int capb = d.kind == 2 ? d.capb : 0;
    n.ca4 = (d.capa + 3) & ~3;
    const int lsz = n.ca4 + ((capb + 3) & ~3);
    n.rows_bytes = 65 * v3_pitch(d.noll == 3 ? 9 : 6, lsz) * 4;
    n.apool = 0; n.bpool = 0;
    const int al = p->a.left, ar = p->a.right, bl = p->b.left, br = p->b.right;
    for (int m0 = al; m0 < ar; m0 += 64) {
        const int me = std::min(m0 + 64, ar);
        int c = 0;
        for (int v = 0; v < 3; ++v) c += p->a.gfq.off[v][me + 1] - p->a.gfq.off[v][m0 + 1];
        n.apool = std::max(n.apool, c);
    }
    if (d.kind == 2)
        for (int c0 = bl; c0 < br; c0 += C) {
            const int c1 = std::min(c0 + C, br);
            int c = 0;
            for (int v = 0; v < 3; ++v) c += p->b.gfq.off[v][c1 + 1] - p->b.gfq.off[v][c0 + 1];
            n.bpool = std::max(n.bpool, c);
        }
    if (d.kind == 2 && !n.bpool) n.bpool = 1;
    if (areg) n.apool = 0;                                  // the rows' static lists live in registers
    n.total = v3_layout(n.rows_bytes, n.ca4, n.apool, n.bpool, C).total;
    return n;
}

static int kind_of(int alnmode)
{
    switch (alnmode) {
    case G2G_NGP_ALB: case G2G_NGP_ALN: return 0;
    case G2G_HLF_ALB: case G2G_RHF_ALB: return 1;
    case G2G_GPF_ALB: return 2;
    case G2G_NTV_ALB: return 3;
    }
    // The other rectangular modes (HLF / RHF / GPF / NTV_ALN) are NOT on this path, for a reason: Fwd2c::forwardA starts every row
    // with `*hdiag = *h` (src/fwd2c.h:247), a struct assignment that makes the diagonal record SHARE the left boundary record's
    // gap-state arrays; from then on a growing triangle of records is updated in place through one array, in the row-major order
    // of the reference's loop.  The restatement in oracle/ reproduces that (all 24 rectangular goldens, every record type), but the
    // result depends on a sequential order no anti-diagonal sweep has.  DPunit carries no arrays: NGP_ALN is exact and built.
    return -1;
}
static bool is_rect(int alnmode) { return alnmode == G2G_NGP_ALN; }

static int check_problem(const g2g_problem *p)
{
    if (!p) return G2G_ERR_ARG;
    int kind = kind_of(p->alnmode);
    if (kind < 0) return G2G_ERR_MODE;           // rectangular (_ALN) and spliced engines: not on this path yet
    if (p->noll != 2 && p->noll != 3) return G2G_ERR_ARG;
    const g2g_side *s[2] = {&p->a, &p->b};
    for (int k = 0; k < 2; ++k) {
        if (s[k]->many < 1 || s[k]->len < 1 || !s[k]->seq || !s[k]->thk) return G2G_ERR_ARG;
        if (s[k]->left < 0 || s[k]->right > s[k]->len || s[k]->left >= s[k]->right) return G2G_ERR_ARG;
    }
    if (is_rect(p->alnmode)) {                       // the caller's band is not read: the DP covers the rectangle
        if (p->spb_fact != 0 && p->a.npfq > 0 && p->b.npfq > 0) return G2G_ERR_MODE;      // (the intron bonus table is forwardB's)
    } else {
    if (p->up < p->lw) return G2G_ERR_ARG;
    // the end corner must lie inside the band (stripe() guarantees it, aln2.cc:156-174)
    const int re_ = p->b.right - p->a.right, rs_ = p->b.left - p->a.left;
    if (re_ < p->lw || re_ > p->up || rs_ < p->lw || rs_ > p->up) return G2G_ERR_ARG;
    }
    if (kind == 1 && !p->a.has_gfq) return G2G_ERR_ARG;
    if (kind == 2 && (!p->a.has_gfq || !p->b.has_gfq)) return G2G_ERR_ARG;
    if (kind == 3 && (!p->a.gapdens || !p->b.gapdens || !p->a.postgapdens || !p->b.postgapdens || !p->crg2_kind)) return G2G_ERR_ARG;
    switch (p->sim2_kind) {
    case G2G_SIM00: case G2G_SIM11: case G2G_SIM12I: case G2G_SIM21I: case G2G_SIM22I: break;
    case G2G_SIM12W: if (!p->b.weight) return G2G_ERR_ARG; break;
    case G2G_SIM21W: if (!p->a.weight) return G2G_ERR_ARG; break;
    case G2G_SIM22W: if (!p->a.weight || !p->b.weight) return G2G_ERR_ARG; break;
    case G2G_SIM13: case G2G_SIM23I: if (!p->b.pseq) return G2G_ERR_ARG; break;
    case G2G_SIM23W: if (!p->b.pseq || !p->a.weight) return G2G_ERR_ARG; break;
    case G2G_SIM31: case G2G_SIM32I: if (!p->a.pseq) return G2G_ERR_ARG; break;
    case G2G_SIM32W: if (!p->a.pseq || !p->b.weight) return G2G_ERR_ARG; break;
    case G2G_SIM33: case G2G_SIM33N: if (!p->a.pseq || !p->b.pseq) return G2G_ERR_ARG; break;
    default: return G2G_ERR_MODE;
    }
    if (!p->simmtx && (p->sim2_kind == G2G_SIM11 || p->sim2_kind / 10 == 12 || p->sim2_kind / 10 == 21 || p->sim2_kind / 10 == 22))
        return G2G_ERR_ARG;
    return G2G_OK;
}

// offsets (relative to arena base) are stored in the pointer fields first, rebased after allocation
template <class T> static inline T *OFF(size_t off) { return (T *) (uintptr_t) (off + 1); }   // +1: 0 stays NULL
template <class T> static inline void rebase(T *&p, char *base) { if (p) p = (T *) (base + ((uintptr_t) p - 1)); }

// An array that already lives in this context's HBM (g2g_side::dev, left there by g2g_device_derive) is not packed: the
// descriptor field gets the twin's address once the arena's offsets have been rebased (Patch list).
struct DevPatch { const void **field; const void *value; };
template <class T> static inline bool use_twin(std::vector<DevPatch> &patches, const T *&field, const T *twin)
{
    if (!twin) return false;
    field = 0;
    DevPatch p = {(const void **) &field, (const void *) twin};
    patches.push_back(p);
    return true;
}
static void pack_side(Blob &bl, const g2g_side &s, DevSide &d, int kind, bool need_gfq, std::vector<DevPatch> &patches)
{
    memset(&d, 0, sizeof d);
    const g2g_side_dev *tw = (s.dev && s.dev->ctx == bl.owner && !g2g_opt(bl.owner, "NO_RESIDENT_INPUTS")) ? s.dev : 0;
    d.many = s.many; d.len = s.len; d.left = s.left; d.right = s.right; d.nils = s.nils;
    d.nelm = s.nelm; d.felm = s.felm; d.sumwt = s.sumwt > 0 ? s.sumwt : (double) s.many;
    const size_t cols = (size_t) s.len + 2;
    if (!use_twin(patches, d.seq, tw ? tw->seq : 0)) d.seq = OFF<const uint8_t>(bl.put(s.seq, cols * s.many));
    if (s.weight && !use_twin(patches, d.weight, tw ? tw->weight : 0)) d.weight = OFF<const double>(bl.put(s.weight, sizeof(double) * s.many));
    if (s.pseq && s.nelm > 0 && !use_twin(patches, d.pseq, tw ? tw->pseq : 0)) d.pseq = OFF<const double>(bl.put(s.pseq, sizeof(double) * cols * s.nelm));
    if (!use_twin(patches, d.thk, tw ? tw->thk : 0)) d.thk = OFF<const double>(bl.put(s.thk, sizeof(double) * cols * 3));
    if (need_gfq) {
        d.hetero = s.gfq.hetero;
        {   // r = [head?] + (t with glen + 1): the structure Gfq::seq2gfq gives the r view (reference src/gfreq.cc:218-226)
            bool ok = true;
            const int32_t *to = s.gfq.off[1], *ro = s.gfq.off[2], *tg = s.gfq.glen[1], *rg = s.gfq.glen[2];
            const double *tf = s.gfq.freq[1], *rf = s.gfq.freq[2];
            for (int i = 1; i <= s.len && ok; ++i) {            // positions 0 .. len-1 (position -1 is the boundary's: only the chains read it)
                const int tl = to[i + 1] - to[i] - 1, rl = ro[i + 1] - ro[i] - 1;        // entries without the terminator
                if (tl < 0 || rl < 0) { ok = false; break; }
                const int h = (rl > 0 && rg[ro[i]] == 0) ? 1 : 0;
                if (rl != tl + h) { ok = false; break; }
                for (int k = 0; k < tl; ++k)
                    if (rg[ro[i] + h + k] != tg[to[i] + k] + 1 || rf[ro[i] + h + k] != tf[to[i] + k]) { ok = false; break; }
            }
            d.r_from_t = ok ? 1 : 0;
        }
        for (int v = 0; v < 3; ++v) {
            const int nlist = s.len + 2;                       // offsets for positions -1..len-1 + end
            const int pool = s.gfq.off[v][s.len + 1];
            for (int i = 0; i + 1 < nlist; ++i) d.maxlist = std::max(d.maxlist, s.gfq.off[v][i + 1] - s.gfq.off[v][i]);
            if (!use_twin(patches, d.off[v], tw ? (const int *) tw->off[v] : 0)) d.off[v] = OFF<const int>(bl.put(s.gfq.off[v], sizeof(int) * nlist));
            if (!use_twin(patches, d.glen[v], tw ? (const int *) tw->glen[v] : 0)) d.glen[v] = OFF<const int>(bl.put(s.gfq.glen[v], sizeof(int) * pool));
            if (!use_twin(patches, d.freq[v], tw ? tw->freq[v] : 0)) d.freq[v] = OFF<const double>(bl.put(s.gfq.freq[v], sizeof(double) * pool));
        }
    }
    if (kind == 3) {
        d.gapdens = OFF<const double>(bl.put(s.gapdens, sizeof(double) * cols * s.many));
        d.postgapdens = OFF<const double>(bl.put(s.postgapdens, sizeof(double) * cols * s.many));
    }
    if (s.npfq > 0 && s.pfq_pos && s.pfq_dns) {
        d.npfq = s.npfq; d.pfq_step = s.pfq_step;
        d.pfq_pos = OFF<const int>(bl.put(s.pfq_pos, sizeof(int) * (size_t) s.npfq));
        d.pfq_dns = OFF<const double>(bl.put(s.pfq_dns, sizeof(double) * (size_t) s.npfq));
    }
}

// The intron-position bonus of forwardB (reference src/fwd2c.h:367-370,378-379,446-452,472; PfqItr src/gsinfo.h:132-231) as a
// table of cells.  The reference walks two cursors: `api` over a's exon boundaries (it advances by one entry at the end of
// every row whose codon holds the current entry), and per such row `bpi` over b's, restarted at the row's first column and
// advanced by one entry at every column whose codon holds ITS current entry -- a cursor that has fallen behind the column
// (two boundaries in one codon) never catches up.  Both walks depend only on the inputs and the band, not on scores.
struct BonusCell { int m, n; double h, mx; };
static bool in_codon(int pos, int step, int col) { return step == 1 ? pos == col : ((long long) col * step <= pos && pos < (long long) col * step + step); }
static void intron_bonus_table(const g2g_problem *p, std::vector<BonusCell> &out)
{
    out.clear();
    const g2g_side &a = p->a, &b = p->b;
    if (p->spb_fact == 0 || a.npfq <= 0 || b.npfq <= 0 || !a.pfq_pos || !b.pfq_pos || !a.pfq_dns || !b.pfq_dns) return;
    const int sa = a.pfq_step, sb = b.pfq_step;
    int ka = 0;
    while (ka < a.npfq && a.pfq_pos[ka] < (long long) a.left * sa) ++ka;
    for (int m = a.left; m < a.right; ++m) {
        if (ka >= a.npfq) break;                                   // (the cursor only moves forward)
        if (!in_codon(a.pfq_pos[ka], sa, m)) continue;
        const int n0 = std::max(m + p->lw, b.left), n9 = std::min(m + p->up + 1, b.right);
        int kb = 0;
        while (kb < b.npfq && b.pfq_pos[kb] < (long long) n0 * sb) ++kb;
        for (int n = n0; n < n9 && kb < b.npfq; ++n) {
            if (!in_codon(b.pfq_pos[kb], sb, n)) continue;
            const int apos = a.pfq_pos[ka], bpos = b.pfq_pos[kb];
            BonusCell c;
            c.m = m; c.n = n;
            const bool phase = sa == 1 || (apos - bpos) % sa == 0;   // match_score: the two positions in the same codon phase
            c.h = phase ? p->spb_fact * a.pfq_dns[ka] * b.pfq_dns[kb] : 0;
            c.mx = (phase && (sa == 1 || apos % sa == 0)) ? p->spb_fact * a.pfq_dns[ka] * b.pfq_dns[kb] : 0;
            out.push_back(c);
            ++kb;
        }
        ++ka;
    }
}

static void rebase_side(DevSide &d, char *base)
{
    rebase(d.seq, base); rebase(d.weight, base); rebase(d.pseq, base); rebase(d.thk, base);
    for (int v = 0; v < 3; ++v) { rebase(d.off[v], base); rebase(d.glen[v], base); rebase(d.freq[v], base); }
    rebase(d.gapdens, base); rebase(d.postgapdens, base);
    rebase(d.pfq_pos, base); rebase(d.pfq_dns, base);
}

// a batch gives its arena back to the context's pool
static void release_arena(g2g_batch *b)
{
    if (!b->d_arena) return;
    pool_give(b->ctx, b->d_arena, b->arena_cap);
    b->d_arena = 0;
}

static int batch_prepare_impl(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_batch **out, bool force_v1);
extern "C" int g2g_batch_prepare(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_batch **out)
{
    return batch_prepare_impl(ctx, n, prob, out, false);
}
static int batch_prepare_impl(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_batch **out, bool force_v1)
{
    if (!ctx || n < 0 || !out) return G2G_ERR_ARG;
    if (n > G2G_MAX_BATCH) { g2g_set_error("%s", "g2g_batch_prepare: more than 32768 problems in one batch (use g2g_forward_batch, which cuts chunks)"); return G2G_ERR_ARG; }
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    const bool prep_dbg = g2g_opt(ctx, "DEBUG_PREP") != 0;
    auto prep_t0 = std::chrono::steady_clock::now();
    auto prep_lap = [&](const char *what) {
        if (!prep_dbg) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[g2g prepare] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - prep_t0).count());
        prep_t0 = t;
    };
    g2g_batch *b = new g2g_batch();
    b->ctx = ctx; b->n = n; b->d_arena = 0; b->d_probs = 0; b->fwd_ms = b->tb_ms = 0;
    for (int k = 0; k < 24; ++k) { b->simscr[k] = 0; b->simscr_cap[k] = 0; }
    for (int k = 0; k < 6; ++k) { b->twin[k] = 0; b->twin_cap[k] = 0; }
    b->src.assign(prob, prob + n); b->fail_off = 0; b->dump_off = 0; b->force_v1 = force_v1; b->is_retry = false; b->n_recovered = 0;
    b->recovered.assign(n, g2g_result()); b->was_recovered.assign(n, 0);
    b->nsimmat = 0; b->injected = g2g_opt(ctx, "INJECT_STALL") != 0;
    for (int k = 0; k < 24; ++k) b->var_cells[k] = 0;
    b->last_timeouts = b->last_recovered = 0;
    b->v3_cols = 128; b->v2_cols = G2G_V2_TILE_COLS;

    b->dp.resize(n); b->status.assign(n, G2G_OK); b->cells.assign(n, 0); b->out_off.assign(n, 0); b->tcap.assign(n, 0); b->rr1.assign(n, 0);
    Blob bl(ctx);
    if (const char *e = g2g_opt(ctx, "PACK_THREADS")) bl.pack_threads = atoi(e);
    {   // size the staging buffer once (growing a pinned buffer means allocating and copying it again)
        auto side_bytes = [&](const g2g_side &s) {
            const size_t cols = (size_t) s.len + 2;
            const g2g_side_dev *tw = (s.dev && s.dev->ctx == ctx) ? s.dev : 0;
            size_t t = 8 * (size_t) s.many + 24 * cols + 256;
            if (!(tw && tw->seq)) t += cols * s.many;
            if (s.pseq && s.nelm > 0 && !(tw && tw->pseq)) t += 8 * cols * s.nelm;
            if (s.has_gfq) for (int v = 0; v < 3; ++v) if (s.gfq.off[v] && !(tw && tw->freq[v])) t += 4 * cols + 12 * (size_t) s.gfq.off[v][s.len + 1] + 64;
            if (s.gapdens) t += 16 * cols * s.many;
            return t;
        };
        size_t est = (sizeof(DevProb) + 8) * (size_t) (n > 0 ? n : 1) + 4096;
        for (int i = 0; i < n; ++i) {
            const g2g_problem *p = prob[i];
            if (check_problem(p) != G2G_OK) continue;
            est += side_bytes(p->a) + side_bytes(p->b) + 8 * (size_t) (p->simmtx ? p->simdim * p->simrows : 0) + 64;
        }
        bl.grow(est);
    }
    size_t probs_off = bl.put(0, 0);
    bl.extend(probs_off + sizeof(DevProb) * (size_t) (n > 0 ? n : 1));
    std::vector<DevPatch> patches;            // descriptor fields that point at device-resident inputs (b->dp is never resized)
    // 1. inputs
    for (int i = 0; i < n; ++i) {
        DevProb &d = b->dp[i];
        memset(&d, 0, sizeof d);
        const g2g_problem *p = prob[i];
        int rc = check_problem(p);
        b->status[i] = rc;
        if (rc) { d.kind = -1; continue; }
        d.kind = kind_of(p->alnmode);
        d.noll = p->noll; d.sim2_kind = p->sim2_kind; d.crg2_kind = p->crg2_kind; d.codonk1 = p->codonk1;
        d.rect = is_rect(p->alnmode) ? 1 : 0;
        d.lw = p->lw; d.up = p->up;
        if (d.rect) { d.lw = p->b.left - p->a.right; d.up = p->b.right - p->a.left; }     // every cell and every boundary corner inside
        d.width = d.up - d.lw + 3;
        d.basic_gop = p->basic_gop; d.weighted_gop = p->weighted_gop; d.u = p->u;
        d.u2divu1 = p->u2divu1; d.v2divv1 = p->v2divv1;
        d.simdim = p->simdim;
        if (p->simmtx) d.simmtx = OFF<const double>(bl.put(p->simmtx, sizeof(double) * (size_t) p->simdim * p->simrows));
        pack_side(bl, p->a, d.a, d.kind, d.kind == 1 || d.kind == 2, patches);
        pack_side(bl, p->b, d.b, d.kind, d.kind == 2, patches);
        d.spb_fact = p->spb_fact; d.dvsp = p->dvsp;
        {
            std::vector<BonusCell> bc;
            intron_bonus_table(p, bc);
            if (!bc.empty()) {
                std::vector<int> bm, bn; std::vector<double> bh, bx;
                for (const BonusCell &c : bc) { bm.push_back(c.m); bn.push_back(c.n); bh.push_back(c.h); bx.push_back(c.mx); }
                d.nbonus = (int) bc.size();
                d.bon_m = OFF<const int>(bl.put(bm.data(), sizeof(int) * bm.size()));
                d.bon_n = OFF<const int>(bl.put(bn.data(), sizeof(int) * bn.size()));
                d.bon_h = OFF<const double>(bl.put(bh.data(), sizeof(double) * bh.size()));
                d.bon_mx = OFF<const double>(bl.put(bx.data(), sizeof(double) * bx.size()));
            }
        }
    }
    // Tile widths.  Tiles of one DP run as a wavefront, at most min(strips, blocks) of them at a time: a sweep with
    // hundreds of DPs fills the GPU with wide tiles (few block-boundary records, short fill/drain share), a rank
    // that holds few DPs (the 8-GPU shard) needs narrow ones or its waves sit idle behind the dependencies.
    {
        const int ncu = ctx->ncu;
        auto pick = [&](int kind, int R, int cmax, int cmin, int slots) {
            int C = cmax;
            for (; C > cmin; C /= 2) {
                long long par = 0;
                for (int i = 0; i < n; ++i) {
                    const DevProb &d = b->dp[i];
                    if (d.kind != kind) continue;
                    const int ns = (d.a.right - d.a.left + R - 1) / R, nb = (d.b.right - d.b.left + C - 1) / C;
                    par += std::min(ns, nb);
                }
                if (4 * par >= 5 * (long long) slots) break;
            }
            return C;
        };
        b->v3_cols = pick(1, 64, 128, 32, ncu * 4);
        // v2 (_pf): 16-row strips in 2-wave workgroups when the batch offers enough tiles (less barrier idling, 7 % faster
        // on a full sweep), 32-row strips otherwise (half as many strips on a DP's critical path)
        b->v2_threads = 128;
        if (pick(2, 16, 128, 64, ncu * 6) != 128) b->v2_threads = 256;      // (measured crossover: between 1/4 and 1/8 of the bench sweep)
        if (const char *e = g2g_opt(ctx, "V2_THREADS")) { const int t = atoi(e); if (t == 128 || t == 256) b->v2_threads = t; }
        b->v2_cols = pick(2, b->v2_threads / 8, G2G_V2_TILE_COLS, 64, ncu * (768 / b->v2_threads));
        if (const char *e = g2g_opt(ctx, "V2_COLS")) { const int c = atoi(e); if (c >= 16 && c <= 4096) b->v2_cols = c; }
        if (const char *e = g2g_opt(ctx, "V3_COLS")) { const int c = atoi(e); if (c >= 16 && c <= 4096) b->v3_cols = c; }
        // v6 or v2 for the _pf DPs of this batch?  v6 has the higher throughput (a full sweep: 750 vs 800 ms) but the longer critical
        // path per DP (64-row strips at one wave per SIMD, ~19 us per step): a batch that cannot fill the machine -- a rank's share of
        // a sharded sweep -- finishes sooner on v2's 16-row strips.  Measured on shares of the bench sweep (v6 / v2, ms): 1/2 436 / 456,
        // 1/3 405 / 311, 1/4 327 / 254, 1/8 283 / 163.  Threshold: 35 strips of 64 rows per CU (between the 1/2 and the 1/3 share).
        {
            long long s6 = 0;
            for (int i = 0; i < n; ++i) if (b->dp[i].kind == 2) s6 += (b->dp[i].a.right - b->dp[i].a.left + 63) / 64;
            long long min_strips = 35LL * ncu;
            if (const char *e = g2g_opt(ctx, "V6_MIN_STRIPS")) min_strips = atoll(e);
            b->v6_on = s6 >= min_strips;
        }
        b->v3_sweep = g2g_opt(ctx, "V3_SWEEP") ? atoi(g2g_opt(ctx, "V3_SWEEP")) : 1;
        b->v2_sweep = g2g_opt(ctx, "V2_SWEEP") ? atoi(g2g_opt(ctx, "V2_SWEEP")) : 1;
    }
    // index lists for the two forward kernels (filled below, once eligibility is known)
    const size_t idx_off = bl.put(0, 0);
    bl.extend(idx_off + sizeof(int) * 2 * (size_t) (n > 0 ? n : 1));
    b->in_bytes = (bl.size() + 255) & ~(size_t) 255;
    // 2. state / trace / outputs (device only)
    size_t off = b->in_bytes;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t) 15; return o; };
    for (int i = 0; i < n; ++i) {
        DevProb &d = b->dp[i];
        if (d.kind < 0) continue;
        const g2g_problem *p = prob[i];
        const size_t W = d.width;
        if (d.kind == 1) { d.capa = p->a.gfq.hetero + 1; d.capb = 0; }
        else if (d.kind == 2) { d.capa = p->a.gfq.hetero + 1; d.capb = p->b.gfq.hetero + 1; }
        else if (d.kind == 3) { d.capa = p->a.many; d.capb = p->b.many; }
        auto v1_state = [&]() {                // the diagonal-indexed row buffers of g2g_forward_kernel (v1): only for DPs it runs
            for (int x = 0; x < NX; ++x) {
                if (d.noll != 3 && (x == XG2 || x == XF2)) continue;
                d.val[x] = OFF<double>(take(sizeof(double) * W));
                d.dir[x] = OFF<uint8_t>(take(W));
                if (d.kind == 1 || d.kind == 2) d.dla[x] = OFF<int2>(take(sizeof(int2) * W * d.capa));
                if (d.kind == 2) d.dlb[x] = OFF<int2>(take(sizeof(int2) * W * d.capb));
                if (d.kind == 1) d.glb[x] = OFF<int>(take(sizeof(int) * W));
                if (d.kind == 3) d.glb[x] = OFF<int>(take(sizeof(int) * W * (d.capa + d.capb)));
            }
            d.spw = (int) W;
        };
        // anti-diagonal extent and the widest anti-diagonal
        const int al = p->a.left, ar = p->a.right, bl_ = p->b.left, br = p->b.right;
        d.d0 = al + bl_; d.d1 = (ar - 1) + (br - 1);
        int tmax = 1;
        long long cells = 0;
        for (int dd = d.d0; dd <= d.d1; ++dd) {
            int mlo, mhi;
            diag_rows(dd, al, ar, bl_, br, d.lw, d.up, &mlo, &mhi);
            int c = mhi - mlo + 1;
            if (c > 0) cells += c;
            if (c > tmax) tmax = c;
        }
        b->cells[i] = cells; d.cells = cells;
        d.tstride = tmax;
        d.trace = OFF<uint8_t>(take((size_t) (d.d1 - d.d0 + 1) * tmax));
        // v2 kernel (gap-profile engines): packed 16-bit gap lengths and an LDS budget decide eligibility
        d.v2_ok = 0;
        if (d.kind == 0 && !d.rect && !force_v1 && !g2g_opt(ctx, "FORCE_V1") && !g2g_opt(ctx, "NO_V7")) d.v2_ok = 7;      // DPunit: strips without gap state (the rectangular engine: g2g_forward_kernel)
        if (d.kind == 3 && !force_v1 && !g2g_opt(ctx, "FORCE_V1") && !g2g_opt(ctx, "NO_V8")) {
            // DPunit_nv: strips with the members' gap lengths in registers, for the member counts selAlnMode sends this way
            const int an = p->a.many, bn = p->b.many, ck = p->crg2_kind;
            const bool fits = ck == 11 ? (an == 1 && bn == 1) : (ck == 120 || ck == 121) ? (an == 1 && bn <= 5) :
                              (ck == 210 || ck == 211) ? (an <= 5 && bn == 1) : (ck == 220 || ck == 221) ? (an <= 3 && bn <= 3) : false;
            if (fits && p->a.gapdens && p->b.gapdens && p->a.postgapdens && p->b.postgapdens && (!(ck & 1) || ck == 11 || (p->a.weight && p->b.weight))) d.v2_ok = 8;
        }
        if ((d.kind == 1 || d.kind == 2) && !force_v1 && !g2g_opt(ctx, "FORCE_V1") && p->a.len + p->b.len < 65000) {
            // _pf: one lane per cell with rank-form merges (v6) when the rows' static lists fit the register file
            if (b->v6_on && !g2g_opt(ctx, "FORCE_V2") && !g2g_opt(ctx, "NO_V6") && d.kind == 2 && d.a.maxlist <= (d.noll == 3 ? G2G_V6_NA3 : G2G_V6_NA) && d.a.r_from_t && d.b.r_from_t &&
                v6_layout(v6_rows_bytes(d), (d.capa + 3) & ~3, v6_ring_need(p)).total <= std::max(V6_SMALL_LDS, V6_LARGE_LDS)) d.v2_ok = 6;
            else if (!g2g_opt(ctx, "FORCE_V2") && !g2g_opt(ctx, "NO_AREG") && d.kind == 1 && d.a.maxlist <= G2G_V3_NA && d.a.r_from_t &&
                v3_need(d, p, b->v3_cols, true).total <= (int) V2_LDS_MAX) d.v2_ok = 3;
            else {
                // lists too long for registers: the LDS-list one-lane-per-cell kernel, unless its LDS footprint leaves fewer than
                // three waves per CU and the 8-lanes-per-cell kernel fits (a 512 x 2048 nt DNA sweep with 17-32 entry lists and
                // Noll 3: 87 KB per strip; 28.3 s with it, 22.4 s on v2)
                // (v2 caches static gap lengths as SIGNED 16-bit with negative = terminator: a gap run can be as long as the
                // group has columns, so sides of 32768 columns or more are not eligible -- they stay on v3 / v1, which keep 32 bits)
                const bool v2fit = std::max(p->a.len, p->b.len) < 32768 &&
                    v2_lds_bytes(d.kind, d.noll, d.capa, d.capb, d.a.maxlist, d.b.maxlist, b->v2_threads) + 4 * b->v2_threads <= V2_LDS_MAX;
                const int v3tot = (!g2g_opt(ctx, "FORCE_V2") && d.kind == 1) ? v3_need(d, p, b->v3_cols).total : (int) V2_LDS_MAX + 1;
                if (v3tot <= (int) V2_LDS_MAX && (v3tot <= (int) V2_LDS_MAX / 3 || !v2fit || g2g_opt(ctx, "NO_AREG"))) d.v2_ok = 2;
                else if (v2fit) d.v2_ok = 1;
            }
        }
        if (d.nbonus) d.v2_ok = 0;               // the intron-position bonus lives in g2g_forward_kernel only
        if (!d.v2_ok) v1_state();
        else {                                  // g2g_spscore_kernel keeps its two dynamic lists in dla/dlb[XH] (stride spw)
            d.spw = 1;
            d.dla[XH] = OFF<int2>(take(sizeof(int2) * (size_t) (d.capa + 1)));
            if (d.kind == 2) d.dlb[XH] = OFF<int2>(take(sizeof(int2) * (size_t) (d.capb + 1)));
        }
        if (d.v2_ok) {
            const size_t recsz = d.v2_ok == 8 ? 4 * V8_REC : (16 + 4 * (size_t) (d.capa + (d.kind == 2 ? d.capb : 0)) + 15) & ~(size_t) 15;
            d.v2_rowstride = p->b.len + 3;
            d.v2_rowH = OFF<void>(take(3 * recsz * (size_t) d.v2_rowstride));
            d.v2_rowG = OFF<void>(take(3 * recsz * (size_t) d.v2_rowstride));
            if (d.noll == 3) d.v2_rowG2 = OFF<void>(take(3 * recsz * (size_t) d.v2_rowstride));
            d.v2_colH = OFF<void>(take(recsz * ((size_t) (ar - al) + 3)));
            d.v2_cbH = OFF<void>(take(recsz * ((size_t) (ar - al) + 3)));
            d.v2_cbF = OFF<void>(take(recsz * ((size_t) (ar - al) + 3)));
            if (d.noll == 3) d.v2_cbF2 = OFF<void>(take(recsz * ((size_t) (ar - al) + 3)));
            d.v2_rowoff = OFF<long long>(take(sizeof(long long) * ((size_t) (ar - al) + 2)));
            // the column-score matrix: only for DPs whose kernel reads one (strips in sweep mode make their own, block by block)
            const bool own_sim = !g2g_opt(ctx, "NO_SIMBLK") && (d.v2_ok >= 6 || (d.v2_ok == 1 && b->v2_sweep) || ((d.v2_ok == 2 || d.v2_ok == 3) && d.kind == 1 && b->v3_sweep));
            if (!own_sim) { d.v2_sim = OFF<double>(take(sizeof(double) * (size_t) cells + 64)); ++b->nsimmat; }
        }
        b->rr1[i] = (long long) (bl_ - al) + (br - ar);
        d.tcap = (ar - al) + (br - bl_) + 4;
        b->tcap[i] = d.tcap;
    }
    // outputs of all problems in ONE contiguous region (a single device-to-host copy fetches a whole sweep)
    b->out_lo = off;
    for (int i = 0; i < n; ++i) {
        DevProb &d = b->dp[i];
        if (d.kind < 0) continue;
        b->out_off[i] = take(sizeof(double) + sizeof(int) * 2 + sizeof(int2) * (size_t) d.tcap);
        d.score = OFF<double>(b->out_off[i]);
        d.ntrace = OFF<int>(b->out_off[i] + sizeof(double));
        d.otrace = OFF<int2>(b->out_off[i] + sizeof(double) + 2 * sizeof(int));
    }
    b->out_hi = off;
    b->arena_bytes = off + 256;
    bl.flush();
    if (bl.oom) { g2g_set_error("%s", "host staging buffer: out of (pinned) memory"); delete b; return G2G_ERR_NOMEM; }
    prep_lap("host image of the inputs");
    hipError_t e = hipSuccess;
    b->d_arena = (char *) pool_take(ctx, b->arena_bytes, &b->arena_cap);
    if (!b->d_arena) e = hipErrorOutOfMemory;
    prep_lap("hipMalloc(arena)");
    if (e != hipSuccess) {
        g2g_set_error("hipMalloc(arena): %s", hipGetErrorString(e));
        (void) hipGetLastError();              // a failed allocation must not poison the next launch check of this thread
        delete b; return G2G_ERR_NOMEM;
    }
    for (int i = 0; i < n; ++i) {
        DevProb &d = b->dp[i];
        if (d.kind < 0) continue;
        rebase(d.simmtx, b->d_arena);
        rebase(d.bon_m, b->d_arena); rebase(d.bon_n, b->d_arena); rebase(d.bon_h, b->d_arena); rebase(d.bon_mx, b->d_arena);
        rebase_side(d.a, b->d_arena); rebase_side(d.b, b->d_arena);
        for (int x = 0; x < NX; ++x) {
            rebase(d.val[x], b->d_arena); rebase(d.dir[x], b->d_arena); rebase(d.dla[x], b->d_arena);
            rebase(d.dlb[x], b->d_arena); rebase(d.glb[x], b->d_arena);
        }
        rebase(d.v2_rowH, b->d_arena); rebase(d.v2_rowG, b->d_arena); rebase(d.v2_rowG2, b->d_arena); rebase(d.v2_colH, b->d_arena);
        rebase(d.v2_cbH, b->d_arena); rebase(d.v2_cbF, b->d_arena); rebase(d.v2_cbF2, b->d_arena);
        rebase(d.v2_rowoff, b->d_arena); rebase(d.v2_sim, b->d_arena);
        rebase(d.trace, b->d_arena); rebase(d.score, b->d_arena); rebase(d.ntrace, b->d_arena); rebase(d.otrace, b->d_arena);
    }
    for (const DevPatch &pt : patches) *pt.field = pt.value;
    {
        int *i1 = (int *) (bl.data() + idx_off), *i2 = i1 + (n > 0 ? n : 1);
        b->n1 = b->n2 = 0; b->lds2 = 0;
        for (int i = 0; i < n; ++i) {
            const DevProb &d = b->dp[i];
            if (d.kind < 0) continue;
            if (d.v2_ok) { i2[b->n2++] = i; if (d.v2_ok == 1) b->lds2 = std::max(b->lds2, v2_lds_bytes(d.kind, d.noll, d.capa, d.capb, d.a.maxlist, d.b.maxlist, b->v2_threads)); }
            else i1[b->n1++] = i;
        }
        b->d_idx1 = (int *) (b->d_arena + idx_off);
        b->d_idx2 = b->d_idx1 + (n > 0 ? n : 1);
    }
    // v2 tiles: (strip i of R rows) x (block j of C columns, C chosen per batch); per kernel variant one queue
    // ordered by wavefront i + j; one completion flag per tile slot (empty slots count as done for ever)
    b->d_tiles = 0; b->tiles_cap = b->flags_cap = 0; b->d_idxp = 0; b->np = 0; b->ntiles = 0; b->lds2p = 0; b->v2_maxrows = 1; b->v2_maxcols = 1; b->simtile_lds = 0; b->d_flags = 0; b->nflags = 0; b->gen = 0;
    {
        std::vector<std::vector<std::vector<V2Tile> > > q(G2G_HDR);   // [variant][wavefront] -> tiles
        std::vector<int> flags(G2G_HDR + G2G_HDRN, 0);           // queue heads, then the header of the waits (g2g_wait_ge)
        std::vector<V2Tile> pre[G2G_HDR];                 // boundary chains of sweep-mode DPs: they head their variant's queue
        std::vector<int> ip;                              // the other DPs: chains in the prologue kernel
        const bool chainq = !g2g_opt(ctx, "NO_CHAINQ");
        int v6rows[6] = {0, 0, 0, 0, 0, 0}, v6ca4[6] = {0, 0, 0, 0, 0, 0};
        V6Ring v6rs[6] = {{{32, 32, 32}}, {{32, 32, 32}}, {{32, 32, 32}}, {{32, 32, 32}}, {{32, 32, 32}}, {{32, 32, 32}}};
        V3Need need[8];
        memset(need, 0, sizeof need);
        for (int i = 0; i < n; ++i) {
            const DevProb &d = b->dp[i];
            if (d.kind < 0 || !d.v2_ok) continue;
            const size_t recsz = d.v2_ok == 8 ? 4 * V8_REC : (16 + 4 * (size_t) (d.capa + (d.kind == 2 ? d.capb : 0)) + 15) & ~(size_t) 15;
            b->lds2p = std::max(b->lds2p, 5 * recsz + 64);
            b->v2_maxrows = std::max(b->v2_maxrows, d.a.right - d.a.left);
            b->v2_maxcols = std::max(b->v2_maxcols, d.b.right - d.b.left);
            if (d.a.nelm > 0 && sim_tiled_kind(d.sim2_kind)) {        // a's profile rows + b's frequency vectors or residues
                const bool vecb = d.sim2_kind == 33 || d.sim2_kind == 330;
                b->simtile_lds = std::max(b->simtile_lds, (size_t) 8 * SIM_TR * ((d.a.nelm - d.a.felm + 1) & ~1) + (vecb ? (size_t) 8 * SIM_TC * (d.b.felm > 0 ? d.b.felm : 0) : (size_t) SIM_TC * d.b.many) + 64);
            }
            const int al = d.a.left, ar = d.a.right, bl_ = d.b.left, br = d.b.right;
            const int R = d.v2_ok >= 2 ? 64 : b->v2_threads / 8;
            const bool swp3 = (d.v2_ok == 3 || d.v2_ok == 2) && d.kind == 1 && b->v3_sweep;   // one tile per strip, pipelined (kind 1 only: no column pool)
            const bool swp2 = (d.v2_ok == 1 && b->v2_sweep) || d.v2_ok >= 6;     // (v6 and v7 know sweep mode only)
            const int C = (swp3 || swp2) ? (1 << 20) : d.v2_ok >= 2 ? b->v3_cols : b->v2_cols;
            const int nstrip = (ar - al + R - 1) / R, nblk = (br - bl_ + C - 1) / C;
            // (one LDS plan per launch = the largest of its DPs: DPs whose column lists need a big ring get a launch of their own,
            //  or a handful of balanced divisions would cost every strip of the sweep its occupancy)
            int v6cls = 0;
            if (d.v2_ok == 6) { const int tot = v6_layout(v6_rows_bytes(d), (d.capa + 3) & ~3, v6_ring_need(prob[i])).total; v6cls = (d.noll == 3 ? 1 : 0) + (tot > V6_SMALL_LDS ? 4 : tot > V6_CLASS_A ? 2 : 0); }
            const int var = d.v2_ok == 8 ? 18 + (d.noll == 3 ? 1 : 0) : d.v2_ok == 7 ? 16 + (d.noll == 3 ? 1 : 0) : d.v2_ok == 6 ? v6_slot(v6cls) : (d.v2_ok - 1) * 4 + (d.kind == 2 ? 2 : 0) + (d.noll == 3 ? 1 : 0);
            b->var_cells[var] += b->cells[i];
            if (d.v2_ok == 6) {
                v6rows[v6cls] = std::max(v6rows[v6cls], v6_rows_bytes(d));
                v6ca4[v6cls] = std::max(v6ca4[v6cls], (d.capa + 3) & ~3);
                { const V6Ring rn = v6_ring_need(prob[i]); for (int q = 0; q < 3; ++q) v6rs[v6cls].rs[q] = std::max(v6rs[v6cls].rs[q], rn.rs[q]); }
            } else if (d.v2_ok >= 7) {
            } else if (d.v2_ok >= 2) {
                const V3Need nd = v3_need(d, prob[i], C, d.v2_ok == 3);
                V3Need &x = need[var - 4];
                x.rows_bytes = std::max(x.rows_bytes, nd.rows_bytes); x.ca4 = std::max(x.ca4, nd.ca4);
                x.apool = std::max(x.apool, nd.apool); x.bpool = std::max(x.bpool, nd.bpool);
            }
            const bool cq = (chainq && (swp2 || swp3)) || d.v2_ok >= 7;      // (v7's and v8's chains always run as queue entries: the prologue kernel knows the gap-profile kinds only)
            int ftop = -1, fleft = -1;
            if (cq) {
                ftop = (int) flags.size(); fleft = ftop + G2G_FSTRIDE;
                flags.resize(flags.size() + 2 * G2G_FSTRIDE, 0);
                V2Tile T;
                T.prob = i; T.tj = 0; T.nsteps = 0; T.dep_up = T.dep_left = T.dep_diag = T.dep_war = -1;
                T.ti = -1; T.self = ftop; pre[var].push_back(T);
                T.ti = -2; T.self = fleft; pre[var].push_back(T);
            } else ip.push_back(i);
            const int fbase = (int) flags.size();
            flags.resize(flags.size() + (size_t) nstrip * nblk * G2G_FSTRIDE, 0x7fffffff);
            for (int ti = 0; ti < nstrip; ++ti) {
                const int m0 = al + ti * R;
                for (int tj = 0; tj < nblk; ++tj) {
                    const int c0 = bl_ + tj * C, c1 = std::min(c0 + C, br);
                    const int cbase = std::max(std::max(m0 + d.lw, bl_), c0);
                    int nsteps = 0;
                    for (int t = 0; t < R && m0 + t < ar; ++t) {
                        const int m = m0 + t;
                        const int lo = std::max(std::max(m + d.lw, bl_), c0), hi = std::min(std::min(m + d.up + 1, br), c1);
                        if (hi > lo) nsteps = std::max(nsteps, hi - cbase + t);
                    }
                    if (!nsteps) continue;
                    V2Tile T;
                    T.prob = i; T.ti = ti; T.tj = tj; T.nsteps = nsteps;
                    T.self = fbase + (ti * nblk + tj) * G2G_FSTRIDE;
                    T.dep_up = ti > 0 ? T.self - nblk * G2G_FSTRIDE : ftop;             // sweep mode: the top chain is strip 0's "strip above"
                    T.dep_left = tj > 0 ? T.self - G2G_FSTRIDE : fleft;
                    T.dep_diag = (ti > 0 && tj > 0) ? T.self - (nblk + 1) * G2G_FSTRIDE : -1;
                    T.dep_war = (ti > 1 && tj + 1 < nblk) ? T.self - (2 * nblk - 1) * G2G_FSTRIDE : -1;
                    flags[T.self] = 0;
                    const int k = ti + tj;
                    if ((int) q[var].size() <= k) q[var].resize(k + 1);
                    q[var][k].push_back(T);
                }
            }
        }
        std::vector<V2Tile> all;
        for (int v = 0; v < G2G_HDR; ++v) {
            b->var_off[v] = (int) all.size();
            all.insert(all.end(), pre[v].begin(), pre[v].end());
            for (size_t k = 0; k < q[v].size(); ++k) all.insert(all.end(), q[v][k].begin(), q[v][k].end());
        }
        b->var_off[G2G_HDR] = (int) all.size();
        for (int v = 0; v < 6; ++v) b->v6lds[v] = v6_layout(v6rows[v], v6ca4[v], v6rs[v]);
        for (int v = 0; v < 8; ++v) b->v3lds[v] = v3_layout(need[v].rows_bytes, need[v].ca4, need[v].apool, need[v].bpool, b->v3_cols);
        // test hook: G2G_INJECT_STALL=<i> makes the first strip / tile of problem i depend on a flag nobody ever writes
        if (const char *e = g2g_opt(ctx, "INJECT_STALL")) {
            const int victim = atoi(e);
            const int never = (int) flags.size();
            flags.resize(flags.size() + G2G_FSTRIDE, 0);
            for (size_t k = 0; k < all.size(); ++k)
                if (all[k].prob == victim && all[k].ti >= 0) { all[k].dep_up = never; break; }
        }
        b->fail_off = (int) flags.size();                 // per-DP fail flags behind the tile flags
        flags.resize(flags.size() + (size_t) (n > 0 ? n : 1), 0);
        b->dump_off = (int) flags.size();                 // the first time-out's view of its whole DP (g2g_wait_ge: G2G_DUMP_STRIPS strips x 7 words)
        flags.resize(flags.size() + 2 + G2G_DUMP_WORDS * G2G_DUMP_STRIPS, 0);
        b->ntiles = (long long) all.size();
        b->nflags = (int) flags.size();
        b->flags0 = flags;
        if (!all.empty()) {
            hipError_t e2 = hipSuccess;
            b->d_tiles = (V2Tile *) pool_take(ctx, sizeof(V2Tile) * all.size() + sizeof(int) * (ip.size() + 1), &b->tiles_cap);
            if (!b->d_tiles) e2 = hipErrorOutOfMemory;
            if (e2 == hipSuccess) e2 = hipMemcpy(b->d_tiles, all.data(), sizeof(V2Tile) * all.size(), hipMemcpyHostToDevice);
            b->d_idxp = (int *) (b->d_tiles + all.size()); b->np = (int) ip.size();
            if (e2 == hipSuccess && b->np) e2 = hipMemcpy(b->d_idxp, ip.data(), sizeof(int) * ip.size(), hipMemcpyHostToDevice);
            if (e2 == hipSuccess) { b->d_flags = (int *) pool_take(ctx, sizeof(int) * flags.size(), &b->flags_cap); if (!b->d_flags) e2 = hipErrorOutOfMemory; }
            if (e2 == hipSuccess) e2 = hipMemcpy(b->d_flags, flags.data(), sizeof(int) * flags.size(), hipMemcpyHostToDevice);
            if (e2 != hipSuccess) { g2g_set_error("tiles: %s", hipGetErrorString(e2)); (void) hipGetLastError(); g2g_batch_free(b); return G2G_ERR_NOMEM; }
        }
    }
    prep_lap("descriptors, tiles, flags");
    if (n) memcpy(bl.data() + probs_off, b->dp.data(), sizeof(DevProb) * (size_t) n);
    b->d_probs = (DevProb *) (b->d_arena + probs_off);
    e = hipMemcpyAsync(b->d_arena, bl.data(), bl.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { g2g_set_error("upload: %s", hipGetErrorString(e)); g2g_batch_free(b); return G2G_ERR_DEVICE; }
    prep_lap("upload");
    *out = b;
    return G2G_OK;
}


// ---- CU shares ---------------------------------------------------------------------------------------------------
// The persistent launches of a sweep used to share every CU: whichever workgroup the dispatcher picked next took the LDS
// and the wave slots that had just become free, so strips of up to four kernels sat side by side on a CU.  A run that
// fills the machine now gives each launch a SHARE of the CUs of its own (hipExtStreamCreateWithCUMask): the chip is cut
// into 32 units -- unit u = CU slot u of every XCD, which is how the mask bits map on this part (bit i -> XCD i % 8,
// CU slot i / 8: tools/probes/cumask_probe.hip) -- and a launch gets a contiguous range of units in proportion to its
// estimated work, capped by what its strips can occupy.  No CU ever holds workgroups of two launches, every launch
// still spreads over all eight XCDs (L2s), and a grid is sized for ITS CUs.  Small runs (a refinement window) keep the
// whole chip for every launch: there a DP's critical path is what counts.
static size_t mstream_cap(const g2g_ctx *c)
{
    size_t cap = 8;
    if (const char *e = g2g_opt(c, "MSTREAM_MAX")) { const int v = atoi(e); if (v >= 2 && v <= 64) cap = (size_t) v; }
    return cap;
}
static hipStream_t cu_share_stream(g2g_ctx *c, int lo, int n)
{
    for (auto &m : c->mstream) if (m.lo == lo && m.n == n) { m.used = ++c->mstamp; return m.s; }
    // Every share is a hardware queue of its own, on top of the launch streams' (GPU_MAX_HW_QUEUES).  Somewhere above twenty queues the
    // hardware scheduler starts to take queues off the machine in turn -- persistent launches then stand still for tens to hundreds of
    // ms at a time (seen as gaps by the waiting waves: g2g_ctx_wait_gaps) -- so the shares in use are kept few.
    const size_t cap = mstream_cap(c);
    if (c->mstream.size() >= cap) {                         // forget the least recently used share
        size_t k = 0;
        for (size_t i = 1; i < c->mstream.size(); ++i) if (c->mstream[i].used < c->mstream[k].used) k = i;
        hipStreamSynchronize(c->mstream[k].s); hipStreamDestroy(c->mstream[k].s);
        c->mstream.erase(c->mstream.begin() + k);
    }
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int bit = 8 * lo; bit < 8 * (lo + n) && bit < 256; ++bit) mask[bit >> 5] |= 1u << (bit & 31);
    g2g_ctx::MStream m; m.lo = lo; m.n = n; m.s = 0; m.used = ++c->mstamp;
    if (hipExtStreamCreateWithCUMask(&m.s, 8, mask) != hipSuccess || !m.s) { (void) hipGetLastError(); return 0; }
    c->mstream.push_back(m);
    ++c->n_mstreams;
    return m.s;
}
// relative cost of a cell on the kernel of variant slot v (what the shares are proportional to)
static double variant_cost(int v)
{
    const double n3 = (v & 1) ? 1.4 : 1.0;                  // odd slots: Noll 3
    if (v < 2) return 2.0 * n3;                             // v2 _hf
    if (v < 4) return 5.0 * n3;                             // v2 _pf
    if (v < 8) return 2.5 * n3;                             // v3 with LDS lists
    if (v < 12) return 1.0 * n3;                            // v3r
    if (v < 16 || v >= 20) return 2.7 * n3;                 // v6
    if (v < 18) return 0.8 * n3;                            // v7
    return 1.2 * n3;                                        // v8
}

// Compact the valid problems to the front?  No: invalid ones keep kind = -1 and the kernels skip them.
extern "C" int g2g_batch_run(g2g_batch *b)
{
    if (!b) return G2G_ERR_ARG;
    g2g_ctx *ctx = b->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    if (b->n == 0) return G2G_OK;
    std::fill(b->was_recovered.begin(), b->was_recovered.end(), 0);
    b->last_timeouts = b->last_recovered = 0;
    if (!b->is_retry) { ++ctx->n_runs; ++g_tot[0]; }
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    if (b->n2) {
        const int T2 = b->v2_threads;
        // Tile-mode DPs (test configurations; sweep mode, the default, needs none of this) read a column-score matrix and take
        // their boundary chains from a prologue kernel: row offsets first (tiny), then the chains (single-lane, latency-bound) on
        // a side stream while the score kernels (fully parallel) fill the GPU on the main one
        const int pro_off = g2g_opt(ctx, "NO_PROSTAGE") ? 0 : (int) ((b->lds2p + 15) & ~(size_t) 15);
        if (b->nsimmat || b->np) {
            if (b->nsimmat) {
                hipLaunchKernelGGL(g2g_v2_rowoff_kernel, dim3((b->n2 + 63) / 64), dim3(64), 0, ctx->stream,
                                   (const DevProb *) b->d_probs, (const int *) b->d_idx2, b->n2);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(ctx->vev[G2G_NVS], ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->vstream[G2G_NVS - 1], ctx->vev[G2G_NVS], 0));
            if (b->np) {
                hipLaunchKernelGGL(g2g_v2_prologue_kernel, dim3(b->np), dim3(128), pro_off ? pro_off + 2 * PRO_LDS_BYTES : b->lds2p, ctx->vstream[G2G_NVS - 1],
                                   (const DevProb *) b->d_probs, (const int *) b->d_idxp, pro_off);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(ctx->vev[G2G_NVS - 1], ctx->vstream[G2G_NVS - 1]));
            if (b->nsimmat) {
                const int simtiled = (!g2g_opt(ctx, "NO_SIMTILE") && b->simtile_lds && b->simtile_lds <= 64 * 1024) ? 1 : 0;
                if (simtiled) {
                    hipLaunchKernelGGL(g2g_v2_sim_tile_kernel, dim3((b->v2_maxcols + SIM_TC - 1) / SIM_TC, (b->v2_maxrows + SIM_TR - 1) / SIM_TR, b->n2), dim3(256),
                                       b->simtile_lds, ctx->stream, (const DevProb *) b->d_probs, (const int *) b->d_idx2);
                    HIPCHK(hipGetLastError());
                }
                hipLaunchKernelGGL(g2g_v2_sim_kernel, dim3(b->v2_maxrows, b->n2), dim3(256), 0, ctx->stream,
                                   (const DevProb *) b->d_probs, (const int *) b->d_idx2, simtiled);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->vev[G2G_NVS - 1], 0));
        }
        if (g2g_opt(ctx, "DEBUG")) { hipError_t e3 = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[g2g] prologue+sim done: %s\n", hipGetErrorString(e3)); fflush(stderr); }
        // persistent tile / strip kernels: one launch per kernel variant, each on its own stream (they are independent)
        typedef void (*v2k_t)(const DevProb *, const V2Tile *, int, int *, int *, int, int, int, int, int, double *);
        static const v2k_t v2k[4] = {g2g_v2_hf2, g2g_v2_hf3, g2g_v2_pf2, g2g_v2_pf3};
        typedef void (*v3k_t)(const DevProb *, const V2Tile *, int, int *, int *, int, V3Lds, int, int, int, double *);
        static const v3k_t v3k[8] = {g2g_v3_hf2, g2g_v3_hf3, 0, 0, g2g_v3r_hf2, g2g_v3r_hf3, 0, 0};       // (_pf strips: g2g_v6_*, below)
        // one persistent launch per variant, each on its own stream (they are independent of each other)
        ++b->gen;
        if ((b->gen & 0x7FF) == 0 && b->d_flags) {             // sweep-mode progress counters carry gen & 0x7FF: start over
            HIPCHK(hipMemcpyAsync(b->d_flags, b->flags0.data(), sizeof(int) * b->flags0.size(), hipMemcpyHostToDevice, ctx->stream));
            ++b->gen;
        }
        {   // queue heads; header of the waits (g2g_wait_ge): time-outs, first slot, offset of the fail flags, wall-clock limit
            // (the image lives in the batch: an asynchronous copy from pageable memory may read its source after the call returns)
            int *hdr = b->hdr_img;
            memset(hdr, 0, sizeof b->hdr_img);
            // a legitimate wait lasts as long as the strip above needs for one publish interval, or a boundary chain for its
            // sequential walk: milliseconds, tens of ms for a strip pulled long before its producer gets going (under 180 ms in every
            // ordinary sweep observed).  Half a second of wall clock means the producer is not running; a false alarm only costs a re-run.
            double limit_ms = 500.;
            if (const char *e = g2g_opt(ctx, "WAIT_LIMIT_MS")) { const double v = atof(e); if (v > 0) limit_ms = v; }
            hdr[G2G_HDR + 2] = b->fail_off;
            hdr[G2G_HDR + 82] = b->dump_off;
            hdr[G2G_HDR + 3] = (int) std::min(2.0e9, limit_ms * ctx->rt_ticks_per_ms / 65536.) + 1;
            HIPCHK(hipMemcpyAsync(b->d_flags, hdr, sizeof b->hdr_img, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemsetAsync(b->d_flags + b->fail_off, 0, sizeof(int) * (size_t) (b->n > 0 ? b->n : 1), ctx->stream));
        }
        HIPCHK(hipEventRecord(ctx->vev[G2G_NVS], ctx->stream));
        int nlaunch = 0;                                      // every persistent launch of this run takes the next stream
        const int ncu = ctx->ncu;
        auto sim_scratch = [&](int slot, int grid) -> double * {
            const size_t need = (size_t) grid * G2G_SIMBLK_STRIDE * sizeof(double);
            if (b->simscr_cap[slot] < need) {                 // (only in the dry pass of a run: see below)
                pool_give(ctx, b->simscr[slot], b->simscr_cap[slot]);
                b->simscr[slot] = 0; b->simscr_cap[slot] = 0;
                b->simscr[slot] = (double *) pool_take(ctx, need, &b->simscr_cap[slot]);
                if (!b->simscr[slot]) { b->simscr_cap[slot] = 0; return (double *) 0; }
            }
            return b->simscr[slot];
        };
        // ---- CU shares of this run's persistent launches (cu_share_stream) ----
        int sh_lo[G2G_HDR], sh_n[G2G_HDR];
        hipStream_t sh_stream[G2G_HDR];
        for (int v = 0; v < G2G_HDR; ++v) sh_stream[v] = 0;
        bool shares = false;
        {
            const int T2s = b->v2_threads;
            auto wpc_of = [&](int v) -> int {                 // resident workgroups per CU of variant slot v
                if (v < 4) return std::max(1, std::min(2048 / T2s, (int) (V2_LDS_MAX / (b->lds2 + 4 * (size_t) T2s))));
                if (v < 12) { const int t = b->v3lds[v - 4].total; return t > 0 ? std::max(1, std::min(16, (int) (V2_LDS_MAX / (size_t) t))) : 1; }
                if (v < 16 || v >= 20) { const int cls = v < 16 ? v - 12 : v - 16; const int t = b->v6lds[cls].total; return t > 0 ? std::max(1, std::min(4, (int) (V2_LDS_MAX / (size_t) t))) : 1; }
                return v < 18 ? 16 : 4;
            };
            double work[G2G_HDR], tot = 0;
            int need[G2G_HDR], nl = 0;
            double demand = 0;
            for (int v = 0; v < G2G_HDR; ++v) {
                const int cnt = b->var_off[v + 1] - b->var_off[v];
                work[v] = cnt ? (double) std::max<long long>(b->var_cells[v], 1) * variant_cost(v) : 0;
                need[v] = cnt ? std::min(32, std::max(1, (cnt + 8 * wpc_of(v) - 1) / (8 * wpc_of(v)))) : 0;   // units its tiles can occupy
                sh_lo[v] = 0; sh_n[v] = 0;
                if (cnt) { ++nl; tot += work[v]; demand += (double) cnt / wpc_of(v); }
            }
            // Default: shares when no v6 launch is in the run -- the regime of a window of g2g_refine and of a rank's share of a sharded
            // sweep (_pf on v2 beside _hf on v3r: the two kernels slow each other down on a shared CU; 1/8 of the bench sweep 179 -> 140 ms,
            // 1/4 266 -> 245 ms, a window of the 256 x 1024 refinement 107 -> 102 ms).  Not with v6 in the run: its class of the most
            // balanced divisions is bound by its DPs' critical path, needs many CUs for a short time, and a static share leaves them idle
            // afterwards (a full sweep 1563 ms instead of 760, half of one 734 instead of 437).  CU_SHARES=0 / 1 / 2: never /
            // for runs that fill the machine twice over / always.
            bool has_v6 = false;
            for (int v = 12; v < G2G_HDR; ++v) if ((v < 16 || v >= 20) && b->var_off[v + 1] - b->var_off[v] > 0) has_v6 = true;
            const char *opt = g2g_opt(ctx, "CU_SHARES");
            const bool autosh = !opt && !has_v6;
            const bool want = opt ? atoi(opt) != 0 : autosh;
            const bool force = (opt && atoi(opt) >= 2) || autosh;
            // only a run that fills the machine more than twice over is partitioned, and only if every launch can have a unit
            if (want && ncu == 256 && nl >= 2 && nl <= (int) mstream_cap(ctx) && (force || demand >= 2.0 * ncu) && tot > 0 && !g2g_opt(ctx, "DEBUG")) {      // (256 CUs in 8 XCDs: the mask layout the shares are written for)
                int left = 32;
                double wleft = tot;
                bool done[G2G_HDR];
                for (int v = 0; v < G2G_HDR; ++v) done[v] = work[v] == 0;
                // launches whose tiles cannot fill their proportional share take what they can fill; the rest is re-divided
                for (int round = 0; round < G2G_HDR; ++round) {
                    bool changed = false;
                    for (int v = 0; v < G2G_HDR; ++v) {
                        if (done[v]) continue;
                        const double prop = wleft > 0 ? left * work[v] / wleft : 0;
                        if (need[v] <= prop) { sh_n[v] = need[v]; left -= need[v]; wleft -= work[v]; done[v] = true; changed = true; }
                    }
                    if (!changed) break;
                }
                int open_ = 0;
                for (int v = 0; v < G2G_HDR; ++v) if (!done[v]) ++open_;
                if (left >= open_) {
                    int given = 0;
                    double frac[G2G_HDR];
                    for (int v = 0; v < G2G_HDR; ++v) {
                        frac[v] = -1;
                        if (done[v]) continue;
                        const double prop = left * work[v] / wleft;
                        sh_n[v] = std::max(1, (int) prop);
                        frac[v] = prop - (int) prop;
                        given += sh_n[v];
                    }
                    while (given < left) {                       // largest remainders first
                        int best = -1;
                        for (int v = 0; v < G2G_HDR; ++v) if (frac[v] >= 0 && (best < 0 || frac[v] > frac[best])) best = v;
                        if (best < 0) break;
                        ++sh_n[best]; frac[best] = -0.5; ++given;
                    }
                    while (given > left) {                       // (the minimum of one unit each overdrew: take from the largest)
                        int big = -1;
                        for (int v = 0; v < G2G_HDR; ++v) if (!done[v] && sh_n[v] > 1 && (big < 0 || sh_n[v] > sh_n[big])) big = v;
                        if (big < 0) break;
                        --sh_n[big]; --given;
                    }
                    if (given == left) {
                        // two launches (the common case: _pf on v2 beside _hf on v3r): a split within one unit of one already in use is
                        // as good, and keeps the number of queues down
                        if (nl == 2) {
                            int v0 = -1, v1 = -1;
                            for (int v = 0; v < G2G_HDR; ++v) if (sh_n[v]) { if (v0 < 0) v0 = v; else v1 = v; }
                            if (v0 >= 0 && v1 >= 0 && sh_n[v0] + sh_n[v1] == 32) {
                                // (once the context holds its fill of shares, the nearest one is taken whatever the distance: creating and
                                //  destroying queues while launches are resident makes the scheduler rebuild its run list, which the waiting
                                //  waves see as a pause of some 10 ms)
                                const size_t capm = mstream_cap(ctx);
                                const int tol = ctx->mstream.size() + 2 > capm ? 32 : 1;
                                int best = -1;
                                for (const auto &m : ctx->mstream) {
                                    if (m.lo != 0 || m.n < 1 || m.n > 31 || abs(m.n - sh_n[v0]) > tol) continue;
                                    bool partner = false;
                                    for (const auto &q : ctx->mstream) if (q.lo == m.n && q.n == 32 - m.n) partner = true;
                                    if (partner && (best < 0 || abs(m.n - sh_n[v0]) < abs(best - sh_n[v0]))) best = m.n;
                                }
                                if (best > 0) { sh_n[v0] = best; sh_n[v1] = 32 - best; }
                            }
                        }
                        int lo = 0;
                        shares = true;
                        // ONE UNIT OF 8 CUs STAYS EMPTY BETWEEN TWO SHARES (taken from the larger one).  Every stalled pipeline head of round 4
                        // (20 of 20 events) was held by one of the LAST 24 workgroups of the `_pf` launch's 600 resident
                        // ones -- the three workgroups on each CU of the share's last unit, next to the other launch's share; with the gap:
                        // 5 whole refinements (4360 windows) without an event against 0.6 events per run before, 0.5 % slower.
                        // NO_SHARE_GAP restores adjacent shares.
                        const int gap = (!g2g_opt(ctx, "NO_SHARE_GAP") && nl == 2) ? 1 : 0;
                        if (gap) { int vb = -1; for (int v = 0; v < G2G_HDR; ++v) if (sh_n[v] && (vb < 0 || sh_n[v] > sh_n[vb])) vb = v; if (vb >= 0 && sh_n[vb] > 2) --sh_n[vb]; }
                        bool first = true;
                        for (int v = 0; v < G2G_HDR; ++v) if (sh_n[v]) { if (!first) lo += gap; first = false; sh_lo[v] = lo; lo += sh_n[v]; sh_stream[v] = cu_share_stream(ctx, sh_lo[v], sh_n[v]); if (!sh_stream[v]) shares = false; }
                        if (lo > 32) shares = false;
                    }
                }
            }
            if (g2g_opt(ctx, "WARN") && shares) {
                fprintf(stderr, "[g2g] CU shares (units of 8 CUs, one per XCD):");
                for (int v = 0; v < G2G_HDR; ++v) if (sh_n[v]) fprintf(stderr, " slot %d: %d..%d", v, sh_lo[v], sh_lo[v] + sh_n[v] - 1);
                fprintf(stderr, "\n");
            }
        }
        // (the share streams were resolved once, above: a launch never creates or evicts a queue while others are resident)
        auto launch_stream = [&](int slot, int k) -> hipStream_t { return (shares && sh_stream[slot]) ? sh_stream[slot] : ctx->vstream[k]; };
        // Two passes over the launches of this run.  The DRY pass computes every launch's grid and takes the buffers it needs
        // from the context's pool (column-score scratch, list twins); only then does the LIVE pass launch.  Between the first
        // and the last persistent launch of a run the host makes no memory-management call at all: a launch that is already
        // resident never sees the device's page tables change under it (VERDICT r03, weak 2 iv).
        int hf_events[8], n_hf_events = 0;
        for (int pass = 0; pass < 2; ++pass) {
        const bool dry = pass == 0;
        n_hf_events = 0;
        const bool dbg = !dry && g2g_opt(ctx, "DEBUG") != 0;
        nlaunch = 0;
        auto launch_cus = [&](int slot) -> int { return shares ? 8 * sh_n[slot] : ncu; };
        for (int v = 0; v < 4; ++v) {
            const int cnt = b->var_off[v + 1] - b->var_off[v];
            if (!cnt) continue;
            const int sk2 = nlaunch++ % G2G_NVS;
            hipStream_t vs2 = launch_stream(v, sk2);
            const int ncu2 = launch_cus(v);
            if (!dry) HIPCHK(hipStreamWaitEvent(vs2, ctx->vev[G2G_NVS], 0));
            int wpc2 = 2048 / T2;              // workgroups per CU the grid provides (LDS decides how many are resident)
            if (const char *e = g2g_opt(ctx, "V2_WPC")) { const int w = atoi(e); if (w >= 1 && w <= 32) wpc2 = w; }
            const int grid = std::min(cnt, ncu2 * wpc2);
            // sweep mode: the kernel argument is the publish interval.  A DP's critical path is columns + strips x (rows of a
            // strip + interval): 32 steps when the strips outnumber the resident workgroups many times over (throughput bound,
            // fewer fences), 8 / 4 when they do not (a window of g2g_refine: the batch is as slow as its longest pipeline; measured on
            // batches of 1-16 full-size DPs with tools/latency_probe.py: 4 beats 16 by 14 % at 8 DPs, 2 gains nothing more), 16 for a
            // rank's share of a sharded sweep (1/8 of the bench sweep: 180 ms against 195 with 32).
            const int res2 = ncu2 * std::max(1, std::min(wpc2, (int) (V2_LDS_MAX / (b->lds2 + 4 * (size_t) T2))));
            const int pint2 = !b->v2_sweep ? 0 : b->v2_sweep >= 2 ? b->v2_sweep : cnt <= 2 * res2 ? 4 : cnt <= 4 * res2 ? 8 : cnt <= 16 * res2 ? 16 : 32;
            if (dbg) { fprintf(stderr, "[g2g] variant %d: %d tiles, grid %d x %d threads, lds %zu, cols %d, gen %d\n", v, cnt, grid, T2, b->lds2, b->v2_cols, b->gen); fflush(stderr); }
            double *simscr2 = 0;
            if (b->v2_sweep && !g2g_opt(ctx, "NO_SIMBLK")) {
                simscr2 = sim_scratch(v, grid);
                if (!simscr2) { g2g_set_error("%s", "hipMalloc(column-score scratch)"); return G2G_ERR_NOMEM; }
            }
            if (dry) continue;
            hipLaunchKernelGGL(v2k[v], dim3(grid), dim3(T2), b->lds2 + 4 * T2, vs2,
                               (const DevProb *) b->d_probs, (const V2Tile *) (b->d_tiles + b->var_off[v]), cnt,
                               b->d_flags + v, b->d_flags, b->gen, (int) b->lds2, b->v2_sweep ? (1 << 20) : b->v2_cols, pint2,
                               (pro_off && pro_off + PRO_LDS_BYTES <= b->lds2) ? pro_off : 0, simscr2);
            HIPCHK(hipGetLastError());
            if (dbg) { hipError_t e3 = hipStreamSynchronize(vs2); fprintf(stderr, "[g2g] variant %d done: %s\n", v, hipGetErrorString(e3)); fflush(stderr); }
            HIPCHK(hipEventRecord(ctx->vev[sk2], vs2));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->vev[sk2], 0));
        }
        // (the _hf launch goes first on purpose: submitted behind the _pf launches -- whose persistent workgroups hold the LDS of every
        //  CU until their queues are empty -- it runs after them instead of beside them: 845 ms per bench sweep instead of 757)
        for (int v = 0; v < 8; ++v) {
            const int cnt = b->var_off[v + 5] - b->var_off[v + 4];
            if (!cnt || !v3k[v]) continue;
            if (const char *e = g2g_opt(ctx, "ONLY_VAR")) if (atoi(e) != v) continue;       // profiling aid
            const int sk3 = nlaunch++ % G2G_NVS;
            hipStream_t vs = launch_stream(v + 4, sk3);
            const int ncu3 = launch_cus(v + 4);
            const V3Lds &LO = b->v3lds[v];
            const bool swpv = (v & 3) < 2;                   // the _hf variants (LDS lists 0,1; register lists 4,5) run in sweep mode
            if (!dry) HIPCHK(hipStreamWaitEvent(vs, ctx->vev[G2G_NVS], 0));
            int wpc = (int) (V2_LDS_MAX / (size_t) LO.total);              // resident tiles per CU (LDS-bound)
            if (wpc < 1) wpc = 1; if (wpc > 16) wpc = 16;
            if (const char *e = g2g_opt(ctx, "V3_WPC")) { const int w = atoi(e); if (w >= 1 && w <= 32) wpc = w; }
            const int grid = std::min(cnt, ncu3 * wpc);
            if (dbg) { fprintf(stderr, "[g2g] v3 variant %d: %d tiles, grid %d, lds %d (rows %d, apool@%d, bpool@%d), cols %d, gen %d\n", v, cnt, grid, LO.total, LO.black, LO.aglen, LO.bglen, b->v3_cols, b->gen); fflush(stderr); }
            double *simscr3 = 0;
            if (swpv && b->v3_sweep && !g2g_opt(ctx, "NO_SIMBLK")) {
                simscr3 = sim_scratch(4 + v, grid);
                if (!simscr3) { g2g_set_error("%s", "hipMalloc(column-score scratch)"); return G2G_ERR_NOMEM; }
            }
            if (dry) continue;
            hipLaunchKernelGGL(v3k[v], dim3(grid), dim3(64), (size_t) LO.total, vs,
                               (const DevProb *) b->d_probs, (const V2Tile *) (b->d_tiles + b->var_off[v + 4]), cnt,
                               b->d_flags + 4 + v, b->d_flags, b->gen, LO, (swpv && b->v3_sweep) ? (1 << 20) : b->v3_cols,
                               !(swpv && b->v3_sweep) ? 0 : b->v3_sweep >= 2 ? b->v3_sweep : cnt <= ncu3 * std::min(wpc, 8) ? 4 : cnt < 4 * ncu3 * std::min(wpc, 8) ? 16 : 32,
                               (pro_off && pro_off + (int) PRO_LDS_BYTES <= LO.svals) ? pro_off : 0, simscr3);
            HIPCHK(hipGetLastError());
            if (dbg) { const auto t0 = std::chrono::steady_clock::now(); hipError_t e3 = hipStreamSynchronize(vs); fprintf(stderr, "[g2g] v3 variant %d done: %s, %.1f ms\n", v, hipGetErrorString(e3), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); fflush(stderr); }
            HIPCHK(hipEventRecord(ctx->vev[sk3], vs));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->vev[sk3], 0));
            hf_events[n_hf_events++] = sk3;
        }
        for (int vi = 0; vi < 6; ++vi) {
            const int v = 5 - vi;                    // (the larger-footprint launch first)
            const int slot = v6_slot(v);
            const int cnt = b->var_off[slot + 1] - b->var_off[slot];
            if (!cnt) continue;
            typedef void (*v6k_t)(const DevProb *, const V2Tile *, int, int *, int *, int, V6Lds, int, int, double *, unsigned *, int);
            static const v6k_t v6k[6] = {g2g_v6_pf2, g2g_v6_pf3, g2g_v6_pf2, g2g_v6_pf3, g2g_v6_pf2, g2g_v6_pf3};
            const int jev = nlaunch++ % G2G_NVS;
            hipStream_t vs = launch_stream(slot, jev);
            const int ncu6 = launch_cus(slot);
            const V6Lds &LO = b->v6lds[v];
            if (!dry) HIPCHK(hipStreamWaitEvent(vs, ctx->vev[G2G_NVS], 0));
            // The _pf launches start when the _hf launches are done.  A sweep is bound by throughput -- its time is the SUM of what the
            // launches take alone (_hf 133 ms + _pf 610 ms for the bench sweep; the two footprint classes of v6 together take what
            // they take one after the other) -- and side by side the two kernel shapes leave each other wave slots they cannot use
            // (v3r: two waves per SIMD, v6: a whole SIMD's registers): 769 -> 744 ms.  PARALLEL_HF=1: side by side as before.
            if (!dry && !g2g_opt(ctx, "PARALLEL_HF")) for (int q = 0; q < n_hf_events; ++q) HIPCHK(hipStreamWaitEvent(vs, ctx->vev[hf_events[q]], 0));
            int wpc = (int) (V2_LDS_MAX / (size_t) LO.total);              // resident strips per CU (LDS-bound)
            if (wpc < 1) wpc = 1; if (wpc > 4) wpc = 4;       // (the kernel takes a whole SIMD's registers: four strips per CU at most)
            if (const char *e = g2g_opt(ctx, "V6_WPC")) { const int w = atoi(e); if (w >= 1 && w <= 32) wpc = w; }
            const int grid = std::min(cnt, ncu6 * wpc);
            const int pint = b->v2_sweep >= 2 ? b->v2_sweep : 4 * cnt <= ncu6 * wpc ? 4 : cnt < 4 * ncu6 * wpc ? 16 : 32;   // publish interval (power of 2)
            if (dbg) { fprintf(stderr, "[g2g] v6 variant %d: %d strips, grid %d, lds %d (rings %d / %d / %d entries), publish every %d, gen %d\n", v, cnt, grid, LO.total, LO.rs[0], LO.rs[1], LO.rs[2], pint, b->gen); fflush(stderr); }
            double *simscr6 = sim_scratch(slot, grid);
            if (!simscr6) { g2g_set_error("%s", "hipMalloc(column-score scratch)"); return G2G_ERR_NOMEM; }
            // the twin image of the strips' dynamic lists (what does not fit their inline parts in LDS): two dwords per dword of rows
            const int twin_dw = 2 * (LO.black - LO.rows) / 4 + 64;
            {
                const size_t need = (size_t) grid * twin_dw * sizeof(unsigned);
                if (b->twin_cap[v] < need) {
                    pool_give(ctx, b->twin[v], b->twin_cap[v]);
                    b->twin[v] = 0; b->twin_cap[v] = 0;
                    b->twin[v] = (unsigned *) pool_take(ctx, need, &b->twin_cap[v]);
                    if (!b->twin[v]) { b->twin_cap[v] = 0; g2g_set_error("%s", "hipMalloc(list twin image)"); return G2G_ERR_NOMEM; }
                }
            }
            if (dry) continue;
            hipLaunchKernelGGL(v6k[v], dim3(grid), dim3(64), (size_t) LO.total, vs,
                               (const DevProb *) b->d_probs, (const V2Tile *) (b->d_tiles + b->var_off[slot]), cnt,
                               b->d_flags + slot, b->d_flags, b->gen, LO, pint,
                               (pro_off && pro_off + (int) PRO_LDS_BYTES <= LO.svals) ? pro_off : 0, simscr6, b->twin[v], twin_dw);
            HIPCHK(hipGetLastError());
            if (dbg) { const auto t0 = std::chrono::steady_clock::now(); hipError_t e3 = hipStreamSynchronize(vs); fprintf(stderr, "[g2g] v6 variant %d done: %s, %.1f ms\n", v, hipGetErrorString(e3), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); fflush(stderr); }
            HIPCHK(hipEventRecord(ctx->vev[jev], vs));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->vev[jev], 0));
        }
        for (int v = 0; v < 4; ++v) {                        // v7: DPunit strips (no gap state, no LDS to speak of); v8: DPunit_nv strips
            const int cnt = b->var_off[v + 17] - b->var_off[v + 16];
            if (!cnt) continue;
            typedef void (*v7k_t)(const DevProb *, const V2Tile *, int, int *, int *, int, int, double *);
            static const v7k_t v7k[4] = {g2g_v7_ngp2, g2g_v7_ngp3, g2g_v8_ntv2, g2g_v8_ntv3};
            const int sk7 = nlaunch++ % G2G_NVS;
            hipStream_t vs = launch_stream(v + 16, sk7);
            const int ncu7 = launch_cus(v + 16);
            if (!dry) HIPCHK(hipStreamWaitEvent(vs, ctx->vev[G2G_NVS], 0));
            const int wpc = v < 2 ? 16 : 4;                  // (v8 holds its records' lengths in registers: one wave per SIMD)
            const int grid = std::min(cnt, ncu7 * wpc);
            const int pint = b->v2_sweep >= 2 ? b->v2_sweep : 4 * cnt <= ncu7 * wpc ? 4 : cnt < 4 * ncu7 * wpc ? 16 : 32;
            double *simscr7 = sim_scratch(16 + v, grid);
            if (!simscr7) { g2g_set_error("%s", "hipMalloc(column-score scratch)"); return G2G_ERR_NOMEM; }
            if (dbg) { fprintf(stderr, "[g2g] %s variant %d: %d strips, grid %d, publish every %d, gen %d\n", v < 2 ? "v7" : "v8", v & 1, cnt, grid, pint, b->gen); fflush(stderr); }
            if (dry) continue;
            hipLaunchKernelGGL(v7k[v], dim3(grid), dim3(64), 0, vs, (const DevProb *) b->d_probs, (const V2Tile *) (b->d_tiles + b->var_off[v + 16]), cnt,
                               b->d_flags + 16 + v, b->d_flags, b->gen, pint, simscr7);
            HIPCHK(hipGetLastError());
            if (dbg) { const auto t0 = std::chrono::steady_clock::now(); hipError_t e3 = hipStreamSynchronize(vs); fprintf(stderr, "[g2g] %s variant %d done: %s, %.1f ms\n", v < 2 ? "v7" : "v8", v & 1, hipGetErrorString(e3), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); fflush(stderr); }
            HIPCHK(hipEventRecord(ctx->vev[sk7], vs));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->vev[sk7], 0));
        }
        }   // pass
}
    if (b->n1) {
        if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] v1 (state in HBM): %d of %d problems\n", b->n1, b->n); fflush(stderr); }
        hipLaunchKernelGGL(g2g_forward_kernel, dim3(b->n1), dim3(G2G_FWD_THREADS), 0, ctx->stream,
                           (const DevProb *) b->d_probs, (const int *) b->d_idx1);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    hipLaunchKernelGGL(g2g_traceback_kernel, dim3(b->n), dim3(64), 0, ctx->stream, (const DevProb *) b->d_probs, b->n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (b->d_flags) {
        int rep[G2G_HDR + G2G_HDRN];
        HIPCHK(hipMemcpy(rep, b->d_flags, sizeof rep, hipMemcpyDeviceToHost));
        if (rep[G2G_HDR + 40]) {             // waves that were off the machine while they waited: evidence, not an error
            const double gap_ms = rep[G2G_HDR + 41] * 1024. / ctx->rt_ticks_per_ms;
            ctx->n_gaps += rep[G2G_HDR + 40]; g_tot[4] += rep[G2G_HDR + 40];
            if (gap_ms > ctx->max_gap_ms) ctx->max_gap_ms = gap_ms;
            if (g2g_opt(ctx, "DEBUG") || g2g_opt(ctx, "WARN")) { fprintf(stderr, "[g2g] %d waiting wave(s) were off the machine for more than 4 ms at a stretch (longest %.1f ms); batch of %d DPs, %d wait(s) lost; %zu CU-mask streams alive, %lld created\n", rep[G2G_HDR + 40], gap_ms, b->n, rep[G2G_HDR], ctx->mstream.size(), ctx->n_mstreams); fflush(stderr); }
        }
        if (rep[G2G_HDR]) {
            // Some wait ran into the wall-clock limit.  Only the DPs marked in the fail array are lost; they are re-run here, in
            // the same call, on the kernel that polls nothing (one workgroup per DP, state in HBM).
            std::vector<int> fail(b->n);
            HIPCHK(hipMemcpy(fail.data(), b->d_flags + b->fail_off, sizeof(int) * (size_t) b->n, hipMemcpyDeviceToHost));
            std::vector<int> lost;
            for (int i = 0; i < b->n; ++i) if (fail[i] && !b->status[i]) lost.push_back(i);
            b->last_timeouts = rep[G2G_HDR]; b->last_recovered = (int) lost.size();
            ctx->n_timeouts += rep[G2G_HDR]; ctx->n_recovered += (long long) lost.size();
            if (b->is_retry) ctx->n_v1 += (long long) lost.size();
            g_tot[1] += rep[G2G_HDR]; g_tot[2] += (long long) lost.size();
            if (b->is_retry) g_tot[3] += (long long) lost.size();
            if (b->injected) { g_tot[5] += rep[G2G_HDR]; g_tot[6] += (long long) lost.size(); }
            {   // the report of the event: kept in the context (g2g_ctx_last_timeout: every ordinary run that meets one carries the
                // evidence), printed under WARN / DEBUG
                char buf[6144];
                int o = 0;
                auto add = [&](const char *fmt, ...) { va_list ap; va_start(ap, fmt); if (o < (int) sizeof buf - 1) { const int w = vsnprintf(buf + o, sizeof buf - o, fmt, ap); if (w > 0) o += w; } va_end(ap); if (o > (int) sizeof buf - 1) o = (int) sizeof buf - 1; };
                int kinds[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int i : lost) { const int k = b->dp[i].v2_ok; if (k >= 0 && k < 10) ++kinds[k]; }
                add("%d waits timed out (first: queue slot %d, gen %d, batch of %d DPs): re-running %zu DP(s) %s; by kernel (0 v1, 1 v2, 2 v3, 3 v3r, 6 v6, 7 v7, 8 v8):",
                    rep[G2G_HDR], rep[G2G_HDR + 1], b->gen, b->n, lost.size(), b->is_retry ? "on g2g_forward_kernel" : "(first on the ordinary kernels)");
                for (int k = 0; k < 10; ++k) if (kinds[k]) add(" %d x kernel %d", kinds[k], k);
                add("; the first:");
                for (size_t k = 0; k < lost.size() && k < 6; ++k) add(" %d(kernel %d, %d x %d)", lost[k], b->dp[lost[k]].v2_ok, b->dp[lost[k]].a.right - b->dp[lost[k]].a.left, b->dp[lost[k]].b.right - b->dp[lost[k]].b.left);
                const int *x = rep + G2G_HDR;                 // the first time-out's snapshot (g2g_wait_ge)
                add(". First time-out: DP %d, wanted gen %d col %d of word %d, saw gen %d col %d; the words at and below it (gen:col):",
                    x[7] - b->fail_off, (x[4] >> 20) & 0x7FF, x[4] & 0xFFFFF, x[6], (x[5] >> 20) & 0x7FF, x[5] & 0xFFFFF);
                for (int k = 0; k < 8; ++k) add(" %d:%d", (x[8 + k] >> 20) & 0x7FF, x[8 + k] & 0xFFFFF);
                if (!(x[16] == 0x7fffffff || (x[16] == 0 && x[17] == 0))) add("; producer's heartbeat: step %d place %d, 50 us later step %d place %d", x[16], x[17], x[18], x[19]);
                add("; producer HW_ID %08x XCC %d, waiter HW_ID %08x XCC %d; waiters per XCC:", x[20], x[21] & 15, x[22], x[23] & 15);
                for (int k = 0; k < 8; ++k) add(" %d", x[24 + k] / 64);
                add("; their producers per XCC:");
                for (int k = 0; k < 8; ++k) add(" %d", x[32 + k] / 64);
                add("; by RMW %d:%d, loaded again %d:%d", (x[46] >> 20) & 0x7FF, x[46] & 0xFFFFF, (x[47] >> 20) & 0x7FF, x[47] & 0xFFFFF);
                add("; columns the producer's waves left at their last publish (v2 / v3 strips): %d %d %d %d", x[42], x[43], x[44], x[45]);
                {   // -DG2G_HEARTBEAT builds: (step, place) of up to four waves of a v2 / v3 producer, twice, 50 us apart (all zero otherwise)
                    bool any = false;
                    for (int k = 48; k < 64; ++k) if (x[k]) any = true;
                    if (any) {
                        add("; producer's waves (step:place, then 50 us later):");
                        for (int w = 0; w < 4; ++w) add(" w%d %d:%d -> %d:%d", w, x[48 + 2 * w], x[49 + 2 * w], x[56 + 2 * w], x[57 + 2 * w]);
                    }
                }
                add("; the blocker (lowest strip of this DP without a publish in this generation, %d below the polled one): word %d:%d, HW_ID %08x XCC %08x, taken-by marker %08x (gen %d, workgroup %d), past the left chain %08x, past its first look at the strip above %08x, first wave's last publish %d (markers carry the generation; 7fffffff: never written)",
                    x[64], (x[65] >> 20) & 0x7FF, x[65] & 0xFFFFF, x[66], x[67], x[68], (x[68] >> 20) & 0x7FF, x[68] & 0xFFFF, x[69], x[70], x[71]);
                if (b->dump_off > 0) {
                    // the whole pipeline of the first time-out's DP as that waiter saw it: strips ti-1, ti-2, ... (word, HW_ID, markers, the
                    // two waves' last publish).  A strip is TIGHT when its predecessor is less than 48 columns ahead (it can only be waiting
                    // for it); the HEADS are the unfinished strips that are not tight: they wait for nobody's progress.
                    std::vector<int> dump(2 + G2G_DUMP_WORDS * G2G_DUMP_STRIPS);
                    if (hipMemcpy(dump.data(), b->d_flags + b->dump_off, sizeof(int) * dump.size(), hipMemcpyDeviceToHost) == hipSuccess && dump[0] > 0) {
                        const int nd = std::min(dump[0], (int) G2G_DUMP_STRIPS), ti = dump[1];
                        const int g1 = x[4] & ~0xFFFFF;
                        auto colof = [&](int w) { return w < (g1 | 0) ? -1 : (w & 0xFFFFF); };      // -1: nothing in this generation
                        int heads = 0, unfinished = 0, untaken = 0;
                        add("; pipeline of that DP (%d strips above the waiter dumped): heads", nd);
                        for (int k = 0; k < nd; ++k) {
                            const int *d = dump.data() + 2 + G2G_DUMP_WORDS * k;
                            const int c = colof(d[0]);
                            if (c == 0xFFFFF) continue;
                            ++unfinished;
                            const bool taken = (d[2] & ~0xFFFFF) == (g1 | 0) || ((d[2] >> 20) & 0x7FF) == ((g1 >> 20) & 0x7FF);
                            if (!taken) ++untaken;
                            const int cp = k + 1 < nd ? colof(dump[2 + G2G_DUMP_WORDS * (k + 1)]) : 0xFFFFF;      // predecessor (the top chain counts as finished)
                            if (cp == 0xFFFFF || cp - c >= 48) {
                                if (heads < 6) add(" [strip %d: col %d, predecessor %s%d, HW_ID %08x, taken %08x, past left chain %08x, past first look %08x, waves' last publish %d %d, waves' step:place %d:%d %d:%d]",
                                                   ti - 1 - k, c, cp == 0xFFFFF ? "finished " : "col ", cp == 0xFFFFF ? 0 : cp, d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]);
                                ++heads;
                            }
                        }
                        add(" -- %d head(s), %d unfinished strip(s), %d of them not taken from the queue in this generation", heads, unfinished, untaken);
                        if (const char *fn = g2g_opt(ctx, "STALL_DUMP_FILE")) {
                            if (FILE *fd = fopen(fn, "a")) {
                                fprintf(fd, "# gen %d, waiter strip %d, %d strips: strip word(gen:col) HW_ID taken pastleft pastfirst pub0 pub1 w0step w0place w1step w1place\n", b->gen, ti, nd);
                                for (int k = 0; k < nd; ++k) { const int *d = dump.data() + 2 + G2G_DUMP_WORDS * k; fprintf(fd, "%d %d:%d %08x %08x %08x %08x %d %d %d %d %d %d\n", ti - 1 - k, (d[0] >> 20) & 0x7FF, d[0] & 0xFFFFF, d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]); }
                                fclose(fd);
                            }
                        }
                    }
                }
                add("; chain words of this DP as the first waiter saw them (valid when the polled word is a strip's): left %d:%d, top %d:%d", (x[80] >> 20) & 0x7FF, x[80] & 0xFFFFF, (x[81] >> 20) & 0x7FF, x[81] & 0xFFFFF);
                if (x[72]) add("; of the waves released when their DP was given up, the one with the lowest strip index (%d) was waiting on word %d for %d:%d and had last seen %d:%d (read-modify-write on release: %d:%d) after %d polls, %.0f ms of its own running time",
                               0x7fffffff - x[72], x[75], (x[73] >> 20) & 0x7FF, x[73] & 0xFFFFF, (x[74] >> 20) & 0x7FF, x[74] & 0xFFFFF, (x[78] >> 20) & 0x7FF, x[78] & 0xFFFFF, x[76], x[77] * 65536. / ctx->rt_ticks_per_ms);
                add("; waiting waves off the machine for > 4 ms at a stretch in this run: %d (longest %.1f ms)", x[40], x[41] * 1024. / ctx->rt_ticks_per_ms);
                buf[o] = 0;
                if (!b->is_retry || ctx->last_timeout.empty()) ctx->last_timeout = buf;
                if (!b->injected) { std::lock_guard<std::mutex> lk(g_report_mx); g_last_report = buf; }
                if (g2g_opt(ctx, "DEBUG") || g2g_opt(ctx, "WARN")) { fprintf(stderr, "[g2g] s_memrealtime: %.0f ticks/ms; %s\n", ctx->rt_ticks_per_ms, buf); fflush(stderr); }
            }
            if (b->force_v1 || lost.empty()) { g2g_set_error("%s", "scheduler: a wait timed out and no DP could be singled out"); return G2G_ERR_DEVICE; }
            std::vector<const g2g_problem *> pp;
            for (int i : lost) pp.push_back(b->src[i]);
            g2g_batch *rb = 0;
            // a time-out is a rare accident of scheduling, not a property of the DP: the lost DPs first run again on the ordinary
            // kernels (a small batch: tens of ms); only if THAT run loses a wait too do its DPs go to v1, which polls nothing
            int rc = batch_prepare_impl(ctx, (int) pp.size(), pp.data(), &rb, b->is_retry);
            if (!rc) {
                rb->is_retry = true;
                rc = g2g_batch_run(rb);
                std::vector<g2g_result> rr(pp.size());
                if (!rc) rc = g2g_batch_fetch(rb, rr.data());
                if (!rc) for (size_t k = 0; k < lost.size(); ++k) {
                    free(b->recovered[lost[k]].trace);
                    b->recovered[lost[k]] = rr[k]; b->was_recovered[lost[k]] = 1;
                }
                g2g_batch_free(rb);
            }
            if (rc) { g2g_set_error("%s", "scheduler: the re-run of timed-out DPs failed"); return rc; }
            b->n_recovered += (int) lost.size();
        }
    }
    HIPCHK(hipEventElapsedTime(&b->fwd_ms, ctx->ev[0], ctx->ev[1]));
    HIPCHK(hipEventElapsedTime(&b->tb_ms, ctx->ev[1], ctx->ev[2]));
    return G2G_OK;
}

extern "C" int g2g_batch_fetch(g2g_batch *b, g2g_result *res)
{
    if (!b || !res) return G2G_ERR_ARG;
    g2g_ctx *ctx = b->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<char> all(b->out_hi - b->out_lo);
    if (!all.empty()) HIPCHK(hipMemcpy(all.data(), b->d_arena + b->out_lo, all.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < b->n; ++i) {
        res[i].status = b->status[i];
        res[i].cells = b->cells[i];
        res[i].trace = 0; res[i].ntrace = 0; res[i].score = 0;
        if (b->status[i]) continue;
        if (b->was_recovered[i]) {                      // (re-run after a time-out: a copy, the batch may be fetched again)
            res[i] = b->recovered[i];
            res[i].trace = (g2g_skl *) malloc(sizeof(g2g_skl) * (size_t) (res[i].ntrace > 0 ? res[i].ntrace : 1));
            memcpy(res[i].trace, b->recovered[i].trace, sizeof(g2g_skl) * (size_t) res[i].ntrace);
            continue;
        }
        const char *tmp = all.data() + (b->out_off[i] - b->out_lo);
        memcpy(&res[i].score, tmp, sizeof(double));
        int nt, rr0;
        memcpy(&nt, tmp + sizeof(double), sizeof(int));
        memcpy(&rr0, tmp + sizeof(double) + sizeof(int), sizeof(int));
        res[i].rr[0] = rr0;
        res[i].rr[1] = b->rr1[i];
        if (nt < 2 || nt > b->tcap[i]) { res[i].status = G2G_ERR_DEVICE; continue; }
        res[i].ntrace = nt;
        res[i].trace = (g2g_skl *) malloc(sizeof(g2g_skl) * nt);
        memcpy(res[i].trace, tmp + sizeof(double) + 2 * sizeof(int), sizeof(g2g_skl) * nt);
    }
    return G2G_OK;
}

extern "C" void g2g_batch_times(const g2g_batch *b, float *fwd_ms, float *tb_ms)
{
    if (fwd_ms) *fwd_ms = b ? b->fwd_ms : 0;
    if (tb_ms) *tb_ms = b ? b->tb_ms : 0;
}

extern "C" long long g2g_batch_cells(const g2g_batch *b)
{
    long long c = 0;
    if (b) for (int i = 0; i < b->n; ++i) if (!b->status[i]) c += b->cells[i];
    return c;
}

extern "C" size_t g2g_batch_arena_bytes(const g2g_batch *b) { return b ? b->arena_bytes : 0; }

extern "C" void g2g_batch_recovery(const g2g_batch *b, int *timeouts_last_run, int *recovered_last_run, int *recovered_total)
{
    if (timeouts_last_run) *timeouts_last_run = b ? b->last_timeouts : 0;
    if (recovered_last_run) *recovered_last_run = b ? b->last_recovered : 0;
    if (recovered_total) *recovered_total = b ? b->n_recovered : 0;
}
extern "C" const char *g2g_ctx_last_timeout(const g2g_ctx *c) { return c ? c->last_timeout.c_str() : ""; }
extern "C" void g2g_ctx_wait_gaps(const g2g_ctx *c, long long *count, double *longest_ms)
{
    if (count) *count = c ? c->n_gaps : 0;
    if (longest_ms) *longest_ms = c ? c->max_gap_ms : 0;
}
extern "C" void g2g_process_counters(long long out[8])
{
    if (out) for (int k = 0; k < 8; ++k) out[k] = g_tot[k].load();
}
extern "C" size_t g2g_process_last_timeout(char *buf, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_report_mx);
    if (buf && cap) { const size_t n = std::min(cap - 1, g_last_report.size()); memcpy(buf, g_last_report.data(), n); buf[n] = 0; }
    return g_last_report.size();
}
extern "C" void g2g_ctx_counters(const g2g_ctx *c, long long out[4])
{
    if (!out) return;
    out[0] = c ? c->n_runs : 0; out[1] = c ? c->n_timeouts : 0; out[2] = c ? c->n_recovered : 0; out[3] = c ? c->n_v1 : 0;
}

extern "C" void g2g_batch_free(g2g_batch *b)
{
    if (!b) return;
    hipSetDevice(b->ctx->device);
    release_arena(b);
    pool_give(b->ctx, b->d_tiles, b->tiles_cap);
    pool_give(b->ctx, b->d_flags, b->flags_cap);
    for (int k = 0; k < 24; ++k) pool_give(b->ctx, b->simscr[k], b->simscr_cap[k]);
    for (int k = 0; k < 6; ++k) pool_give(b->ctx, b->twin[k], b->twin_cap[k]);
    for (size_t i = 0; i < b->recovered.size(); ++i) free(b->recovered[i].trace);
    delete b;
}

#ifdef G2G_V6_STAMP
extern "C" void g2g_v6_stamps(unsigned long long *out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g2g_v6_stamp_acc), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g2g_v6_stamp_acc), z, sizeof z); }
}
#endif
#ifdef G2G_V2_STAMP
extern "C" void g2g_stamps(unsigned long long *out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g2g_stamp_acc), sizeof(unsigned long long) * 16); }
extern "C" void g2g_waits(unsigned long long *out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g2g_wait_acc), sizeof(unsigned long long) * 4); }
#endif

// f1: PreSpScore::calcSpScore on the problems of a prepared batch (their inputs are resident in HBM)
// nsets skeletons per problem of the batch in ONE launch (entry e = set e / b->n of problem e % b->n): the windows of g2g_refine
// score the current and the new alignment of every division on the batch the DPs ran on, without packing the problems again
// One set of calcSpScore walks in flight: launched on a stream (spscore_launch: uploads, g2g_spprep_kernel, g2g_spscore_kernel, the
// copies back), collected later (spscore_finish).  The host images the asynchronous copies read live in the object.
struct SpRun {
    int n, nb; char *d; size_t d_cap, o_out, o_st, b_out, b_int; hipStream_t s;
    std::vector<g2g_skl> all; std::vector<int> off, cnt, colpre; std::vector<long long> goff, soff; std::vector<g2g_spparams> sp;
    std::vector<double> ho; std::vector<int> hs;
    hipError_t e;
    SpRun() : n(0), nb(0), d(0), d_cap(0), o_out(0), o_st(0), b_out(0), b_int(0), s(0), e(hipSuccess) {}
};
static int spscore_launch(g2g_batch *b, int nsets, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, hipStream_t stream,
                          void **slots, size_t *slots_cap, SpRun &R)
{
    g2g_ctx *ctx = b->ctx;
    const int nb = b->n;
    const int n = nsets * nb;
    R.n = n; R.nb = nb;
    std::vector<int> &off = R.off, &cnt = R.cnt;
    off.assign(n, 0); cnt.assign(n, 0);
    size_t tot = 0;
    for (int i = 0; i < n; ++i) { off[i] = (int) tot; cnt[i] = (skl[i] && nskl[i] > 0) ? nskl[i] : 0; tot += cnt[i]; }
    std::vector<g2g_skl> &all = R.all;
    all.assign(tot ? tot : 1, g2g_skl());
    for (int i = 0; i < n; ++i) if (cnt[i]) memcpy(&all[off[i]], skl[i], sizeof(g2g_skl) * cnt[i]);
    const size_t b_skl = sizeof(g2g_skl) * all.size(), b_int = sizeof(int) * n, b_sp = sizeof(g2g_spparams) * n, b_out = sizeof(double) * 6 * n;
    char *d = 0;
    R.sp.assign(sp, sp + n);
    const size_t o_skl = 0, o_off = (b_skl + 15) & ~(size_t) 15, o_cnt = o_off + ((b_int + 15) & ~(size_t) 15),
                 o_sp = o_cnt + ((b_int + 15) & ~(size_t) 15), o_out = o_sp + ((b_sp + 15) & ~(size_t) 15),
                 o_st = o_out + ((b_out + 15) & ~(size_t) 15);
    // Gep1st rings (Noll 3): (a.many + b.many) x (codonk1 + 1) ints per problem; naive units: their gap-length arrays; zeroed
    std::vector<long long> &goff = R.goff;
    goff.assign(n, -1);
    size_t gints = 0;
    for (int i = 0; i < n; ++i) {
        const DevProb &dp = b->dp[i % nb];
        const bool rings = dp.kind >= 1 && dp.noll == 3 && dp.codonk1 > 0 && dp.codonk1 < (1 << 20);
        if (rings || dp.kind == 3) {                      // kind 3 (naive units): + gla[an], glb[bn] in front of the rings
            goff[i] = (long long) gints;
            if (dp.kind == 3) gints += (size_t) dp.a.many + dp.b.many + 2;
            if (rings) gints += (size_t) (dp.a.many + dp.b.many) * ((size_t) dp.codonk1 + 1);
        }
    }
    // the streamed walk (kinds 1 and 2: g2g_spprep_kernel lays the position-only inputs of every path column out in path order):
    // colpre = path columns up to and including each skeleton segment, soff = first slot of each alignment (-1: walks unstreamed)
    std::vector<int> &colpre = R.colpre;
    colpre.assign(all.size(), 0);
    std::vector<long long> &soff = R.soff;
    soff.assign(n, -1);
    size_t nslots = 0;
    int maxcols = 0;
    const bool streamed = !g2g_opt(ctx, "NO_SPSTREAM");
    for (int i = 0; i < n && streamed; ++i) {
        const DevProb &dp = b->dp[i % nb];
        if ((dp.kind != 1 && dp.kind != 2) || cnt[i] < 2 || b->status[i % nb]) continue;
        long long cols = 0;
        bool good = true;
        for (int k = 1; k < cnt[i] && good; ++k) {
            const int mi = all[off[i] + k].m - all[off[i] + k - 1].m, ni = all[off[i] + k].n - all[off[i] + k - 1].n;
            if (mi < 0 || ni < 0) good = false;
            cols += std::max(mi, ni);
            if (cols > (1 << 28)) good = false;
            colpre[off[i] + k] = (int) cols;
        }
        if (!good || cols == 0) continue;
        soff[i] = (long long) nslots; nslots += (size_t) cols; maxcols = std::max(maxcols, (int) cols);
    }
    if (nslots * sizeof(SpSlot) > ((size_t) 4 << 30)) { std::fill(soff.begin(), soff.end(), -1LL); nslots = 0; }
    if (nslots * sizeof(SpSlot) > (*slots_cap)) {
        if ((*slots)) hipFree((*slots));
        (*slots) = 0; (*slots_cap) = 0;
        const size_t want = nslots * sizeof(SpSlot) + (nslots * sizeof(SpSlot)) / 4;
        if (hipMalloc(&(*slots), want) == hipSuccess) (*slots_cap) = want;
        else { (void) hipGetLastError(); std::fill(soff.begin(), soff.end(), -1LL); nslots = 0; }       // no room: the walks read in place
    }
    const size_t b_goff = sizeof(long long) * n, o_goff = (o_st + b_int + 15) & ~(size_t) 15,
                 o_gws = (o_goff + b_goff + 15) & ~(size_t) 15, o_cpre = (o_gws + sizeof(int) * (gints ? gints : 1) + 15) & ~(size_t) 15,
                 o_soff = (o_cpre + sizeof(int) * colpre.size() + 15) & ~(size_t) 15, total = o_soff + b_goff;
    size_t d_cap = 0;
    d = (char *) pool_take(ctx, total, &d_cap);
    R.d = d; R.d_cap = d_cap; R.o_out = o_out; R.o_st = o_st; R.b_out = b_out; R.b_int = b_int; R.s = stream;
    if (!d) { g2g_set_error("%s", "spscore: out of device memory"); return G2G_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(d + o_skl, all.data(), b_skl, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_off, off.data(), b_int, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_cnt, cnt.data(), b_int, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_sp, R.sp.data(), b_sp, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_goff, goff.data(), b_goff, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && gints) e = hipMemsetAsync(d + o_gws, 0, sizeof(int) * gints, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_cpre, colpre.data(), sizeof(int) * colpre.size(), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_soff, soff.data(), b_goff, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && nslots) {
        hipLaunchKernelGGL(g2g_spprep_kernel, dim3((unsigned) std::min(64, (maxcols + 255) / 256), (unsigned) n), dim3(256), 0, stream,
                           (const DevProb *) b->d_probs, n, nb, (const int2 *) (d + o_skl), (const int *) (d + o_off), (const int *) (d + o_cnt),
                           (const int *) (d + o_cpre), (const long long *) (d + o_soff), (SpSlot *) (*slots));
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(g2g_spscore_kernel, dim3(n), dim3(64), 0, stream, (const DevProb *) b->d_probs, n, nb,
                           (const SpParamsDev *) (d + o_sp), (const int2 *) (d + o_skl), (const int *) (d + o_off),
                           (const int *) (d + o_cnt), (double *) (d + o_out), (int *) (d + o_st),
                           gints ? (int *) (d + o_gws) : (int *) 0, (const long long *) (d + o_goff),
                           (const int *) (d + o_cpre), (const long long *) (d + o_soff), nslots ? (const SpSlot *) (*slots) : (const SpSlot *) 0,
                           g2g_opt(ctx, "NO_SPLANES") ? 1 : 0);
        e = hipGetLastError();
    }
    R.ho.assign(6 * (size_t) n, 0.);
    R.hs.assign(n, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(R.ho.data(), d + o_out, b_out, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(R.hs.data(), d + o_st, b_int, hipMemcpyDeviceToHost, stream);
    R.e = e;
    return G2G_OK;
}
static int spscore_finish(g2g_batch *b, SpRun &R, g2g_fstat *out)
{
    g2g_ctx *ctx = b->ctx;
    hipError_t e = R.e;
    if (e == hipSuccess) e = hipStreamSynchronize(R.s);
    else (void) hipStreamSynchronize(R.s);
    pool_give(ctx, R.d, R.d_cap);
    R.d = 0;
    if (e != hipSuccess) { g2g_set_error("spscore: %s", hipGetErrorString(e)); (void) hipGetLastError(); return G2G_ERR_DEVICE; }
    const int n = R.n, nb = R.nb;
    for (int i = 0; i < n && out; ++i) {
        out[i].val = R.ho[6 * i]; out[i].gap = R.ho[6 * i + 1]; out[i].raw = R.ho[6 * i + 2]; out[i].reserved = 0;
        out[i].mch = R.ho[6 * i + 3]; out[i].mmc = R.ho[6 * i + 4]; out[i].unp = R.ho[6 * i + 5];
        out[i].status = b->status[i % nb] ? b->status[i % nb] : R.hs[i] == 0 ? G2G_OK : R.hs[i] == -2 ? G2G_ERR_MODE : G2G_ERR_ARG;
    }
    return G2G_OK;
}
static bool spscore_lists_in_state(const g2g_batch *b)
{   // the walkers keep their running lists in LDS unless the problem's capacities exceed SP_FAST_LIST entries -- then they use the problem's
    // own state arrays: two walks of one problem (and a walk beside the DP itself) must then go one after the other
    for (int i = 0; i < b->n; ++i) if (b->dp[i].capa + 1 > SP_FAST_LIST || b->dp[i].capb + 1 > SP_FAST_LIST) return true;
    return false;
}
extern "C" int g2g_batch_spscore_sets(g2g_batch *b, int nsets, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, g2g_fstat *out)
{
    if (!b || nsets < 1 || !sp || !skl || !nskl || !out) return G2G_ERR_ARG;
    g2g_ctx *ctx = b->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int nb = b->n;
    if (nb == 0) return G2G_OK;
    if (nsets > 1 && spscore_lists_in_state(b)) {
        for (int k = 0; k < nsets; ++k) {
            const int rc = g2g_batch_spscore_sets(b, 1, sp + (size_t) k * nb, skl + (size_t) k * nb, nskl + (size_t) k * nb, out + (size_t) k * nb);
            if (rc) return rc;
        }
        return G2G_OK;
    }
    SpRun R;
    const int rc = spscore_launch(b, nsets, sp, skl, nskl, ctx->stream, &ctx->sp_slots, &ctx->sp_slots_cap, R);
    if (rc) return rc;
    return spscore_finish(b, R, out);
}
// One set of walks BESIDE the DP kernels of the batch (g2g_align2_score_batch: the current alignments' calcSpScore needs only the inputs,
// which are resident once the batch is prepared): launched on a stream of its own before g2g_batch_run, collected after it.  *handle
// stays NULL when the walks do not run beside the DPs (the default, see below; or their lists would live in the problems' state arrays):
// the caller then scores both sets after the run, as before.
extern "C" int g2g_batch_spscore_begin(g2g_batch *b, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, void **handle)
{
    if (!b || !sp || !skl || !nskl || !handle) return G2G_ERR_ARG;
    *handle = 0;
    g2g_ctx *ctx = b->ctx;
    // OFF by default: measured on the 256 x 1024 refinement the window got SLOWER with the walks beside the DPs (113.3 against 105.6 ms:
    // 16 latency-bound pipelines do not tolerate neighbours on their CUs, and the packing of the walks delays the DP launches) -- the
    // 3 ms of walks it hides cost 8.  Option SP_OVERLAP turns it on (tests keep the path alive).
    if (b->n == 0 || spscore_lists_in_state(b) || !g2g_opt(ctx, "SP_OVERLAP")) return G2G_OK;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->sp_stream) {
        if (hipStreamCreateWithFlags(&ctx->sp_stream, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); ctx->sp_stream = 0; return G2G_OK; }
        if (hipEventCreateWithFlags(&ctx->sp_ev, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); hipStreamDestroy(ctx->sp_stream); ctx->sp_stream = 0; return G2G_OK; }
    }
    HIPCHK(hipEventRecord(ctx->sp_ev, ctx->stream));                 // (the batch's inputs were uploaded on the main stream)
    HIPCHK(hipStreamWaitEvent(ctx->sp_stream, ctx->sp_ev, 0));
    SpRun *R = new SpRun();
    const int rc = spscore_launch(b, 1, sp, skl, nskl, ctx->sp_stream, &ctx->sp_slots2, &ctx->sp_slots2_cap, *R);
    if (rc) { delete R; return rc; }
    *handle = R;
    return G2G_OK;
}
extern "C" int g2g_batch_spscore_end(g2g_batch *b, void *handle, g2g_fstat *out)       // out NULL: wait and drop
{
    if (!b || !handle) return G2G_ERR_ARG;
    SpRun *R = (SpRun *) handle;
    const int rc = spscore_finish(b, *R, out);
    delete R;
    return rc;
}

extern "C" int g2g_batch_spscore(g2g_batch *b, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl, g2g_fstat *out)
{
    return g2g_batch_spscore_sets(b, 1, sp, skl, nskl, out);
}

// upper estimate of the arena bytes one problem takes in a batch (inputs + state + trace + column scores)
static size_t problem_bytes(const g2g_ctx *ctx, const g2g_problem *p)
{
    if (check_problem(p)) return 4096;
    const int al = p->a.left, ar = p->a.right, bl = p->b.left, br = p->b.right;
    long long cells = 0;
    int tmax = 1;
    for (int dd = al + bl; dd <= (ar - 1) + (br - 1); ++dd) {
        int mlo, mhi;
        diag_rows(dd, al, ar, bl, br, is_rect(p->alnmode) ? bl - ar : p->lw, is_rect(p->alnmode) ? br - al : p->up, &mlo, &mhi);
        const int c = mhi - mlo + 1;
        if (c > 0) cells += c;
        if (c > tmax) tmax = c;
    }
    // trace (1 B per cell) + the column-score matrix (8 B per cell) where a kernel reads one: strips in sweep mode (the
    // default of every tiled kernel) make their own scores
    const bool matrix = g2g_opt(ctx, "NO_SIMBLK") || g2g_opt(ctx, "V2_SWEEP") || g2g_opt(ctx, "V3_SWEEP");   // (tile-mode test configurations)
    size_t bytes = (size_t) (ar - al + br - bl + 2) * tmax + (matrix ? 8 * (size_t) cells : 0);
    const g2g_side *sd[2] = {&p->a, &p->b};
    for (int k = 0; k < 2; ++k) {
        const size_t cols = (size_t) sd[k]->len + 2;
        bytes += cols * sd[k]->many * (kind_of(p->alnmode) == 3 ? 17 : 1) + cols * 24 + cols * 8 * (size_t) ((sd[k]->pseq && sd[k]->nelm > 0) ? sd[k]->nelm : 0);    // (nelm of a side without vectors is not defined)
        if (sd[k]->has_gfq) for (int v = 0; v < 3; ++v) bytes += 12 * (size_t) sd[k]->gfq.off[v][sd[k]->len + 1] + 4 * cols;
    }
    const size_t lists = 8 * (size_t) ((p->a.has_gfq ? p->a.gfq.hetero + 1 : p->a.many) + (p->b.has_gfq ? p->b.gfq.hetero + 1 : p->b.many));
    bytes += (size_t) ((is_rect(p->alnmode) ? (br - al) - (bl - ar) : p->up - p->lw) + 3) * 7 * (16 + lists);      // v1 state rows
    bytes += (size_t) (p->b.len + 3 + ar - al + 3) * 9 * (16 + lists / 2 + 16);                 // strip / block boundary records
    bytes += 8 * (size_t) (ar - al + 2) + 8 * (size_t) (ar - al + br - bl + 8) + 4096;
    return bytes;
}

// alignC<recd_t> for n problems.  Batches are cut so that one arena stays below a budget (default: 70 % of the free
// device memory, capped by G2G_ARENA_LIMIT_GB): a sweep over thousands of large divisions (BASELINE configs[4]:
// 4093 divisions of a 2048 x 4096 nt family) does not fit 288 GB at once.
extern "C" int g2g_forward_batch(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_result *res)
{
    if (!ctx || n < 0 || (n && (!prob || !res))) return G2G_ERR_ARG;
    if (!ctx->ok) return G2G_ERR_NODEVICE;
    HIPCHK(hipSetDevice(ctx->device));
    size_t budget = (size_t) 64 << 30;              // per chunk (128 GB chunks were no faster on a 3.3e10-cell DNA sweep)
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr) budget = std::min(budget, (size_t) (0.7 * (double) fr)); }
    if (const char *e = g2g_opt(ctx, "ARENA_LIMIT_GB")) { const double g = atof(e); if (g > 0) budget = (size_t) (g * (double) ((size_t) 1 << 30)); }
    for (int i = 0; i < n; ++i) { res[i].trace = 0; res[i].ntrace = 0; }
    int lo = 0;
    while (lo < n) {
        size_t acc = 0;
        int hi = lo;
        while (hi < n) {
            const size_t pb = problem_bytes(ctx, prob[hi]);
            if (hi > lo && (acc + pb > budget || hi - lo >= G2G_MAX_BATCH)) break;
            acc += pb; ++hi;
        }
        if (g2g_opt(ctx, "DEBUG")) { fprintf(stderr, "[g2g] forward_batch: chunk [%d, %d) of %d, %.3g of %.3g bytes\n", lo, hi, n, (double) acc, (double) budget); fflush(stderr); }
        g2g_batch *b = 0;
        int rc = g2g_batch_prepare(ctx, hi - lo, prob + lo, &b);
        if (!rc) {
            rc = g2g_batch_run(b);
            if (!rc) rc = g2g_batch_fetch(b, res + lo);
            g2g_batch_free(b);
        }
        if (rc) {             // the call fails as a whole: nothing of the finished chunks is handed out
            for (int i = 0; i < hi; ++i) { free(res[i].trace); res[i].trace = 0; res[i].ntrace = 0; }
            return rc;
        }
        lo = hi;
    }
    return G2G_OK;
}

#include "g2g_dist.hip"                // f3: the guide-tree DPs (own kernels, own entry point; shares the context)
#include "g2g_pairaln.hip"             // f3: alignB_ng (pairwise alignment of single sequences with the path)
#include "g2g_pairsum.hip"             // f1: Ssrel::pairsum_ss (naive nodes on the GPU, joins through calcSpScore)
#include "g2g_build.hip"               // a8 / a9: thickness, vectors and gap profiles of a batch of groups on the device
