/*
 * g2g.h -- C ABI of the MI355X-native group-to-group alignment DP (libg2g.so).
 *
 * Drop-in boundary for ONE hot path of ogotoh/prrn_aln: the group-to-group affine / double-affine gap
 * DP that align2() dispatches to (reference src/maln2.cc:1875 -> alignC<recd_t> src/fwd2c.h:671 ->
 * Fwd2c<recd_t>::forwardB src/fwd2c.h:359 + Vmf::traceback src/vmf.cc:105 + stdskl src/gaps.cc:139).
 * Plain pointers and sizes only; no C++/torch types.  Every entry point names the reference interface
 * it replaces.  All scores are IEEE-754 doubles (the prrn build: -DDVAL=1, reference src/cmn.h:40-50).
 *
 * Two levels:
 *   level 1 ("operator")  g2g_group / g2g_pwdm / g2g_align2      <-> mSeq / PwdM / align2()
 *   level 0 ("engine")    g2g_problem / g2g_forward_batch        <-> Fwd2c<recd_t>(seqs,pwd).forwardB()+traceback()
 * Level 1 builds a g2g_problem on the host (mode selection, thickness, profile vectors, gap profiles:
 * SURVEY.md §8 rows a7-a9) and hands it to level 0, which runs on the GPU.  There is no CPU fallback:
 * if no HIP device is usable every compute entry point returns G2G_ERR_NODEVICE.
 */
#ifndef G2G_H_
#define G2G_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2G_ABI_VERSION 5

/* error codes (negative) */
enum {
    G2G_OK              =  0,
    G2G_ERR_ARG         = -1,   /* malformed argument                                            */
    G2G_ERR_MODE        = -2,   /* alignment mode not on this path (spliced _ALS/_ALH, ether u0>0, SSHP):
                                   the reference itself would fatal() (maln2.cc:1918) or use Fwd2h/Fwd2s */
    G2G_ERR_NODEVICE    = -3,   /* no usable HIP device / kernel image                           */
    G2G_ERR_DEVICE      = -4,   /* HIP runtime error (message via g2g_last_error)                */
    G2G_ERR_NOMEM       = -5,
    G2G_ERR_ENDS        = -6    /* skeleton ends mismatch even at sh = -100 (maln2.cc:1946-1952) */
};

/* reference enum ALN_MODE, src/aln.h:71-76 (banded "_ALB" = default because algmode.bnd = 1) */
enum {
    G2G_NGP_ALN = 1, G2G_HLF_ALN = 2, G2G_RHF_ALN = 3, G2G_GPF_ALN = 4, G2G_NTV_ALN = 5,
    G2G_NGP_ALB = 6, G2G_HLF_ALB = 7, G2G_RHF_ALB = 8, G2G_GPF_ALB = 9, G2G_NTV_ALB = 10
};

/* which PwdM::sim?? column scorer (reference src/maln.h:159-173,205-209; table maln2.cc:347-399).
   value = 10*kind + (weighted ? 1 : 0) for the member-loop scorers */
enum {
    G2G_SIM00 = 0, G2G_SIM11 = 11, G2G_SIM12I = 120, G2G_SIM12W = 121, G2G_SIM13 = 13,
    G2G_SIM21I = 210, G2G_SIM21W = 211, G2G_SIM22I = 220, G2G_SIM22W = 221,
    G2G_SIM23I = 230, G2G_SIM23W = 231, G2G_SIM31 = 31, G2G_SIM32I = 320, G2G_SIM32W = 321,
    G2G_SIM33 = 33, G2G_SIM33N = 330
};

/* reference enum TraceBackDir, src/aln.h:47-52 (only the values Fwd2c produces) */
enum { G2G_DEAD = 0, G2G_DIAG = 2, G2G_NEWD = 3, G2G_VERT = 4, G2G_HORI = 8, G2G_NEWV = 12, G2G_NEWH = 13 };

/* reference struct SKL {int m, n;}, src/cmn.h:124 */
typedef struct g2g_skl { int32_t m, n; } g2g_skl;

/* ---- level 0: the flattened DP problem -------------------------------------------------------- */

/* Static gap profile of one group: the three views Gfq::sfrq/tfrq/rfrq (reference src/gfreq.h:44-64,
   src/gfreq.cc:247-312).  View v (0 = s, 1 = t, 2 = r): the list of position p (p = -1 .. len-1) is
   glen[v][off[v][p+1] ...], ascending in glen and terminated by an entry with glen = -1 (the
   terminator is part of the pool).  off[v] has len+2 entries (the last one = pool length).
   freq of the s view is in accumulated (suffix-sum) form, as the reference stores it. */
typedef struct g2g_gapprof {
    int32_t        hetero;          /* Gfq::hetero: capacity of a dynamic gap-state list is hetero+1 */
    const int32_t *off[3];
    const int32_t *glen[3];
    const double  *freq[3];
} g2g_gapprof;

/* One side of the DP as mSeqItr hands it out (reference src/mseq.h:206-350, src/mseq.cc:760-815). */
typedef struct g2g_side {
    int32_t        many;            /* members                                                    */
    int32_t        len;             /* columns                                                    */
    int32_t        left, right;     /* range to align, reference Seq::left/right                  */
    int32_t        nils;            /* inex.nils                                                  */
    int32_t        dels;            /* inex.dels (any internal/terminal gap)                      */
    const uint8_t *seq;             /* (len+2)*many residue codes, position -1 first ([pos][member],
                                       reference Seq::at(), nil_code=0 gap_code=1 seq.h:76-77)    */
    const double  *weight;          /* many, or NULL                                              */
    int32_t        nelm, felm;      /* profile vector geometry (mseq.h:53-59); 0 if not vectorised */
    const double  *pseq;            /* (len+2)*nelm, position -1 first, or NULL                   */
    const double  *thk;             /* (len+2)*3 {cfq,dfq,efq} at positions -1..len: SeqThk as the
                                       iterator yields it for each position (thk_mode folded in)  */
    int32_t        has_gfq;
    g2g_gapprof    gfq;             /* valid iff has_gfq                                          */
    /* NTV modes only: per position/member gap densities (mSeq::gapdensity/postgapdensity,
       mseq.h:148-160), (len+2)*many each, position -1 first; NULL otherwise */
    const double  *gapdens;
    const double  *postgapdens;
    /* exon-boundary (intron position) annotation of the group, reference SigII::pfq (src/gsinfo.h:33,41-46): npfq entries
       ascending in pos (nucleotide coordinate: column x pfq_step + phase), dns = weighted number of members with an intron
       there; pfq_step = SigII::step (3 for protein columns, 1 for nucleotides).  npfq = 0 / NULL: not annotated.          */
    int32_t        npfq, pfq_step;
    const int32_t *pfq_pos;
    const double  *pfq_dns;
    double         sumwt;           /* Seq::sumwt: sum of the members' weights (the member count when unweighted); read by the
                                       PwdM::stt?? statistics (van / vbn and the thickness of the zero iterators)          */
    /* ABI 5: device-resident twins of the arrays above (same contents, in the HBM of the context that built them:
       g2g_pwdm_create_batch builds a group's derived arrays on the device and leaves them there).  NULL for a side that lives on
       the host only.  A batch prepared from such a side reads the twins where they lie -- nothing is packed or uploaded for them --
       so the objects that own them (the g2g_group behind the g2g_pwdm) must outlive the batch, like the host arrays. */
    const struct g2g_side_dev *dev;
} g2g_side;
typedef struct g2g_side_dev {
    const struct g2g_ctx *ctx;      /* the twins are addresses in THIS context's device                                      */
    const uint8_t *seq;
    const double  *weight, *pseq, *thk;
    const int32_t *off[3], *glen[3];
    const double  *freq[3];
} g2g_side_dev;

typedef struct g2g_problem {
    int32_t  alnmode;               /* G2G_*_ALB / _ALN                                            */
    int32_t  sim2_kind;             /* G2G_SIM*                                                   */
    int32_t  noll;                  /* PwdB::Noll: 2 affine, 3 double affine (aln2.cc:100)        */
    int32_t  codonk1;               /* PwdB::codonk1 (aln2.cc:116-117)                            */
    int32_t  lw, up;                /* band: WINDOW from stripe() (aln2.cc:156-174)               */
    int32_t  crg2_kind;             /* NTV engines: which PwdM::crg?? (maln2.cc:881-1024,1454-1614):
                                       11, 120/121 (12i/12w), 210/211, 220/221; 0 otherwise         */
    int32_t  dvsp;                  /* PwdB::DvsP: 0 nucleotide x nucleotide (the statistics read compacted codes), 3 protein x protein */
    double   basic_gop;             /* PwdM::Basic_GOP  = -scale*v (maln2.cc:232)                 */
    double   weighted_gop;          /* PwdM::Weighted_GOP = -v     (maln2.cc:237)                 */
    double   u;                     /* alnprm.u, unpaired-column penalty in unp1 (maln.h:185)     */
    double   u2divu1, v2divv1;      /* LongGEP/BasicGEP, LongGOP/BasicGOP (fwd2c.h:85-86)         */
    const double *simmtx;           /* rows x dim substitution matrix, Simmtx::mtx (simmtx.h:49)  */
    int32_t  simdim, simrows;
    g2g_side a, b;
    double   spb_fact;              /* SpbFact = alprm.scale * alprm2.spb (src/gsinfo.cc:35): weight of the intron-position bonus
                                       PfqItr::match_score adds in forwardB (src/fwd2c.h:378-379,446-452,472) when BOTH sides are
                                       annotated; 0 switches it off.  DPs with the bonus run on g2g_forward_kernel.            */
} g2g_problem;

/* result of one DP */
typedef struct g2g_result {
    double    score;                /* Fwd2c::forwardB return value (= *scr of align2)            */
    int64_t   cells;                /* in-band cells visited: sum_m (n9 - n), fwd2c.h:373-374,393 */
    int32_t   ntrace;               /* records in `trace`                                         */
    int32_t   status;               /* G2G_OK or error for this item                              */
    g2g_skl  *trace;                /* Vmf::traceback(-1) order: end corner first, origin last
                                       (vmf.cc:105-120); free with g2g_free()                     */
    int64_t   rr[2];                /* what forwardB(pp) reports without a Vmf (HomScoreC, fwd2c.h:664-668,
                                       468-469,477-480): diagonal n-m on which the path leaves the first row,
                                       and b.left - a.left + b.right - a.right                      */
} g2g_result;

typedef struct g2g_ctx g2g_ctx;

/* Create a context bound to one HIP device (device < 0: current device).  Returns NULL on failure.
   Threading: a context (its streams, its pinned staging buffer and its pool of device memory) serves ONE host thread at a
   time -- callers with several threads serialise their calls or batch them (integration/g2g_bind.cc does the latter);
   the group / PwdM builders (g2g_group_create, g2g_pwdm_create) touch only their own objects and may run on any
   number of threads.  g2g_last_error() is per thread.  One process per GPU: the persistent kernels poll flags
   written by other resident workgroups of the same launch and must not compete with another process for the device
   (every wait is bounded; an oversubscribed device shows up as a failed batch, not a hang).                       */
g2g_ctx *g2g_create(int device);
void     g2g_destroy(g2g_ctx *ctx);
const char *g2g_last_error(void);
int      g2g_abi_version(void);
/* 1 if a HIP device + gfx950 code object are usable, else 0 (never falls back to the host). */
int      g2g_device_ok(g2g_ctx *ctx);
/* Tuning and diagnostic switches of ONE context (kernel-variant selection, strip widths, the waits' time limit, arena
   limits, debug output: DESIGN.md section 4 lists the names).  `name` is given with or without the "G2G_" prefix.
   A switch the context has not set reads the environment variable G2G_<NAME> (the environment only supplies
   defaults); value == NULL turns a switch off for this context whatever the environment says.  Takes effect at the
   next g2g_batch_prepare / g2g_forward_batch.  g2g_get_option returns the effective value or NULL (valid until the
   next g2g_set_option on the context). */
int         g2g_set_option(g2g_ctx *ctx, const char *name, const char *value);
const char *g2g_get_option(const g2g_ctx *ctx, const char *name);
void        g2g_reset_options(g2g_ctx *ctx);     /* forget every g2g_set_option: back to the environment's defaults */

/* Fwd2c<recd_t>(seqs, pwd, trb=true).forwardB() + traceback() for a batch of independent problems
   (alignC<recd_t>, reference src/fwd2c.h:671-677).  res[i].trace is malloc'ed by the library. */
int      g2g_forward_batch(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_result *res);

/* The same in three steps, for callers that keep a sweep resident in HBM (bench.py times g2g_batch_run
   only: inputs are already in device memory when it starts).  g2g_batch_times() reports the HIP-event
   durations of the forward and backtrack kernels of the last run, measured on the context's stream.
   The problem descriptions (and every array they point to) must stay valid until g2g_batch_free: a DP that loses
   a scheduler wait is packed again from them and re-run inside g2g_batch_run. */
typedef struct g2g_batch g2g_batch;
int       g2g_batch_prepare(g2g_ctx *ctx, int n, const g2g_problem *const *prob, g2g_batch **out);
int       g2g_batch_run(g2g_batch *b);
int       g2g_batch_fetch(g2g_batch *b, g2g_result *res);
void      g2g_batch_times(const g2g_batch *b, float *fwd_ms, float *tb_ms);
long long g2g_batch_cells(const g2g_batch *b);
size_t    g2g_batch_arena_bytes(const g2g_batch *b);
/* The scheduler's waits are bounded by wall clock: a wait that runs into the limit costs its DP a re-run
   inside g2g_batch_run, and the results stay complete.  These two make every such event visible: for the LAST run of a
   batch the number of waits that gave up and of DPs re-run (and the DPs re-run over the batch's life); for a context
   out[0] = batch runs, out[1] = waits that gave up, out[2] = DPs re-run, out[3] = of those, DPs that needed the
   non-polling kernel -- counted over every entry point that runs DPs (g2g_align2_batch, g2g_refine, ...).  An ordinary
   run reports zeros. */
void      g2g_batch_recovery(const g2g_batch *b, int *timeouts_last_run, int *recovered_last_run, int *recovered_total);
void      g2g_ctx_counters(const g2g_ctx *ctx, long long out[4]);
/* the report of the context's last recovered time-out (which DPs on which kernel, what the first waiting wave saw and where
   its producer ran): "" if there was none; valid until the next event or g2g_destroy */
const char *g2g_ctx_last_timeout(const g2g_ctx *ctx);
/* The waits count the time the waiting wave itself was running: a wave that finds more than 4 ms between two looks at the clock
   was off the machine (with, as a rule, the rest of its kernel) and counts 4 ms of it.  count = such gaps seen by waiting
   waves since g2g_create, longest_ms = the longest: zeros on an undisturbed device. */
void      g2g_ctx_wait_gaps(const g2g_ctx *ctx, long long *count, double *longest_ms);
/* The same over the whole PROCESS (every context, also destroyed ones): out[0] = batch runs, [1] = waits that gave up, [2] = DPs
   re-run, [3] = of those on the non-polling kernel, [4] = gaps seen by waiting waves, [5] / [6] = the part of [1] / [2] that
   came from batches prepared with the INJECT_STALL test hook, [7] reserved.  A host (or a test session: tests/conftest.py)
   that wants a recovery to be an ERROR rather than a counter compares [2] with [6] at its end.  g2g_process_last_timeout copies
   the report of the last event that was NOT injected into buf (NUL-terminated, truncated to cap) and returns its full length. */
void      g2g_process_counters(long long out[8]);
size_t    g2g_process_last_timeout(char *buf, size_t cap);
/* Device memory is owned by the context (a pool of blocks that batches take from and give back to): out[0] = hipMalloc calls
   made for it since g2g_create, [1] = hipFree calls, [2] = requests served from the pool, [3] = bytes free in the pool now.
   A steady-state loop (a refinement: one batch per window) shows [0] standing still. */
void      g2g_ctx_mem_counters(const g2g_ctx *ctx, long long out[4]);
void      g2g_batch_free(g2g_batch *b);

/* stdskl(): sort + normalise a raw traceback into ascending unique corners (reference src/gaps.cc:139).
   in[0..n) raw records; returns malloc'ed corners (count in *nout), caller g2g_free()s. */
g2g_skl *g2g_stdskl(const g2g_skl *in, int n, int *nout);

void     g2g_free(void *p);

/* ---- level 1: the operator surface ------------------------------------------------------------- */

/* subset of reference struct ALPRM (src/seq.h:27-28) + the algmode bits this path reads */
typedef struct g2g_params {
    double  u, v, u0, u1, tgapf, scale, gamma;
    int32_t k1, ls, sh;
    int32_t banded;                 /* algmode.bnd                                                */
    int32_t molc;                   /* 1 PROTEIN, 2 DNA (cmn.h:107)                               */
    const double *simmtx;           /* rows x dim                                                  */
    int32_t simdim, simrows;
    int32_t max_code;               /* SEQ_CODE::max_code of the alphabet                         */
} g2g_params;

typedef struct g2g_group g2g_group;   /* <-> mSeq  */
typedef struct g2g_pwdm  g2g_pwdm;    /* <-> PwdM  */

/* <-> aggregate()/mSeq construction (src/mgaps.cc:282) followed by exg_seq(0,0): `seq` is
   (len)*many residue codes [pos][member] for positions 0..len-1; the library adds the sentinel
   columns.  weight may be NULL. */
g2g_group *g2g_group_create(g2g_ctx *ctx, const g2g_params *prm, int many, int len,
                            const uint8_t *seq, const double *weight);
void       g2g_group_free(g2g_group *g);

/* <-> PwdM::PwdM(mSeq** seqs, const ALPRM*) (src/maln2.cc:254): selects alnmode, swaps so the
   profile side is `a`, builds thickness / vectors / gap profiles.  *swapped receives PwdM::swp. */
g2g_pwdm  *g2g_pwdm_create(g2g_ctx *ctx, const g2g_params *prm, g2g_group *a, g2g_group *b, int *swapped);
/* n PwdMs at once with the derived arrays of all their groups -- column thickness (mSeq::mkthick, src/mseq.cc:149-354), frequency /
   profile vectors (mSeq::convseq, src/mseq.cc:392-587) and static gap profiles (Gfq::Gfq / seq2gfq, src/gfreq.cc:134-312) -- built
   ON THE DEVICE in one go (csrc/g2g_build.hip): what a sweep over hundreds of divisions wants.  Same objects and, bit for bit, the
   same arrays as n calls of g2g_pwdm_create (tests/test_gpu_builders.py).  Groups with nil codes (tgapf < 1) are built on the
   host inside the same call.  swapped (n ints) may be NULL.  On failure nothing is handed out. */
int        g2g_pwdm_create_batch(g2g_ctx *ctx, const g2g_params *prm, int n, g2g_group *const *a, g2g_group *const *b, int *swapped, g2g_pwdm **out);
void       g2g_pwdm_free(g2g_pwdm *p);
const g2g_problem *g2g_pwdm_problem(const g2g_pwdm *p);

/* <-> SKL* align2(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo*) (src/maln2.cc:1875): forward fill,
   traceback, stdskl, end check with the sh = -100 retry.  *skl is malloc'ed: skl[0..*nskl) corners
   ascending (the reference's skl[1..n]); caller g2g_free()s.  Batched form = one randiv sweep. */
/* ---- f1: the sum-of-pairs score of the alignment a skeleton describes ---------------------------------
 * <-> VTYPE PreSpScore::calcSpScore(Gsinfo*) (src/fspscore.cc:584-622) = SpScore<SPunit|SPunit_hf|SPunit_pf>::calcSkl
 * (src/fspscore.h:202-254, calscr src/fspscore.cc:346-541) followed by PwdM::rescale (src/maln2.cc:245-252): the score is
 * re-evaluated ALONG the path (column scores, unpaired-column penalties and the gap-open counts of the gap-profile
 * algebra) and returned per unit pair weight; Prrn::onecycle takes fstat.val of the old and the new alignment as the
 * acceptance delta.  On this path: NGP / HLF / RHF / GPF modes and the naive units SPunit_nv / _w11 / _w21 / _w22 of the
 * NTV modes, with Noll 2 and 3 (-yl3: the long-gap bookkeeping `Gep1st`, src/mseq.cc:658-758, included).  mch/mmc/unp of
 * FSTAT (PwdM::stt2) are not computed.                                                                             */
typedef struct {
    double vab;          /* PwdM::Vab = scale * wa * wb (src/maln2.cc:234)                         */
    double basic_gep;    /* PwdB::BasicGEP = -u * axbscale (src/aln2.cc:103)                       */
    double diffu;        /* PwdB::diffu = LongGEP - BasicGEP (src/aln2.cc:105)                      */
    double diff_u;       /* PwdM::diff_u = scale * (u - u1) (src/maln2.cc:233): weight of the long-gap count `lunp` in
                            PwdM::wgop (maln.h:321-325); only read when Noll = 3                                */
    int32_t flags;       /* G2G_SP_NOSTATS: leave FSTAT's mch / mmc / unp at 0 (the statistics cost about a third of the walk;
                            a refinement loop reads val and raw only)                                               */
    int32_t reserved;
} g2g_spparams;
#define G2G_SP_NOSTATS 1
typedef struct {
    double  val, gap;    /* FSTAT::val / gap after PwdM::rescale: per unit pair weight (what align2 leaves in Gsinfo.fstat)      */
    int32_t status, reserved;
    double  raw;         /* the score BEFORE rescale = the return value of PreSpScore::calcSpScore(SKL*), src/fspscore.cc:544-582:
                            Prrn::onecycle takes THIS for the current alignment and fstat.val for the new one (src/prrn5.cc:523,535) */
    double  mch, mmc, unp;   /* FSTAT::mch / mmc / unp after rescale: matched, mismatched and unpaired member pairs per unit pair
                                weight (PwdM::stt?? src/maln2.cc:627-850,1300-1450; the naive units count in calcstat)        */
} g2g_fstat;
/* level 0: on a prepared batch (inputs resident in HBM); skl[i] = the standardised skeleton of problem i
 * (g2g_stdskl output: corners ascending, first = (a.left, b.left), last = (a.right, b.right))              */
int        g2g_batch_spscore(g2g_batch *b, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl,
                             g2g_fstat *out);
/* ... nsets skeletons per problem of the batch in one launch: entry e (of nsets * n) is set e / n of problem e % n */
int        g2g_batch_spscore_sets(g2g_batch *b, int nsets, const g2g_spparams *sp, const g2g_skl *const *skl, const int *nskl,
                                  g2g_fstat *out);
/* level 1 */
int        g2g_pwdm_spparams(const g2g_pwdm *p, g2g_spparams *sp);
int        g2g_spscore_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *p, const g2g_skl *const *skl, const int *nskl,
                             g2g_fstat *out);
int        g2g_spscore_batch_flags(g2g_ctx *ctx, int n, g2g_pwdm *const *p, const g2g_skl *const *skl, const int *nskl,
                                   int flags, g2g_fstat *out);       /* the same with g2g_spparams::flags set for every item */
/* What a refinement needs of a window of divisions in one call: align2() of every pair (as g2g_align2_batch: scr, skl,
   nskl, status) plus calcSpScore of its CURRENT alignment `cur` (fs_cur) and of the new one (fs_new), flags as for
   g2g_spscore_batch_flags (<-> Prrn::onecycle, src/prrn5.cc:522-535).  Same results as the two calls; the problems are
   packed and uploaded once. */
int        g2g_align2_score_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *pw, const g2g_skl *const *cur, const int *ncur, int flags,
                                  double *scr, g2g_skl **skl, int *nskl, int *status, g2g_fstat *fs_cur, g2g_fstat *fs_new);
int        g2g_align2(g2g_ctx *ctx, g2g_pwdm *p, double *scr, g2g_skl **skl, int *nskl);
/* <-> VTYPE HomScore(mSeq* seqs[], PwdM* pwdm, long rr[]) (src/maln2.cc:1837): score only; rr may be NULL. */
int        g2g_homscore(g2g_ctx *ctx, g2g_pwdm *p, double *scr, int64_t rr[2]);
int        g2g_align2_batch(g2g_ctx *ctx, int n, g2g_pwdm *const *p, double *scr,
                            g2g_skl **skl, int *nskl, int *status);

/* ---- f2: the refinement loop --------------------------------------------------------------------------------------
 * <-> Prrn::rir (reference src/prrn5.cc:633-666) with onecycle / divideseq / gather / calcfact (:414-543), Randiv in TREEDIV
 * mode + McRand (src/randiv.cc:34-239, the glibc rand() seeding included), delcommongap / synthgap (src/mgaps.cc:181-369) and
 * gap2skl (src/gaps.cc:274): randomised iterative refinement of an MSA over the branches of its weighting tree.  The SERIAL
 * trajectory of the reference is reproduced -- same branch sequence, same accepted moves, same final MSA -- while the DPs of a
 * WINDOW of upcoming divisions run as one batch on the GPU (speculation: the divisions behind an accepted move are drawn again).
 *   codes: len x many residue codes [column][member] (gap = 1), no all-gap column.
 *   tree : Ktree::lead[] (src/phyl.h): node id = tid, leaves 0 .. many-1 are the members, left / right / parent = -1 where
 *          absent, vol / cur = the Kirchhoff weights the reference's Ktree computed (building the tree is phylogeny code,
 *          outside this path).
 *   opts : seed (Randiv's rn: 1 = "seed from rand()" as prrn does), maxitr (prrn -I, default 10), window (largest speculative
 *          batch, default 32).  rank / world / exchange: with world > 1 every rank calls g2g_refine with the SAME inputs; a
 *          window is sharded (largest rectangles first, round-robin), each rank scores its share and `exchange` all-gathers
 *          the fixed-size result slots (n_ints int32 per rank in, world x n_ints out, rank-major) -- RCCL / MPI / gloo on the
 *          caller's side; every rank then takes the same decisions and ends with the same MSA.  Return 0 from the callback.
 * Outputs: *out_codes (malloc'ed, *out_len x many; g2g_free), the trajectory (*steps, malloc'ed; may be NULL) and counters. */
typedef struct g2g_tree {
    int32_t n_nodes;                /* 2 * many - 1                                                */
    const int32_t *left, *right, *parent;
    const double  *vol, *cur;
} g2g_tree;
typedef int (*g2g_exchange_fn)(void *user, const int32_t *mine, int n_ints, int32_t *all);
/* Optional: the caller scores the divisions of a window instead of the GPU (ctx may then be NULL).  For each of the n PwdMs:
   scr[i] = the DP score and skl[i] / nskl[i] = the standardised skeleton of align2() (malloc'ed, the library frees it),
   raw_cur[i] = calcSpScore(SKL*) of the CURRENT alignment cur[i] (not rescaled), val_new[i] = fstat.val of the new one.
   The tests drive the loop -- window logic, sharding, exchange, error paths -- on a machine without a GPU this way, with the
   CPU checker in the scorer's seat; the product never sets it.  Return 0. */
typedef int (*g2g_score_fn)(void *user, int n, g2g_pwdm *const *pw, const g2g_skl *const *cur, const int *ncur,
                            double *scr, g2g_skl **skl, int *nskl, double *raw_cur, double *val_new);
/* An accepted move: the two member lists of the division (larger group first, as Prrn::divideseq orders them) and the new
   skeleton in that order (corners over the two groups' columns with their all-gap columns dropped). */
typedef void (*g2g_accept_fn)(void *user, int branch, int na, const int32_t *la, int nb, const int32_t *lb, int nskl, const g2g_skl *skl);
typedef struct g2g_refine_opts {
    int32_t seed, maxitr, window;
    int32_t rank, world, slot_cap;  /* slot_cap: most corners of a skeleton an exchange slot holds (default 4096) */
    g2g_exchange_fn exchange;       /* A rank that fails locally still enters the exchange: its error code travels in the first
                                       word of its buffer and EVERY rank returns it after the gather (no rank is left waiting in
                                       the collective).  The callback itself must fail on all ranks or on none.                */
    void   *exchange_user;
    g2g_score_fn scorer;            /* NULL: g2g_align2_batch + g2g_spscore_batch on the GPU                                   */
    void   *scorer_user;
    g2g_accept_fn on_accept;        /* optional: told about every accepted move, in order (what synthgap applies, src/prrn5.cc:536-541) */
    void   *on_accept_user;
    int32_t window_min, reserved;   /* window after an accepted move (default: 16 when len x len >= 2^22 -- MSAs of 2048 columns or more --, else 4); it doubles up
                                       to `window` while nothing is accepted */
} g2g_refine_opts;
typedef struct g2g_refine_step {
    int32_t branch, na, nb, swp, accepted, skipped;   /* skipped: neither group had a column to drop -- no DP (prrn5.cc:497) */
    double  scr, val_new, val_old, delta;             /* DP score; fstat.val of the new alignment; raw score of the current one */
    double  t_ms;                                     /* wall clock since the call began when this division's window had been scored */
} g2g_refine_step;
typedef struct g2g_refine_stats {
    int32_t divisions, accepted, batches, divisions_scored_here, divisions_wasted;
    int32_t wait_timeouts, recovered_dps, reserved;   /* scheduler waits that ran into their limit during the call / DPs re-run (g2g_ctx_counters) */
} g2g_refine_stats;
int        g2g_refine(g2g_ctx *ctx, const g2g_params *prm, int many, int len, const uint8_t *codes, const g2g_tree *tree,
                      const g2g_refine_opts *opts, uint8_t **out_codes, int *out_len, g2g_refine_step **steps, int *nsteps,
                      g2g_refine_stats *stats);

/* <-> VTYPE Ssrel::pairsum_ss(mSeq* sd, bool use_pw) (reference src/fspscore.cc:896-922; Sptree::sptree :784-821): the weighted
 * sum-of-pairs score of a whole MSA, the number prrn reports for an alignment (initial / refined, the -O4 line).  The reference
 * walks the weighting tree: a node of at most ndesc_thr = 60 leaves is scored naively (Msap::ps_nml with the pair weights of
 * Ktree::recalcpw when use_pw), a larger node is the sum of its children plus the score BETWEEN their two groups
 * (calcscore_grp: PwdM + SpScore::calcJxt).  Here the naive nodes run on the GPU (g2g_pairsum_kernel) and the joins through the
 * level-1 builders + g2g_batch_spscore; the sum is formed in the reference's order.  codes / tree as for g2g_refine.
 * Not on this path: tgapf != 1, the ether term (u0), exon-boundary annotations (spSigII), subset trees (ss->num < ss->elms). */
int        g2g_pairsum(g2g_ctx *ctx, const g2g_params *prm, int many, int len, const uint8_t *codes, const g2g_tree *tree,
                       int use_pw, double *out);

/* ---- f3: the guide-tree stage -- score-only pairwise DPs between single sequences ----------------------
 * <-> VTYPE alnScoreD(const Seq* seqs[], const Simmtx* sm, int* ends = 0) (reference src/fwd2d1.cc:324-338), global branch:
 * Fwd2d::Fwd2d (the boundary values, :58-93), Fwd2d::forwardD (:136-158, anti-diagonal order, affine gaps -(v + k u),
 * band = stripe(seqs, alprm.sh), src/aln2.cc:156-174) and Fwd2d::lastD (:100-134, terminal-gap discount tgapf).  This is what
 * dpscore() (src/phyl.cc:222-252) runs for every pair of members when the distance matrix comes from alignment scores.
 * Sequences are uploaded once, pairs are index pairs; scores are bit-equal to the reference's doubles.
 * prm: u, v, scale (floats in the reference: the library forms (float) u * (float) scale as the reference does), tgapf, sh,
 * simmtx / simdim / simrows (Simmtx::mtx as doubles).  Not on this path: the local (algmode.lcl) and `ends` variants.  */
typedef struct g2g_dseq {
    const uint8_t *res;             /* Seq::at(0): len residue codes (one member)                 */
    int32_t len, left, right;       /* Seq::len, ::left, ::right                                  */
} g2g_dseq;
int        g2g_alnscored_batch(g2g_ctx *ctx, const g2g_params *prm, int nseq, const g2g_dseq *seqs,
                               int npairs, const int32_t *ia, const int32_t *ib, double *score, int32_t *status);
/* <-> SKL* alignB_ng(const Seq* seqs[], const PwdB* pwd, VTYPE* scr) (reference src/fwd2b1.cc:1347-1353 -> globalB_ng :1286-1315
 * -> lspB_ng :1053-1095).  DPs of fewer than MaxVmfSpace cells (16 Mi; context option MAX_VMF_SPACE <-> setVmfSpace, vmf.cc:27) are
 * traced in one piece (trcbkalignB_ng :1025-1051): Aln2b1::forwardB_ng (:145-279; affine gaps, with prm->ls = 3 the second,
 * long-gap pair of layers) + initB_ng / lastB_ng (:64-143) + the Vmf record chain.  Larger ones go through the reference's
 * linear-space recursion: centerB_ng (:492-782, with binitB_ng / finitB_ng :382-490) splits the DP at the middle row into two
 * parts with narrowed windows, the parts recurse (a part of one diagonal: diagonalB_ng's two records, :1015-1021), the value is
 * the top centre's.  Then stdskl.  PwdB's constants (BasicGOP ... codonk1, src/aln2.cc:80-120) are formed from prm's u, v, u1, k1,
 * ls, scale, molc as the reference does.  skl[i]: malloc'ed corners ascending (g2g_free), nskl[i] of them.
 * Status G2G_ERR_MODE: a band of one diagonal at the top level; sides of 46341 and more (the reference's int volume overflows);
 * a pair on which centerB_ng hands back a part that is no DP (seen with ls = 3: the reference itself reads outside its arrays and
 * crashes there).  Local modes are not on this path. */
int        g2g_alignb_ng_batch(g2g_ctx *ctx, const g2g_params *prm, int nseq, const g2g_dseq *seqs, int npairs,
                               const int32_t *ia, const int32_t *ib, double *scr, g2g_skl **skl, int *nskl, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* G2G_H_ */
