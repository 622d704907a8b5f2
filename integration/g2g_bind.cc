// g2g_bind.cc -- the reference-side binding of INTEGRATION.md, option A, as real code.
//
// Compiled against the headers of ogotoh/prrn_aln where they lie (recipe: oracle/Makefile.ref, outputs under
// oracle/_ref/) and linked into the reference's own programs with
//     -Wl,--wrap=_Z6align2PP4mSeqP4PwdMPdP6Gsinfo  -lg2g
// so that every call of
//     SKL* align2(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI)           (src/maln2.cc:1875)
// made by aln.cc:337,343 and Prrn::onecycle / thread_onecycle (src/prrn5.cc:530,579) lands here.  The binding
// keeps PwdM (mode selection, profiles, gap profiles: the reference's own builders), takes the addresses of
// the arrays the DP reads, runs forward fill + traceback on the GPU through the C ABI (include/g2g.h, level 0)
// and hands the raw Vmf-ordered corners back to the reference's own stdskl / calcSpScore / trimskl.
// What the GPU path does not cover (quick mode, rectangular _ALN other than NGP_ALN, spliced _ALH/_ALS, the ether `_p` scorers)
// goes to the reference's align2 unchanged -- in the reference's process that is the documented contract of
// the boundary, not a fallback inside the product library.
//
// G2G_BIND=off     every call goes to the reference (A/B switch)
// G2G_BIND=verify  both run; score and skeleton are compared, mismatches counted, the reference's result returned
// G2G_BIND_MIN_CELLS=<n>  DPs whose rectangle has fewer than n cells stay on the CPU (default 0: none)
// A summary line is written to stderr at exit when G2G_BIND_STATS is set.

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>
#include <mutex>
#include <thread>
#include <chrono>
#include <condition_variable>
#include "aln.h"
#include "mseq.h"
#include "maln.h"
#include "gfreq.h"
#include "fwd2c.h"
#include "gaps.h"
#include "phyl.h"
#include "consreg.h"
#include "fspscore.h"
#include "g2g.h"

extern "C" SKL* __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI);

namespace {

struct Stats {
	long	calls, gpu, cpu, mismatch, batches, maxbatch;
	long	dcalls, dgpu, dcpu, dmismatch, dbatches, dmaxbatch;	// alnScoreD (the guide-tree stage)
	~Stats() {
	    if (getenv("G2G_BIND_STATS")) {
		fprintf(stderr, "g2g_bind: %ld align2 calls, %ld on the GPU, %ld by the reference, %ld mismatches"
		    "; %ld GPU batches, largest %ld\n", calls, gpu, cpu, mismatch, batches, maxbatch);
		fprintf(stderr, "g2g_bind: %ld alnScoreD calls, %ld on the GPU, %ld by the reference, %ld mismatches"
		    "; %ld GPU batches, largest %ld\n", dcalls, dgpu, dcpu, dmismatch, dbatches, dmaxbatch);
	    }
	}
} stats = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
std::mutex	stats_mu;		// (prrn5 -t<n> calls align2 from several pthreads)
#define COUNT(field) do {std::lock_guard<std::mutex> l_(stats_mu); ++stats.field;} while (0)

// prrn5 -t<n>: thread_onecycle (src/prrn5.cc:565-592) aligns n candidate divisions in n pthreads at once.  One g2g_ctx
// serves them all, and instead of running their DPs one after the other the calls that arrive together are run as
// ONE g2g_forward_batch: the first caller leads, collects whatever arrives until the callers have been quiet for 0.5 ms, launches, and
// hands every waiting caller its own result.  A lone caller (serial prrn5, aln) goes straight through.
struct Req {const g2g_problem* p; g2g_result r; bool done;};
struct Batcher {
	std::mutex	mu;
	std::condition_variable	cv;
	std::vector<Req*>	pending;
	std::set<std::thread::id>	callers;
	bool	busy;
	Batcher() : busy(false) {}
	void submit(g2g_ctx* ctx, Req* q) {
	    std::unique_lock<std::mutex>	lk(mu);
	    callers.insert(std::this_thread::get_id());
	    pending.push_back(q);
	    cv.notify_all();
	    while (!q->done) {
		if (busy) {cv.wait(lk); continue;}
		busy = true;					// this caller leads the next batch
		if (callers.size() > 1) {			// let the other threads' calls arrive: until none has come for
		    // `quiet` us (default 500), `quiet` x 15 at most.  A batch costs about the same 50-100 ms whether it holds
		    // one DP or sixty (a DP is a latency-bound chain), so a few ms of waiting for company are well spent.
		    static const long	quiet = getenv("G2G_BIND_QUIET_US")? atol(getenv("G2G_BIND_QUIET_US")): 500;
		    const auto	t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(15 * quiet);
		    for (;;) {
			const size_t	before = pending.size();
			cv.wait_for(lk, std::chrono::microseconds(quiet));
			if (pending.size() == before || pending.size() >= callers.size() ||
			    std::chrono::steady_clock::now() >= t_end) break;
		    }
		}
		std::vector<Req*>	mine;
		mine.swap(pending);
		lk.unlock();
		const int	n = (int) mine.size();
		std::vector<const g2g_problem*>	probs(n);
		std::vector<g2g_result>	res(n);
		for (int i = 0; i < n; ++i) {probs[i] = mine[i]->p; memset(&res[i], 0, sizeof(g2g_result));}
		const int	rc = g2g_forward_batch(ctx, n, probs.data(), res.data());
		for (int i = 0; i < n; ++i) {
		    mine[i]->r = res[i];
		    if (rc != G2G_OK && mine[i]->r.status == G2G_OK) mine[i]->r.status = rc;
		}
		{
		    std::lock_guard<std::mutex>	l2(stats_mu);
		    ++stats.batches;
		    if (n > stats.maxbatch) stats.maxbatch = n;
		}
		lk.lock();
		for (int i = 0; i < n; ++i) mine[i]->done = true;
		busy = false;
		cv.notify_all();
	    }
	}
} batcher;

g2g_ctx* context()
{
	static g2g_ctx*	ctx = 0;
	static std::once_flag	once;
	std::call_once(once, []() {
	    ctx = g2g_create(-1);
	    if (!ctx) fprintf(stderr, "g2g_bind: no GPU context (%s); align2 stays on the CPU\n", g2g_last_error());
	});
	return ctx;
}

// which PwdM::sim?? / crg?? the constructor selected (maln2.cc:288-483); ids as in include/g2g.h
int scorer_id(const PwdM* pwd)
{
#define K(fn, id) if (pwd->Sim2 == &PwdM::fn) return id;
	K(sim11, 11) K(sim12i, 120) K(sim12w, 121) K(sim13, 13)
	K(sim21i, 210) K(sim21w, 211) K(sim22i, 220) K(sim22w, 221)
	K(sim23i, 230) K(sim23w, 231) K(sim31, 31) K(sim32i, 320) K(sim32w, 321)
	K(sim33, 33) K(sim33_n, 330)
#undef K
	return -1;
}
int crg_id(const PwdM* pwd)
{
#define K(fn, id) if (pwd->crg2 == &PwdM::fn) return id;
	K(crg11, 11) K(crg12i, 120) K(crg12w, 121) K(crg21i, 210) K(crg21w, 211)
	K(crg22i, 220) K(crg22w, 221)
#undef K
	return -1;
}

struct SideBuf {
	std::vector<double>	thk, gd, pg, wt, pdns;
	std::vector<int32_t>	ppos;
	std::vector<int32_t>	off[3], glen[3];
	std::vector<double>	freq[3];
};

void side(g2g_side& s, mSeq* q, const FTYPE* wt, bool naive, SideBuf& B)
{
	memset(&s, 0, sizeof(s));
	s.many = q->many; s.len = q->len; s.left = q->left; s.right = q->right;
	s.nils = q->inex.nils; s.dels = q->inex.dels;
	s.sumwt = q->sumwt;
	s.seq = q->at(-1);				// Seq::at(), seq.h:348: [pos][member], position -1 first
	if (!wt) wt = q->weight;
	if (wt) {B.wt.assign(wt, wt + q->many); s.weight = B.wt.data();}
	s.nelm = q->nelm; s.felm = q->felm;
	s.pseq = (q->inex.vect && q->pseq)? q->pseq: 0;	// fat(-1) .. fat(len), mseq.h:123
	B.thk.resize((size_t) 3 * (q->len + 2));		// SeqThk as mSeqItr yields it (mseq.h:222-250)
	for (int i = -1; i <= q->len; ++i) {
	    mSeqItr	it(q, i);
	    const SeqThk*	t = it.dns;
	    double*	d = &B.thk[(size_t) 3 * (i + 1)];
	    d[0] = t? t->cfq: 0; d[1] = t? t->dfq: 0; d[2] = t? t->efq: 0;
	}
	s.thk = B.thk.data();
	if (naive) {					// NTV engines: mSeq::gapdensity / postgapdensity, mseq.h:148-160
	    B.gd.assign((size_t) (q->len + 2) * q->many, 0.); B.pg = B.gd;
	    for (int i = -1; i < q->len; ++i)
		for (int k = 0; k < q->many; ++k) {
		    CHAR*	r = q->at(i) + k;
		    const size_t	ix = (size_t) (i + 1) * q->many + k;
		    B.gd[ix] = q->gapdensity(r, k);
		    B.pg[ix] = q->internalres? q->postgapdensity(r, k): 1;
		}
	    s.gapdens = B.gd.data(); s.postgapdens = B.pg.data();
	}
	if (q->sigII && q->sigII->pfqnum > 0 && q->sigII->pfq) {	// exon boundaries (SigII::pfq, gsinfo.h:33,41-46): the intron-position bonus
	    for (int i = 0; i < q->sigII->pfqnum; ++i) {B.ppos.push_back(q->sigII->pfq[i].pos); B.pdns.push_back(q->sigII->pfq[i].dns);}
	    s.npfq = q->sigII->pfqnum; s.pfq_step = q->sigII->step;
	    s.pfq_pos = B.ppos.data(); s.pfq_dns = B.pdns.data();
	}
	s.has_gfq = (q->gfq && q->inex.dels)? 1: 0;
	if (s.has_gfq) {				// the three views, lists with their terminators (gfreq.cc:230-312)
	    s.gfq.hetero = q->gfq->hetero;
	    GFREQ**	views[3] = {q->gfq->sfrq, q->gfq->tfrq, q->gfq->rfrq};
	    for (int v = 0; v < 3; ++v) {
		B.off[v].clear(); B.glen[v].clear(); B.freq[v].clear();
		for (int i = -1; i < q->len; ++i) {
		    B.off[v].push_back((int32_t) B.glen[v].size());
		    for (const GFREQ* g = views[v][i]; ; ++g) {
			B.glen[v].push_back(g->glen); B.freq[v].push_back(g->freq);
			if (!neogfq(g)) break;
		    }
		}
		B.off[v].push_back((int32_t) B.glen[v].size());
		s.gfq.off[v] = B.off[v].data(); s.gfq.glen[v] = B.glen[v].data(); s.gfq.freq[v] = B.freq[v].data();
	    }
	}
}

// <-> alignC<recd_t>(seqs, pwd, scr) (fwd2c.h:671-677): forward fill + Vmf traceback.  0: not on the GPU path.
SKL* alignC_g2g(mSeq* seqs[], PwdM* pwd, VTYPE* scr)
{
	g2g_ctx*	ctx = context();
	if (!ctx) return 0;
	const int	sk = scorer_id(pwd);
	const bool	naive = pwd->alnmode == NTV_ALB;
	const int	ck = naive? crg_id(pwd): 0;
	if (sk < 0 || ck < 0) return 0;			// ether / secondary-structure scorers
	g2g_problem	p;
	memset(&p, 0, sizeof(p));
	p.alnmode = pwd->alnmode; p.noll = pwd->Noll; p.codonk1 = pwd->codonk1;
	p.sim2_kind = sk; p.crg2_kind = ck; p.dvsp = pwd->DvsP;
	p.basic_gop = pwd->Basic_GOP; p.weighted_gop = pwd->Weighted_GOP; p.u = pwd->alnprm.u;
	p.u2divu1 = pwd->BasicGEP < 0? pwd->LongGEP / pwd->BasicGEP: 0;		// fwd2c.h:85-86
	p.v2divv1 = pwd->BasicGOP < 0? pwd->LongGOP / pwd->BasicGOP: 0;
	WINDOW	w;
	stripe((const Seq**) seqs, &w, pwd->alnprm.sh);				// aln2.cc:156-174
	p.lw = w.lw; p.up = w.up;
	const Simmtx*	sm = pwd->simmtx;
	std::vector<double>	mtx((size_t) sm->rows * sm->dim);
	for (int i = 0; i < sm->rows; ++i)
	    for (int j = 0; j < sm->dim; ++j) mtx[(size_t) i * sm->dim + j] = sm->mtx[i][j];
	p.simmtx = mtx.data(); p.simrows = sm->rows; p.simdim = sm->dim;
	SideBuf	A, B;
	side(p.a, seqs[0], pwd->wta, naive, A);
	side(p.b, seqs[1], pwd->wtb, naive, B);
	p.spb_fact = SpbFact;						// gsinfo.cc:33-35
	Req	q;
	q.p = &p; q.done = false;
	memset(&q.r, 0, sizeof(q.r));
	batcher.submit(ctx, &q);
	const g2g_result&	r = q.r;
	if (r.status != G2G_OK) {
	    if (r.trace) g2g_free(r.trace);
	    return 0;
	}
	*scr = r.score;
	SKL*	skl = new SKL[r.ntrace + 1];					// Vmf::traceback layout, vmf.cc:105-120
	skl->m = 0; skl->n = r.ntrace;
	for (int i = 0; i < r.ntrace; ++i) {skl[i + 1].m = r.trace[i].m; skl[i + 1].n = r.trace[i].n;}
	g2g_free(r.trace);
	return skl;
}

bool on_gpu_path(mSeq* seqs[], PwdM* pwdm, Gsinfo* GsI)
{
	if (!GsI || (algmode.qck & 1)) return false;
	if (seqs[0]->left == seqs[0]->right || seqs[1]->left == seqs[1]->right) return false;
	// G2G_BIND_MIN_CELLS=<n>: rectangles smaller than n cells stay on the CPU (one GPU round trip costs ~2 ms of launches
	// and synchronisation, about 3e4 cells of a host core); default 0 = everything on the path goes to the GPU
	static const double	min_cells = getenv("G2G_BIND_MIN_CELLS")? atof(getenv("G2G_BIND_MIN_CELLS")): 0;
	if (min_cells > 0 && (double) (seqs[0]->right - seqs[0]->left) * (seqs[1]->right - seqs[1]->left) < min_cells) return false;
	// (Intron-position bonus -- PfqItr::match_score, fwd2c.h:367-379,446-452, live when BOTH inputs carry exon-boundary
	//  annotations and SpbFact != 0: the flattened problem carries the boundary lists since ABI 3 and libg2g.so applies it.
	//  G2G_BIND_NO_INTRON=1 keeps such pairs with the reference's own forwardB, as before.)
	static const bool	no_intron = getenv("G2G_BIND_NO_INTRON") != 0;
	if (no_intron && SpbFact != 0 && seqs[0]->sigII && seqs[1]->sigII && seqs[0]->sigII->pfqnum && seqs[1]->sigII->pfqnum) return false;
	switch (pwdm->alnmode) {
	    case NGP_ALB: case HLF_ALB: case RHF_ALB: case GPF_ALB: case NTV_ALB: return true;
	    // the rectangular engine (-A: algmode.bnd = 0 -> alignC<recd_t>(..., rectangle) = Fwd2c::forwardA, fwd2c.h:232-356): on the GPU
	    // for DPunit; the record types with gap state stay with the reference (libg2g.so answers G2G_ERR_MODE for them: DESIGN.md 1)
	    case NGP_ALN: return true;
	    default: return false;
	}
}

// align2 with the DP on the GPU; everything after the traceback is the reference's own code
SKL* align2_gpu(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI, bool& handled)
{
	mSeq*	a = seqs[0];
	mSeq*	b = seqs[1];
	handled = true;
	for (int attempt = 0; attempt < 2; ++attempt) {
	    SKL*	skl = alignC_g2g(seqs, pwdm, scr);
	    if (!skl) {handled = false; return 0;}
	    if (!skl->n) {delete[] skl; return 0;}
	    GsI->skl = stdskl(&skl);
	    const int	num = skl->n;
	    if (skl[1].m == a->left && skl[num].m == a->right && skl[1].n == b->left && skl[num].n == b->right) {
		PreSpScore	pss(seqs, pwdm);
		pss.calcSpScore(GsI);
		if (OutPrm.trimend) skl = trimskl((const Seq**) seqs, skl);
		GsI->skl = 0;
		return skl;
	    }
	    delete[] skl;			// path fell off the band: once more with the widest band (sh = -100)
	    GsI->skl = 0;
	    pwdm->alnprm.sh = -100;
	}
	handled = false;
	return 0;
}

bool same_skl(const SKL* x, const SKL* y)
{
	if (!x || !y) return x == y;
	if (x->n != y->n) return false;
	for (int i = 1; i <= x->n; ++i) if (x[i].m != y[i].m || x[i].n != y[i].n) return false;
	return true;
}

}	// namespace

extern "C" SKL* __wrap__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI)
{
	static const char*	mode = getenv("G2G_BIND");
	COUNT(calls);
	const bool	off = mode && !strcmp(mode, "off");
	const bool	verify = mode && !strcmp(mode, "verify");
	if (off || !on_gpu_path(seqs, pwdm, GsI)) {
	    COUNT(cpu);
	    return __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(seqs, pwdm, scr, GsI);
	}
	if (verify) {
	    const int	sh0 = pwdm->alnprm.sh;
	    Gsinfo	g2;
	    VTYPE	s2 = 0;
	    bool	handled = false;
	    SKL*	mine = align2_gpu(seqs, pwdm, &s2, &g2, handled);
	    pwdm->alnprm.sh = sh0;
	    SKL*	ref = __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(seqs, pwdm, scr, GsI);
	    if (handled) {
		COUNT(gpu);
		if (!same_skl(mine, ref) || (ref && (s2 != *scr || g2.fstat.val != GsI->fstat.val))) {
		    COUNT(mismatch);
		    fprintf(stderr, "g2g_bind: MISMATCH mode %d score %.17g vs %.17g\n", pwdm->alnmode, s2, *scr);
		}
	    } else COUNT(cpu);
	    delete[] mine;
	    return ref;
	}
	bool	handled = false;
	SKL*	skl = align2_gpu(seqs, pwdm, scr, GsI, handled);
	if (handled) {COUNT(gpu); return skl;}
	COUNT(cpu);
	return __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(seqs, pwdm, scr, GsI);
}

// ---- f3: alnScoreD (src/fwd2d1.cc:324-338), the score-only DP behind dpscore() (src/phyl.cc:222-252) ------------------
// DistMat runs dpscore for every pair of members, one call per pair, from as many pthreads as -t gives: the calls that
// arrive together go to the device as ONE g2g_alnscored_batch (same leader scheme as align2 above).
extern "C" VTYPE __real__Z9alnScoreDPPK3SeqPK6SimmtxPi(const Seq* seqs[], const Simmtx* sm, int* ends);

namespace {

struct DReq {const Seq* a; const Seq* b; const Simmtx* sm; double score; int status; bool done;};
struct DistBatcher {
	std::mutex	mu;
	std::condition_variable	cv;
	std::vector<DReq*>	pending;
	std::set<std::thread::id>	callers;
	bool	busy;
	DistBatcher() : busy(false) {}
	void run(g2g_ctx* ctx, std::vector<DReq*>& mine) {
	    const int	n = (int) mine.size();
	    const Simmtx*	sm = mine[0]->sm;
	    std::vector<double>	mtx((size_t) sm->rows * sm->dim);
	    for (int i = 0; i < sm->rows; ++i)
		for (int j = 0; j < sm->dim; ++j) mtx[(size_t) i * sm->dim + j] = sm->mtx[i][j];
	    g2g_params	prm;
	    memset(&prm, 0, sizeof prm);
	    prm.u = alprm.u; prm.v = alprm.v; prm.scale = alprm.scale; prm.tgapf = alprm.tgapf; prm.sh = alprm.sh;
	    prm.u1 = alprm.u1; prm.k1 = alprm.k1; prm.ls = alprm.ls;
	    prm.simmtx = mtx.data(); prm.simdim = sm->dim; prm.simrows = sm->rows;
	    std::vector<g2g_dseq>	sq(2 * (size_t) n);
	    std::vector<int32_t>	ia(n), ib(n), st(n);
	    std::vector<double>	sc(n);
	    for (int i = 0; i < n; ++i) {
		const Seq*	s2[2] = {mine[i]->a, mine[i]->b};
		for (int k = 0; k < 2; ++k) {
		    g2g_dseq&	d = sq[2 * i + k];
		    d.res = s2[k]->at(0); d.len = s2[k]->len; d.left = s2[k]->left; d.right = s2[k]->right;
		}
		ia[i] = 2 * i; ib[i] = 2 * i + 1;
	    }
	    const int	rc = g2g_alnscored_batch(ctx, &prm, 2 * n, sq.data(), n, ia.data(), ib.data(), sc.data(), st.data());
	    for (int i = 0; i < n; ++i) {mine[i]->score = sc[i]; mine[i]->status = rc != G2G_OK? rc: st[i];}
	    std::lock_guard<std::mutex>	l2(stats_mu);
	    ++stats.dbatches;
	    if (n > stats.dmaxbatch) stats.dmaxbatch = n;
	}
	void submit(g2g_ctx* ctx, DReq* q) {
	    std::unique_lock<std::mutex>	lk(mu);
	    callers.insert(std::this_thread::get_id());
	    pending.push_back(q);
	    cv.notify_all();
	    while (!q->done) {
		if (busy) {cv.wait(lk); continue;}
		busy = true;
		if (callers.size() > 1) {
		    static const long	quiet = getenv("G2G_BIND_QUIET_US")? atol(getenv("G2G_BIND_QUIET_US")): 500;
		    const auto	t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(15 * quiet);
		    for (;;) {
			const size_t	before = pending.size();
			cv.wait_for(lk, std::chrono::microseconds(quiet));
			if (pending.size() == before || pending.size() >= callers.size() ||
			    std::chrono::steady_clock::now() >= t_end) break;
		    }
		}
		std::vector<DReq*>	mine, rest;
		for (DReq* r : pending) (r->sm == pending[0]->sm? mine: rest).push_back(r);	// (one matrix per batch)
		pending.swap(rest);
		lk.unlock();
		run(ctx, mine);
		lk.lock();
		for (DReq* r : mine) r->done = true;
		busy = false;
		cv.notify_all();
	    }
	}
} dist_batcher;

}	// namespace

extern "C" VTYPE __wrap__Z9alnScoreDPPK3SeqPK6SimmtxPi(const Seq* seqs[], const Simmtx* sm, int* ends)
{
	static const char*	mode = getenv("G2G_BIND");
	COUNT(dcalls);
	const bool	off = mode && !strcmp(mode, "off");
	const bool	verify = mode && !strcmp(mode, "verify");
	const Seq*	a = seqs[0];
	const Seq*	b = seqs[1];
	if (!sm) sm = getSimmtx(0);
	// on the GPU path: the global, score-only branch between two single sequences taken whole (Fwd2d with exgl = exgr = 0 and
	// left = 0, what dpscore passes); everything else is the reference's
	static const bool	no_dist = getenv("G2G_BIND_NO_DIST") != 0;	// (A/B switch of this wrap alone)
	g2g_ctx*	ctx = (off || no_dist)? 0: context();
	if (!ctx || ends || (algmode.lcl & 16) || a->many != 1 || b->many != 1 || a->left || b->left ||
	    a->inex.exgl || b->inex.exgl || a->inex.exgr || b->inex.exgr || a->right <= a->left || b->right <= b->left) {
	    COUNT(dcpu);
	    return __real__Z9alnScoreDPPK3SeqPK6SimmtxPi(seqs, sm, ends);
	}
	DReq	q = {a, b, sm, 0, 0, false};
	dist_batcher.submit(ctx, &q);
	if (q.status != G2G_OK) {
	    COUNT(dcpu);
	    return __real__Z9alnScoreDPPK3SeqPK6SimmtxPi(seqs, sm, ends);
	}
	COUNT(dgpu);
	if (verify) {
	    const VTYPE	ref = __real__Z9alnScoreDPPK3SeqPK6SimmtxPi(seqs, sm, ends);
	    if (ref != (VTYPE) q.score) {
		COUNT(dmismatch);
		fprintf(stderr, "g2g_bind: MISMATCH alnScoreD %.17g vs %.17g\n", q.score, (double) ref);
	    }
	    return ref;
	}
	return (VTYPE) q.score;
}
