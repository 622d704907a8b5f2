// ref_shim.cc -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// A thin extern "C" window onto the *real* reference (ogotoh/prrn_aln), compiled against the
// reference headers where they lie (oracle/Makefile.ref) and linked into oracle/_ref/libprrn_ref.so.
// It lets tests/ and tools/make_golden.py
//   * read a group (an MSA file) with the reference's own reader,
//   * run the reference's PwdM constructor + alignC<recd_t>() / align2() on a pair of groups,
//   * and serialise everything the group-to-group DP touched (flattened inputs: SURVEY §8 a1/a7/a8/a9
//     arrays) together with what it produced (score, raw VMF traceback, stdskl skeleton, fstat).
// The dump is a flat list of named arrays; tests/refdump.py parses it.
//
// Nothing here restates reference logic: every number written is read out of reference objects.

#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <time.h>
// compiled with -fno-access-control: the dump reads PwdM's private scalars and fn-pointers
#include "aln.h"
#include "vmf.h"
#include "mseq.h"
#include "maln.h"
#include "gfreq.h"
#include "fwd2c.h"
#include "phyl.h"
#include "consreg.h"
#include "fspscore.h"

// every reference main() supplies usage(); library code calls it on bad options (autocomp.h)
void usage() {}

namespace {

struct Dump {
	FILE*	fd;
	explicit Dump(const char* path) : fd(fopen(path, "wb")) {
	    if (fd) fwrite("G2GD0001", 1, 8, fd);
	}
	~Dump() {if (fd) fclose(fd);}
	// dtype codes: 0=u8 1=i32 2=i64 3=f64
	void put(const char* name, int dtype, const void* data, int64_t n0, int64_t n1 = -1) {
	    char	nm[32];
	    memset(nm, 0, sizeof(nm));
	    strncpy(nm, name, 31);
	    fwrite(nm, 1, 32, fd);
	    int32_t	hd[2] = {dtype, n1 < 0? 1: 2};
	    fwrite(hd, 4, 2, fd);
	    int64_t	shp[2] = {n0, n1 < 0? 1: n1};
	    fwrite(shp, 8, 2, fd);
	    static const int esz[] = {1, 4, 8, 8};
	    int64_t	cnt = n0 * (n1 < 0? 1: n1);
	    if (cnt) fwrite(data, esz[dtype], cnt, fd);
	}
	void i32(const char* name, int v) {int32_t x = v; put(name, 1, &x, 1);}
	void f64(const char* name, double v) {put(name, 3, &v, 1);}
};

int sim2_kind(const PwdM* pwd)
{
#define K(fn, id) if (pwd->Sim2 == &PwdM::fn) return id;
	K(sim00, 0) K(sim11, 11) K(sim12i, 120) K(sim12w, 121) K(sim13, 13)
	K(sim21i, 210) K(sim21w, 211) K(sim22i, 220) K(sim22w, 221)
	K(sim23i, 230) K(sim23w, 231) K(sim31, 31) K(sim32i, 320) K(sim32w, 321)
	K(sim33, 33) K(sim33_n, 330)
#undef K
	return -1;	// ether (_p) / sshp variants: not dumped
}

int crg2_kind(const PwdM* pwd)
{
#define K(fn, id) if (pwd->crg2 == &PwdM::fn) return id;
	K(crg11, 11) K(crg12i, 120) K(crg12w, 121) K(crg21i, 210) K(crg21w, 211)
	K(crg22i, 220) K(crg22w, 221)
#undef K
	return -1;
}

void dump_side(Dump& d, const char* pfx, mSeq* s)
{
	std::string p(pfx);
	d.i32((p + "many").c_str(), s->many);
	d.i32((p + "len").c_str(), s->len);
	d.i32((p + "left").c_str(), s->left);
	d.i32((p + "right").c_str(), s->right);
	d.i32((p + "molc").c_str(), s->inex.molc);
	d.i32((p + "dels").c_str(), s->inex.dels);
	d.i32((p + "nils").c_str(), s->inex.nils);
	d.i32((p + "exgl").c_str(), s->inex.exgl);
	d.i32((p + "exgr").c_str(), s->inex.exgr);
	d.i32((p + "vect").c_str(), s->inex.vect);
	d.i32((p + "sngl").c_str(), s->inex.sngl);
	d.i32((p + "max_code").c_str(), s->code->max_code);
	d.f64((p + "sumwt").c_str(), s->sumwt);
	// residues at(-1) .. at(len)   (seq.h:348, column-major [pos][member])
	d.put((p + "seq").c_str(), 0, s->at(-1), s->len + 2, s->many);
	if (s->weight) {
	    std::vector<double> w(s->weight, s->weight + s->many);
	    d.put((p + "weight").c_str(), 3, w.data(), s->many);
	}
	if (s->sigII && s->sigII->pfqnum > 0) {		// exon-boundary annotation (SigII::pfq, gsinfo.h:33,41-46)
	    std::vector<int>	pos(s->sigII->pfqnum);
	    std::vector<double>	dns(s->sigII->pfqnum);
	    for (int i = 0; i < s->sigII->pfqnum; ++i) {pos[i] = s->sigII->pfq[i].pos; dns[i] = s->sigII->pfq[i].dns;}
	    d.put((p + "pfq_pos").c_str(), 1, pos.data(), s->sigII->pfqnum);
	    d.put((p + "pfq_dns").c_str(), 3, dns.data(), s->sigII->pfqnum);
	    d.i32((p + "pfq_step").c_str(), s->sigII->step);
	}
	d.i32((p + "nelm").c_str(), s->nelm);
	d.i32((p + "felm").c_str(), s->felm);
	if (s->inex.vect && s->pseq)		// fat(-1) .. fat(len)  (mseq.h:123)
	    d.put((p + "pseq").c_str(), 3, s->pseq, s->len + 2, s->nelm);
	// thickness as the iterator hands it to the DP: one SeqThk per position -1..len
	// (mSeqItr::reset / repos / operator++, mseq.h:222-250, mseq.cc:760-790)
	{
	    std::vector<double> t(3 * (s->len + 2));
	    mSeqItr	it(s, -1);
	    d.i32((p + "thk_mode").c_str(), it.thk_mode);
	    for (int i = -1; i <= s->len; ++i) {
		mSeqItr	jt(s, i);
		const SeqThk*	q = jt.dns;
		t[3 * (i + 1)] = q? q->cfq: 0;
		t[3 * (i + 1) + 1] = q? q->dfq: 0;
		t[3 * (i + 1) + 2] = q? q->efq: 0;
	    }
	    d.put((p + "thk").c_str(), 3, t.data(), s->len + 2, 3);
	}
	// member gap densities as the naive engine reads them (mSeq::gapdensity / postgapdensity,
	// mseq.h:148-160), positions -1 .. len-1 (row `len` left 0)
	{
	    std::vector<double> gd((size_t) (s->len + 2) * s->many, 0.), pg(gd);
	    for (int i = -1; i < s->len; ++i) {
		for (int k = 0; k < s->many; ++k) {
		    CHAR*	r = s->at(i) + k;
		    size_t	ix = (size_t) (i + 1) * s->many + k;
		    gd[ix] = s->gapdensity(r, k);
		    pg[ix] = s->internalres? s->postgapdensity(r, k): 1;
		}
	    }
	    d.put((p + "gapdens").c_str(), 3, gd.data(), s->len + 2, s->many);
	    d.put((p + "postgapdens").c_str(), 3, pg.data(), s->len + 2, s->many);
	}
	// static gap profiles: lists for positions -1 .. len-1 flattened with their -1 terminators
	d.i32((p + "has_gfq").c_str(), s->gfq? 1: 0);
	if (s->gfq && s->inex.dels) {
	    d.i32((p + "hetero").c_str(), s->gfq->hetero);
	    GFREQ**	views[3] = {s->gfq->sfrq, s->gfq->tfrq, s->gfq->rfrq};
	    const char*	vn[3] = {"sfq", "tfq", "rfq"};
	    for (int v = 0; v < 3; ++v) {
		std::vector<int32_t> off, glen, nres;
		std::vector<double> freq;
		for (int i = -1; i < s->len; ++i) {
		    off.push_back((int32_t) glen.size());
		    const GFREQ*	g = views[v][i];
		    for ( ; ; ++g) {
			glen.push_back(g->glen);
			freq.push_back(g->freq);
			nres.push_back(g->nres);
			if (!neogfq(g)) break;
		    }
		}
		off.push_back((int32_t) glen.size());
		d.put((p + vn[v] + "_off").c_str(), 1, off.data(), off.size());
		d.put((p + vn[v] + "_glen").c_str(), 1, glen.data(), glen.size());
		d.put((p + vn[v] + "_freq").c_str(), 3, freq.data(), freq.size());
		d.put((p + vn[v] + "_nres").c_str(), 1, nres.data(), nres.size());
	    }
	}
}

void dump_skl(Dump& d, const char* name, const SKL* skl)
{
	if (!skl) {d.put(name, 1, 0, 0, 2); return;}
	int	n = skl->n;
	std::vector<int32_t> v;
	for (int i = 1; i <= n; ++i) {v.push_back(skl[i].m); v.push_back(skl[i].n);}
	d.put(name, 1, v.data(), n, 2);
}

}	// namespace

extern "C" {

int ref_shim_version() {return 3;}

// prrn5's defaults: static setdefparam() (prrn5.cc:1262-1278) + the per-molecule part of
// main() (prrn5.cc:1813-1822).  molc: PROTEIN=1, DNA=2 (cmn.h:107).  ls: 1/2 affine, 3 double affine.
int ref_init_prrn(int molc, int ls, int sh)
{
	optimize(GLOBAL, MAXIMUM);
	setlsegs(1);
	setalgmode(0, 0);
	setdefPprm(250, 2., 9., 0);
	setdefNprm(-2., 2., 4.);
	alprm.sh = -60;
	alprm.thr = 70;
	algmode.any = 3;	// DynScr
	algmode.mns = 1;
	algmode.nsa = 1;
	OutPrm.SkipLongGap = 0;
	if (ls > 0) {alprm.ls = ls; setlsegs(ls);}
	if (sh) alprm.sh = sh;
	spb_fact();
	setdefmolc(molc);
	prePwd(molc);
	return 0;
}

void ref_set_tgapf(double f) {alprm.tgapf = (float) f;}
// gap extension / opening penalties as -u / -v set them (alprm is read by Fwd2d / PwdB at construction)
void ref_set_uv(double u, double v) {alprm.u = (float) u; alprm.v = (float) v;}
void ref_set_band(int bnd) {algmode.bnd = bnd? 1: 0;}
// MaxVmfSpace (vmf.cc:25-30; aln's own option sets it the same way, aln.cc:131): DPs of this many cells or more go through
// the linear-space recursion lspB_ng (fwd2b1.cc:1053-1095)
void ref_set_vmfspace(long spc) {setVmfSpace(spc);}
void ref_set_quick(int q) {algmode.qck = q;}
// per-cell trace of forwardB to stdout: "m n dir H diag G F1 [G2 F2]" (fwd2c.h:454-464)
void ref_set_debug(int on) {OutPrm.debug = on? 1: 0; fflush(stdout);}

void* ref_group_read(const char* fname)
{
	mSeq*	sd = new mSeq(fname);
	if (!sd->many || !sd->len) {delete sd; return 0;}
	return sd;
}

void ref_group_free(void* g) {delete (mSeq*) g;}
int ref_group_many(void* g) {return ((mSeq*) g)->many;}
int ref_group_len(void* g) {return ((mSeq*) g)->len;}

// per-member weights as prrn's aggregate() would have left them (mgaps.cc:296-320)
void ref_group_set_weight(void* g, const double* w)
{
	mSeq*	sd = (mSeq*) g;
	if (!sd->weight) sd->weight = new FTYPE[sd->many];
	for (int i = 0; i < sd->many; ++i) sd->weight[i] = (FTYPE) w[i];
}

// Runs PwdM(seqs) + the Fwd2c engine the reference dispatcher would pick (maln2.cc:1899-1910) and
// writes inputs + outputs to `path`.  Returns alnmode, or <0.
int ref_align_dump(void* ga, void* gb, const char* path)
{
	mSeq*	sqs[3] = {(mSeq*) ga, (mSeq*) gb, 0};
	// Prrn::gather() ends with exg_seq(exgl, exgr) (prrn5.cc:484)
	sqs[0]->exg_seq(sqs[0]->inex.exgl, sqs[0]->inex.exgr);
	sqs[1]->exg_seq(sqs[1]->inex.exgl, sqs[1]->inex.exgr);
	PwdM	pwd(sqs);		// may swap sqs[0], sqs[1]
	mSeq*	a = sqs[0];
	mSeq*	b = sqs[1];
	Dump	d(path);
	if (!d.fd) return -1;

	d.i32("alnmode", pwd.alnmode);
	d.i32("swp", pwd.swp);
	d.i32("a_mode", pwd.a_mode);
	d.i32("b_mode", pwd.b_mode);
	d.i32("aprof", pwd.aprof);
	d.i32("bprof", pwd.bprof);
	d.i32("sim2_kind", sim2_kind(&pwd));
	d.i32("crg2_kind", (pwd.alnmode == NTV_ALB || pwd.alnmode == NTV_ALN)? crg2_kind(&pwd): 0);
	d.i32("DvsP", pwd.DvsP);
	d.i32("Noll", pwd.Noll);
	d.i32("codonk1", pwd.codonk1);
	d.f64("Vab", pwd.Vab);
	d.f64("BasicGOP", pwd.BasicGOP);
	d.f64("BasicGEP", pwd.BasicGEP);
	d.f64("LongGOP", pwd.LongGOP);
	d.f64("LongGEP", pwd.LongGEP);
	d.f64("Basic_GOP", pwd.Basic_GOP);
	d.f64("Weighted_GOP", pwd.Weighted_GOP);
	d.f64("diff_u", pwd.diff_u);
	d.f64("van", pwd.van);
	d.f64("vbn", pwd.vbn);
	d.f64("alnprm_u", pwd.alnprm.u);
	d.f64("alnprm_v", pwd.alnprm.v);
	d.f64("alnprm_u0", pwd.alnprm.u0);
	d.f64("alnprm_u1", pwd.alnprm.u1);
	d.f64("alnprm_tgapf", pwd.alnprm.tgapf);
	d.f64("alnprm_scale", pwd.alnprm.scale);
	d.f64("alnprm_gamma", pwd.alnprm.gamma);
	d.f64("alprm_gamma", alprm.gamma);
	d.i32("alnprm_k1", pwd.alnprm.k1);
	d.i32("alnprm_ls", pwd.alnprm.ls);
	d.i32("alnprm_sh", pwd.alnprm.sh);
	d.i32("algmode_bnd", algmode.bnd);
	d.i32("algmode_qck", algmode.qck);
	d.i32("algmode_lcl", algmode.lcl);
	d.f64("spb_fact", SpbFact);
	{
	    const Simmtx*	sm = pwd.simmtx;
	    d.i32("simmtx_dim", sm->dim);
	    d.i32("simmtx_rows", sm->rows);
	    std::vector<double> m((size_t) sm->rows * sm->dim);
	    for (int i = 0; i < sm->rows; ++i)
		for (int j = 0; j < sm->dim; ++j) m[(size_t) i * sm->dim + j] = sm->mtx[i][j];
	    d.put("simmtx", 3, m.data(), sm->rows, sm->dim);
	}
	if (pwd.wta) {std::vector<double> w(pwd.wta, pwd.wta + a->many); d.put("wta", 3, w.data(), a->many);}
	if (pwd.wtb) {std::vector<double> w(pwd.wtb, pwd.wtb + b->many); d.put("wtb", 3, w.data(), b->many);}

	WINDOW	wdw;
	stripe((const Seq**) sqs, &wdw, pwd.alnprm.sh);
	d.i32("wdw_lw", wdw.lw);
	d.i32("wdw_up", wdw.up);
	d.i32("wdw_width", wdw.width);

	dump_side(d, "a_", a);
	dump_side(d, "b_", b);

	if (a->left == a->right || b->left == b->right) return pwd.alnmode;

	// forward + raw VMF traceback exactly as alignC<recd_t> (fwd2c.h:671-677)
	VTYPE	scr = 0;
	SKL*	raw = 0;
	switch (pwd.alnmode) {
	    case NGP_ALB: raw = alignC<DPunit>(sqs, &pwd, &scr); break;
	    case HLF_ALB:
	    case RHF_ALB: raw = alignC<DPunit_hf>(sqs, &pwd, &scr); break;
	    case GPF_ALB: raw = alignC<DPunit_pf>(sqs, &pwd, &scr); break;
	    case NTV_ALB: raw = alignC<DPunit_nv>(sqs, &pwd, &scr); break;
	    case NGP_ALN: raw = alignC<DPunit>(sqs, &pwd, &scr, true); break;
	    case NTV_ALN: raw = alignC<DPunit_nv>(sqs, &pwd, &scr, true); break;
	    case HLF_ALN:
	    case RHF_ALN: raw = alignC<DPunit_hf>(sqs, &pwd, &scr, true); break;
	    case GPF_ALN: raw = alignC<DPunit_pf>(sqs, &pwd, &scr, true); break;
	    default: return -2;
	}
	d.f64("scr", scr);
	dump_skl(d, "vmf_trace", raw);		// end -> start, as Vmf::traceback returns it
	delete[] raw;

	// score-only entry (HomScore, maln2.cc:1837) for the rr[] contract
	long	rr[2] = {0, 0};
	VTYPE	hs = HomScore(sqs, &pwd, rr);
	d.f64("homscore", hs);
	int64_t	rr64[2] = {rr[0], rr[1]};
	d.put("homscore_rr", 2, rr64, 2);

	// the full operator: align2() incl. stdskl, end check/retry and fstat (maln2.cc:1875-1973)
	Gsinfo	gsi;
	VTYPE	scr2 = 0;
	SKL*	skl = align2(sqs, &pwd, &scr2, &gsi);
	d.f64("align2_scr", scr2);
	dump_skl(d, "align2_skl", skl);
	d.i32("align2_sh_after", pwd.alnprm.sh);
	d.f64("fstat_val", gsi.fstat.val);
	d.f64("fstat_mch", gsi.fstat.mch);
	d.f64("fstat_mmc", gsi.fstat.mmc);
	d.f64("fstat_gap", gsi.fstat.gap);
	d.f64("fstat_unp", gsi.fstat.unp);
	delete[] skl;
	return pwd.alnmode;
}

// Timed leg for bench.py's cpu_baseline (kind "reference"): ONLY the hot path -- alignC<recd_t> =
// Fwd2c ctor + forwardB + Vmf traceback (fwd2c.h:671-677) -- is inside the clock; PwdM (profiles, gap
// profiles) is built before it starts.  Returns seconds; *cells = in-band cells (fwd2c.h:373-374,393).
double ref_forward_timed(void* ga, void* gb, int64_t* cells, int* alnmode, double* score)
{
	mSeq*	sqs[3] = {(mSeq*) ga, (mSeq*) gb, 0};
	sqs[0]->exg_seq(sqs[0]->inex.exgl, sqs[0]->inex.exgr);
	sqs[1]->exg_seq(sqs[1]->inex.exgl, sqs[1]->inex.exgr);
	PwdM	pwd(sqs);
	mSeq*	a = sqs[0];
	mSeq*	b = sqs[1];
	if (alnmode) *alnmode = pwd.alnmode;
	WINDOW	wdw;
	stripe((const Seq**) sqs, &wdw, pwd.alnprm.sh);
	int64_t	c = 0;
	for (int m = a->left; m < a->right; ++m) {
	    int	n = std::max(m + wdw.lw, b->left);
	    int	n9 = std::min(m + wdw.up + 1, b->right);
	    if (n9 > n) c += n9 - n;
	}
	if (cells) *cells = c;
	VTYPE	scr = 0;
	SKL*	raw = 0;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	switch (pwd.alnmode) {
	    case NGP_ALB: raw = alignC<DPunit>(sqs, &pwd, &scr); break;
	    case HLF_ALB:
	    case RHF_ALB: raw = alignC<DPunit_hf>(sqs, &pwd, &scr); break;
	    case GPF_ALB: raw = alignC<DPunit_pf>(sqs, &pwd, &scr); break;
	    case NTV_ALB: raw = alignC<DPunit_nv>(sqs, &pwd, &scr); break;
	    default: return -1;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	delete[] raw;
	if (score) *score = scr;
	return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

// Whole-operator timing (PwdM + align2), kept for reference (kind "reference"): PwdM + align2 only, no dump.
// Returns the DP score; *cells receives the in-band cell count of forwardB (fwd2c.h:373-374,393).
double ref_align_timed(void* ga, void* gb, int64_t* cells, int* alnmode)
{
	mSeq*	sqs[3] = {(mSeq*) ga, (mSeq*) gb, 0};
	sqs[0]->exg_seq(sqs[0]->inex.exgl, sqs[0]->inex.exgr);
	sqs[1]->exg_seq(sqs[1]->inex.exgl, sqs[1]->inex.exgr);
	PwdM	pwd(sqs);
	mSeq*	a = sqs[0];
	mSeq*	b = sqs[1];
	if (alnmode) *alnmode = pwd.alnmode;
	WINDOW	wdw;
	stripe((const Seq**) sqs, &wdw, pwd.alnprm.sh);
	int64_t	c = 0;
	for (int m = a->left; m < a->right; ++m) {
	    int	n = std::max(m + wdw.lw, b->left);
	    int	n9 = std::min(m + wdw.up + 1, b->right);
	    if (n9 > n) c += n9 - n;
	}
	if (cells) *cells = c;
	Gsinfo	gsi;
	VTYPE	scr = 0;
	SKL*	skl = align2(sqs, &pwd, &scr, &gsi);
	delete[] skl;
	return scr;
}

// The whole operator for one pair: PwdM + align2 with a Gsinfo; returns the DP score, *val / *gap receive
// GsI->fstat.val / .gap (PreSpScore::calcSpScore, fspscore.cc:584-622) -- the sum-of-pairs score prrn's acceptance
// test reads.  Used by bench.py for the "SP-score delta vs ref" half of the metric.
double ref_align_fstat(void* ga, void* gb, double* val, double* gap)
{
	mSeq*	sqs[3] = {(mSeq*) ga, (mSeq*) gb, 0};
	sqs[0]->exg_seq(sqs[0]->inex.exgl, sqs[0]->inex.exgr);
	sqs[1]->exg_seq(sqs[1]->inex.exgl, sqs[1]->inex.exgr);
	PwdM	pwd(sqs);
	Gsinfo	gsi;
	VTYPE	scr = 0;
	SKL*	skl = align2(sqs, &pwd, &scr, &gsi);
	delete[] skl;
	if (val) *val = gsi.fstat.val;
	if (gap) *gap = gsi.fstat.gap;
	return scr;
}


// ---- f3: single sequences, the guide-tree DPs --------------------------------------------------------------------
void* ref_seq_read(const char* fname)
{
	Seq*	sd = new Seq(fname);
	if (!sd->many || !sd->len) {delete sd; return 0;}
	return sd;
}
void ref_seq_free(void* s) {delete (Seq*) s;}
int ref_seq_len(void* s) {return ((Seq*) s)->len;}
int ref_seq_range(void* s, int* left, int* right) {*left = ((Seq*) s)->left; *right = ((Seq*) s)->right; return ((Seq*) s)->many;}
void ref_seq_codes(void* s, unsigned char* out)
{
	Seq*	sd = (Seq*) s;
	for (int i = 0; i < sd->len; ++i) out[i] = *sd->at(i);
}
// the parameters Fwd2d / Aln2b1 read: alprm and the default similarity matrix
int ref_dist_params(double* uvst, int* sh, double* mtx, int cap, int* dim, int* rows)
{
	const Simmtx*	sm = getSimmtx(0);
	uvst[0] = alprm.u; uvst[1] = alprm.v; uvst[2] = alprm.scale; uvst[3] = alprm.tgapf;
	*sh = alprm.sh; *dim = sm->dim; *rows = sm->rows;
	if (sm->rows * sm->dim > cap) return -1;
	for (int i = 0; i < sm->rows; ++i)
	    for (int j = 0; j < sm->dim; ++j) mtx[i * sm->dim + j] = sm->mtx[i][j];
	return 0;
}
// alnScoreD(seqs, simmtx) global branch (fwd2d1.cc:324-338)
double ref_alnscored(void* sa, void* sb)
{
	const Seq*	sqs[2] = {(Seq*) sa, (Seq*) sb};
	return (double) alnScoreD(sqs, getSimmtx(0), 0);
}
double ref_selfalnscr(void* sa) {return (double) selfAlnScr((Seq*) sa, getSimmtx(0));}
// alnscore2dist (aln2.cc:289-333) as dpscore calls it for two single sequences (phyl.cc:222-252)
double ref_alnscore2dist(void* sa, void* sb, double denome)
{
	Seq*	sqs[2] = {(Seq*) sa, (Seq*) sb};
	PwdB	pwd((const Seq**) sqs);
	return (double) alnscore2dist(sqs, &pwd, 0, (FTYPE) denome);
}
// alignB_ng (fwd2b1.cc:1347-1353): skeleton corners into out[2 * k] (m, n); returns the number of corners, <0 on failure
int ref_alignb_ng(void* sa, void* sb, double* scr, int* out, int cap, double* pwdc)
{
	const Seq*	sqs[2] = {(Seq*) sa, (Seq*) sb};
	PwdB	pwd(sqs);
	VTYPE	s = 0;
	if (pwdc) {pwdc[0] = pwd.BasicGOP; pwdc[1] = pwd.BasicGEP; pwdc[2] = pwd.LongGOP; pwdc[3] = pwd.LongGEP; pwdc[4] = pwd.Noll; pwdc[5] = pwd.codonk1;}
	SKL*	skl = alignB_ng(sqs, &pwd, &s);
	*scr = (double) s;
	if (!skl) return -1;
	int	n = skl->n;
	if (n > cap) {delete[] skl; return -2;}
	for (int k = 0; k < n; ++k) {out[2 * k] = skl[k + 1].m; out[2 * k + 1] = skl[k + 1].n;}
	delete[] skl;
	return n;
}

}	// extern "C"
