// ref_trace.cc -- test infrastructure: a TRACE of the reference's own refinement loop (Prrn::rir, reference
// src/prrn5.cc:633-666), taken from outside by `ld --wrap` on four functions the loop calls in other translation units.
// Nothing of the reference is edited or copied; oracle/Makefile.ref links the reference's unmodified prrn5.cc with this
// file into oracle/_ref/prrn5_trace.  The trace (text, one record per line, file named by $G2G_TRACE) is what
// tools/make_refine_golden.py turns into tests/golden/refine_*.json:
//   T tid left right parent vol cur        one line per node of the Kirchhoff tree the loop divides on (Ktree::lead[])
//   C cycle nn                             Randiv::cycle, number of members
//   D rnbr                                 every branch Randiv::nextrandiv() hands out, in order
//   A na nb swp scr val gap Vab sumwt_a sumwt_b | names of a | names of b | weights of a | weights of b    every align2() of the loop
//   O val | skeleton                       every PreSpScore::calcSpScore(SKL*) of the loop: the CURRENT alignment of a division
//   S n0 n1 | lst0 | lst1 | skeleton       every accepted move (synthgap()): member lists and the skeleton applied
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aln.h"
#include "mseq.h"
#include "maln.h"
#include "gaps.h"
#include "mgaps.h"
#include "phyl.h"
#include "consreg.h"
#include "randiv.h"
#include "fspscore.h"

namespace {
FILE* trace_fd()
{
	static FILE*	fd = 0;
	static bool	tried = false;
	if (!tried) {
	    tried = true;
	    const char*	fn = getenv("G2G_TRACE");
	    if (fn) fd = fopen(fn, "w");
	}
	return fd;
}
}

extern "C" {
void __real__ZN6RandivC1EP3SeqP5Ssrel7DivModei(Randiv*, Seq*, Ssrel*, DivMode, int);
void __wrap__ZN6RandivC1EP3SeqP5Ssrel7DivModei(Randiv* self, Seq* sd, Ssrel* srl, DivMode dm, int rn)
{
	__real__ZN6RandivC1EP3SeqP5Ssrel7DivModei(self, sd, srl, dm, rn);
	FILE*	fd = trace_fd();
	if (!fd) return;
	const int	nn = srl->ss? srl->ss->num: sd->many;
	fprintf(fd, "C %ld %d\n", (long) self->cycle, nn);
	if (srl->ktree && srl->ktree->lead) {
	    Knode*	lead = srl->ktree->lead;
	    for (int k = 0; k < 2 * nn - 1; ++k) {
		Knode&	nd = lead[k];
		fprintf(fd, "T %d %d %d %d %.17g %.17g\n", nd.tid, nd.left? nd.left->tid: -1, nd.right? nd.right->tid: -1,
		    nd.parent? nd.parent->tid: -1, (double) nd.vol, (double) nd.cur);
	    }
	}
	fflush(fd);
}

LRAND __real__ZN6Randiv10nextrandivEv(Randiv*);
LRAND __wrap__ZN6Randiv10nextrandivEv(Randiv* self)
{
	LRAND	r = __real__ZN6Randiv10nextrandivEv(self);
	FILE*	fd = trace_fd();
	if (fd && self->mcr) fprintf(fd, "D %ld\n", (long) self->mcr->mcrand_now());
	return r;
}

SKL* __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI);
SKL* __wrap__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(mSeq* seqs[], PwdM* pwdm, VTYPE* scr, Gsinfo* GsI)
{
	SKL*	skl = __real__Z6align2PP4mSeqP4PwdMPdP6Gsinfo(seqs, pwdm, scr, GsI);
	FILE*	fd = trace_fd();
	if (fd && GsI) {
	    fprintf(fd, "A %d %d %d %.17g %.17g %.17g %.17g %.17g %.17g |", seqs[0]->many, seqs[1]->many, pwdm->swp? 1: 0, (double) *scr, (double) GsI->fstat.val,
		(double) GsI->fstat.gap, (double) pwdm->Vab, (double) seqs[0]->sumwt, (double) seqs[1]->sumwt);
	    for (int k = 0; k < 2; ++k) {
		for (int i = 0; i < seqs[k]->many; ++i) fprintf(fd, " %s", (*seqs[k]->sname)[i]);
		fprintf(fd, " |");
	    }
	    if (seqs[0]->weight) for (int i = 0; i < seqs[0]->many; ++i) fprintf(fd, " %.17g", (double) seqs[0]->weight[i]);
	    fprintf(fd, " |");
	    if (seqs[1]->weight) for (int i = 0; i < seqs[1]->many; ++i) fprintf(fd, " %.17g", (double) seqs[1]->weight[i]);
	    if (getenv("G2G_TRACE_GROUPS")) {			// debugging aid: the groups themselves
		for (int k = 0; k < 2; ++k) {
		    mSeq*	sd = seqs[k];
		    fprintf(fd, " | len %d left %d right %d exgl %d exgr %d nils %d dels %d ambs %d vect %d :", sd->len, sd->left, sd->right,
			(int) sd->inex.exgl, (int) sd->inex.exgr, (int) sd->inex.nils, (int) sd->inex.dels, (int) sd->inex.ambs, (int) sd->inex.vect);
		    for (int p = -1; p <= sd->len; ++p) { fputc(' ', fd); for (int i = 0; i < sd->many; ++i) fprintf(fd, "%x", (int) sd->at(p)[i]); }
		}
	    }
	    fprintf(fd, "\n");
	}
	return skl;
}

VTYPE __real__ZN10PreSpScore11calcSpScoreEP3SKL(PreSpScore* self, SKL* wsk);
VTYPE __wrap__ZN10PreSpScore11calcSpScoreEP3SKL(PreSpScore* self, SKL* wsk)
{
	const VTYPE	v = __real__ZN10PreSpScore11calcSpScoreEP3SKL(self, wsk);
	FILE*	fd = trace_fd();
	if (fd && wsk) {
	    fprintf(fd, "O %.17g |", (double) v);
	    for (int i = 1; i <= wsk->n; ++i) fprintf(fd, " %d %d", wsk[i].m, wsk[i].n);
	    fprintf(fd, "\n");
	}
	return v;
}

bool __real__Z8synthgapPP8GapsListP3SKLPPi(GapsList* glists[], SKL* skl, int* lst[]);
bool __wrap__Z8synthgapPP8GapsListP3SKLPPi(GapsList* glists[], SKL* skl, int* lst[])
{
	FILE*	fd = trace_fd();
	if (fd && skl) {
	    int	n0 = 0, n1 = 0;
	    while (lst[0][n0] >= 0) ++n0;
	    while (lst[1][n1] >= 0) ++n1;
	    fprintf(fd, "S %d %d |", n0, n1);
	    for (int i = 0; i < n0; ++i) fprintf(fd, " %d", lst[0][i]);
	    fprintf(fd, " |");
	    for (int i = 0; i < n1; ++i) fprintf(fd, " %d", lst[1][i]);
	    fprintf(fd, " |");
	    for (int i = 1; i <= skl->n; ++i) fprintf(fd, " %d %d", skl[i].m, skl[i].n);
	    fprintf(fd, "\n");
	}
	return __real__Z8synthgapPP8GapsListP3SKLPPi(glists, skl, lst);
}

// Ssrel::pairsum_ss (src/fspscore.cc:896-922): the whole-MSA sum-of-pairs score prrn reports -- "P <use_pw> <value> <many> <len>"
VTYPE __real__ZN5Ssrel10pairsum_ssEP4mSeqb(Ssrel* self, mSeq* sd, bool use_pw);
VTYPE __wrap__ZN5Ssrel10pairsum_ssEP4mSeqb(Ssrel* self, mSeq* sd, bool use_pw)
{
	const VTYPE	v = __real__ZN5Ssrel10pairsum_ssEP4mSeqb(self, sd, use_pw);
	FILE*	fd = trace_fd();
	if (fd) {
	    if (self->ktree && self->ktree->lead) {			// the tree THIS call used (the -O4 read-out builds a new Ssrel from the refined MSA)
		Knode*	lead = self->ktree->lead;
		for (int k = 0; k < 2 * sd->many - 1; ++k) {
		    Knode&	nd = lead[k];
		    fprintf(fd, "Q %d %d %d %d %.17g %.17g\n", nd.tid, nd.left? nd.left->tid: -1, nd.right? nd.right->tid: -1,
			nd.parent? nd.parent->tid: -1, (double) nd.vol, (double) nd.cur);
		}
	    }
	    fprintf(fd, "P %d %.17g %d %d |", use_pw? 1: 0, (double) v, sd->many, sd->len);
	    for (int m = 0; m < sd->many; ++m) {			// the MSA it was computed on, member by member (residue codes)
		fprintf(fd, " ");
		for (int i = 0; i < sd->len; ++i) fprintf(fd, "%c", 'A' + (int) sd->at(i)[m]);
	    }
	    fprintf(fd, "\n");
	}
	return v;
}

}
