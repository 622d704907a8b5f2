"""The rectangular engine on the GPU: Fwd2c::forwardA (reference src/fwd2c.h:232-356; `-A` clears algmode.bnd, PwdM picks the _ALN
modes, align2 runs alignC<recd_t>(..., rectangle = true), src/maln2.cc:1906-1910).

* DPunit (NGP_ALN): g2g_forward_kernel in rect mode -- score, Vmf chain and standardised skeleton equal to the reference's goldens
  and to the oracle, through level 0 and through the operator (mSeq / PwdM / align2 with banded = 0).
* The gap-state record types (HLF / RHF / GPF / NTV_ALN) are refused with G2G_ERR_MODE, on purpose: the reference's forwardA starts
  every row with a struct assignment (`*hdiag = *h`, fwd2c.h:247) that makes the diagonal record share the left boundary's
  gap-state arrays; the restatement in oracle/ reproduces the resulting row-major in-place updates on all 21 goldens
  (tests/test_oracle_golden.py), a parallel sweep cannot."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi, engine, operator as op
from test_host_builders import groups_from_golden, params_from_golden

pytestmark = pytest.mark.gpu
RECT = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rect_*.npz")))
G2G_ERR_MODE = -2


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


def test_rectangular_goldens_level0(ctx):
    L = oraclelib.load()
    ds = [dict(np.load(f)) for f in RECT]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    res = ctx.forward_batch(hs)
    done = 0
    for f, d, h, (scr, cells, tr, st) in zip(RECT, ds, hs, res):
        name = os.path.basename(f)
        if int(d["alnmode"][0]) != 1:                       # gap-state engines in rectangular mode: refused, see the module text
            assert st == G2G_ERR_MODE, (name, st)
            continue
        assert st == 0, (name, st)
        oscr, ocells, otr = oraclelib.forward(L, h)
        assert scr == d["scr"][0] == oscr, (name, scr, float(d["scr"][0]), oscr)
        assert cells == ocells == (h.c.a.right - h.c.a.left) * (h.c.b.right - h.c.b.left), name
        assert np.array_equal(tr, d["vmf_trace"]), name
        assert np.array_equal(engine.stdskl(tr), d["align2_skl"]), name
        done += 1
    assert done >= 3


def test_rectangular_pairs_through_the_operator(ctx):
    """mSeq / PwdM / align2 with banded = 0: the goldens' pairs, and larger seeded pairs against the oracle"""
    L = oraclelib.load()
    for f in RECT:
        d = dict(np.load(f))
        if int(d["alnmode"][0]) != 1:
            continue
        alp = params_from_golden(d)
        assert alp.banded == 0
        pw = op.PwdM(list(groups_from_golden(d, alp)), alp)
        assert pw.alnmode == 1
        (scr, skl, st), = op.align2_batch(ctx, [pw])
        assert st == 0 and scr == d["align2_scr"][0] and np.array_equal(skl, d["align2_skl"]), os.path.basename(f)
    from prrn_aln_amd.synth import make_family
    alp = op.AlnParam(banded=0)
    alp3 = op.AlnParam(banded=0, ls=3)
    pws = []
    for seed in (51, 52, 53):
        fam = make_family(2, 700 + 150 * (seed - 51), seed, sub=0.3, indel=0.05, max_indel=30)
        rows = [r.replace("-", "") for r in fam.msa]
        for a in (alp, alp3):
            pws.append(op.PwdM([op.mSeq([rows[0]], a), op.mSeq([rows[1]], a)], a))
    res = op.align2_batch(ctx, pws)
    for pw, (scr, skl, st) in zip(pws, res):
        assert st == 0 and pw.alnmode == 1

        class H:
            c = pw.problem
        oscr, ocells, otr = oraclelib.forward(L, H)
        assert scr == oscr and np.array_equal(skl, oraclelib.stdskl(L, otr)), (pw.problem.noll, scr, oscr)
