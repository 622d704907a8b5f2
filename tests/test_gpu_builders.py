"""SURVEY.md section 8 rows a8 / a9 on the DEVICE (csrc/g2g_build.hip through g2g_pwdm_create_batch): column thickness, frequency /
profile vectors and the three views of the static gap profile of every group, against the host builders of g2g_pwdm_create --
which tests/test_host_builders.py pins on the reference's own dumps (mSeq::mkthick src/mseq.cc:149-354, convseq :392-587,
Gfq::Gfq / seq2gfq src/gfreq.cc:134-312).  Exact equality of every array and scalar of the flattened problem; then the DP itself
on device-built inputs against the goldens."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import DNA, make_family
from test_host_builders import arr, groups_from_golden, params_from_golden

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


def same_problem(q, r, what):
    for f in ("alnmode", "sim2_kind", "noll", "codonk1", "lw", "up", "crg2_kind", "dvsp", "basic_gop", "weighted_gop", "u", "u2divu1", "v2divv1"):
        assert getattr(q, f) == getattr(r, f), (what, f)
    for side in ("a", "b"):
        s, t = getattr(q, side), getattr(r, side)
        for f in ("many", "len", "left", "right", "nils", "dels", "nelm", "felm", "has_gfq", "sumwt"):
            assert getattr(s, f) == getattr(t, f), (what, side, f)
        n, many = s.len, s.many
        assert np.array_equal(arr(s.seq, (n + 2) * many, np.uint8), arr(t.seq, (n + 2) * many, np.uint8)), (what, side, "seq")
        assert np.array_equal(arr(s.thk, (n + 2) * 3, np.float64), arr(t.thk, (n + 2) * 3, np.float64)), (what, side, "thk")
        assert bool(s.pseq) == bool(t.pseq), (what, side)
        if s.pseq:
            a, b = arr(s.pseq, (n + 2) * s.nelm, np.float64), arr(t.pseq, (n + 2) * t.nelm, np.float64)
            assert np.array_equal(a, b), (what, side, "pseq", np.flatnonzero(a != b)[:8])
        if s.has_gfq:
            assert s.gfq.hetero == t.gfq.hetero, (what, side, "hetero")
            for v in range(3):
                o1, o2 = arr(s.gfq.off[v], n + 2, np.int32), arr(t.gfq.off[v], n + 2, np.int32)
                assert np.array_equal(o1, o2), (what, side, "off", v, np.flatnonzero(o1 != o2)[:8])
                assert np.array_equal(arr(s.gfq.glen[v], o1[-1], np.int32), arr(t.gfq.glen[v], o1[-1], np.int32)), (what, side, "glen", v)
                f1, f2 = arr(s.gfq.freq[v], o1[-1], np.float64), arr(t.gfq.freq[v], o1[-1], np.float64)
                assert np.array_equal(f1, f2), (what, side, "freq", v, np.flatnonzero(f1 != f2)[:8])


def test_device_builders_equal_host_builders_on_the_goldens(ctx):
    """every reference golden's pair of groups, built on the device in ONE batch per parameter set"""
    sets = {}
    for path in GOLD:
        d = dict(np.load(path))
        alp = params_from_golden(d)
        key = (alp.molc, alp.ls, alp.tgapf, alp.sh, alp.banded, alp.u, alp.v, alp.u1, alp.k1, alp.scale, alp.max_code, alp.simmtx.tobytes())
        sets.setdefault(key, []).append((os.path.basename(path), alp, d))
    taken = 0
    for key, items in sets.items():
        alp = items[0][1]
        host, pairs = [], []
        for name, _, d in items:
            host.append(op.PwdM(list(groups_from_golden(d, alp)), alp))
            pairs.append(list(groups_from_golden(d, alp)))
        dev = op.PwdM.batch(ctx, pairs, alp)
        for (name, _, d), h, g in zip(items, host, dev):
            assert g.swp == h.swp, name
            same_problem(g.problem, h.problem, name)
            sh, sg = op.spparams(h), op.spparams(g)
            assert (sh.vab, sh.basic_gep, sh.diffu, sh.diff_u) == (sg.vab, sg.basic_gep, sg.diffu, sg.diff_u), name
        if alp.tgapf >= 1:
            taken += len(items)
    assert taken >= 40                   # (the tgapf < 1 goldens go through the host inside the same call)


@pytest.mark.parametrize("case", ["prot64x300", "dna40x400_ls3", "prot200x500"])
def test_device_builders_on_whole_sweeps(ctx, case):
    """all divisions of a family's sweep (a few hundred groups of 1 .. N-1 members, weighted): device batch == host threads,
    and the DPs on the device-built inputs give the host-built inputs' scores and skeletons"""
    if case == "dna40x400_ls3":
        fam, alp = make_family(40, 400, 7, alphabet=DNA, indel=0.03, max_indel=12), op.AlnParam(ls=3, molc=op.DNA, max_code=17)
    elif case == "prot64x300":
        fam, alp = make_family(64, 300, 5, indel=0.03, max_indel=10), op.AlnParam()
    else:
        fam, alp = make_family(200, 500, 9, indel=0.02, max_indel=8), op.AlnParam()
    host = sweep.Sweep(fam, alp, weighted=True)
    dev = sweep.Sweep(fam, alp, weighted=True, ctx=ctx)
    assert len(host) == len(dev) > 50
    for k, (h, g) in enumerate(zip(host.pwds, dev.pwds)):
        assert g.swp == h.swp
        same_problem(g.problem, h.problem, (case, k))
    pick = list(host.order[:: max(1, len(host) // 24)])
    rh = op.align2_batch(ctx, [host.pwds[k] for k in pick])
    rg = op.align2_batch(ctx, [dev.pwds[k] for k in pick])
    for (s1, k1, st1), (s2, k2, st2) in zip(rh, rg):
        assert st1 == 0 and st2 == 0 and s1 == s2 and np.array_equal(k1, k2)


def test_no_device_build_option_and_counters(ctx):
    fam, alp = make_family(24, 120, 3, indel=0.03), op.AlnParam()
    ctx.set_option("NO_DEVICE_BUILD", 1)
    try:
        a = sweep.Sweep(fam, alp, weighted=True, ctx=ctx)
    finally:
        ctx.reset_options()
    b = sweep.Sweep(fam, alp, weighted=True, ctx=ctx)
    for h, g in zip(a.pwds, b.pwds):
        same_problem(g.problem, h.problem, "option")
