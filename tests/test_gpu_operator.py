"""The operator surface end to end on the GPU: groups -> PwdM (host builders) -> align2 (HIP DP +
traceback + stdskl + end check) against the reference's align2() results stored in the goldens, plus
size-independent properties at bench-like sizes."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi, engine, operator as op, sweep
from prrn_aln_amd.synth import make_family
from test_host_builders import groups_from_golden, params_from_golden

pytestmark = pytest.mark.gpu

# (level 1 builds its groups from residue codes only: the goldens made from intron-annotated inputs -- exon-boundary lists, the
#  bonus of fwd2c.h:446-452 -- are level-0 fixtures, tests/test_gpu_parity.py)
GOLD = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))) if not os.path.basename(f).startswith(("intron_", "rect_"))]


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


def test_align2_matches_reference_goldens(ctx):
    bad = []
    for f in GOLD:
        d = dict(np.load(f))
        alp = params_from_golden(d)
        ga, gb = groups_from_golden(d, alp)
        pw = op.PwdM([ga, gb], alp)
        scr, skl, st = op.align2(ctx, pw)
        if st != 0 or scr != d["align2_scr"][0] or not np.array_equal(skl, d["align2_skl"]):
            bad.append((os.path.basename(f), st, scr, float(d["align2_scr"][0])))
    assert not bad, bad


def test_homscore_matches_reference_goldens(ctx):
    bad = []
    for f in GOLD:
        d = dict(np.load(f))
        alp = params_from_golden(d)
        ga, gb = groups_from_golden(d, alp)
        pw = op.PwdM([ga, gb], alp)
        scr, rr = op.HomScore(ctx, pw)
        if scr != d["homscore"][0] or list(rr) != d["homscore_rr"].tolist():
            bad.append((os.path.basename(f), scr, float(d["homscore"][0]), rr, d["homscore_rr"].tolist()))
    assert not bad, bad


def test_calcspscore_matches_reference_goldens(ctx):
    """f1: PreSpScore::calcSpScore on the GPU, through level 1 (own PwdM scalars, own align2 skeleton), against the
    reference's Gsinfo.fstat -- bit-exact val and gap on every golden."""
    pws, want, names = [], [], []
    for f in GOLD:
        d = dict(np.load(f))
        alp = params_from_golden(d)
        ga, gb = groups_from_golden(d, alp)
        pws.append(op.PwdM([ga, gb], alp)); want.append(d); names.append(os.path.basename(f))
    res = op.align2_batch(ctx, pws)
    fs = op.calcSpScore_batch(ctx, pws, [skl for (_, skl, _) in res], stats=True)
    n_ok = 0
    for name, d, (val, gap, st, _raw, mch, mmc, unp) in zip(names, want, fs):
        # ... and the FSTAT counters mch / mmc / unp (PwdM::stt??, 14 variants, and the counters of the naive units)
        assert (mch, mmc, unp) == (d["fstat_mch"][0], d["fstat_mmc"][0], d["fstat_unp"][0]), (name, mch, mmc, unp)
        # every mode of the path: plain / half / full profile units (Noll 2, and Noll 3 with the Gep1st long-gap
        # bookkeeping) and the naive units SPunit_nv / _w11 / _w22 of the NTV modes
        assert st == 0, name
        assert val == d["fstat_val"][0] and gap == d["fstat_gap"][0], (name, val, float(d["fstat_val"][0]))
        n_ok += 1
    assert n_ok == len(GOLD) >= 59


@pytest.mark.parametrize("opts", [{"NO_SPLANES": "1"}, {"NO_SPSTREAM": "1"}], ids=["stream_scalar_walk", "unstreamed_walk"])
def test_calcspscore_walkers_agree(opts):
    """The three walkers of g2g_spscore_kernel -- lists across the lanes (default for half / full profile units without Gep1st),
    the scalar walk over the streamed inputs (its fallback: Noll 3, very long lists) and the unstreamed walk (no workspace) --
    give the reference's fstat (val, gap and the counters) on every golden and the checker's on a sweep of larger divisions."""
    c = engine.Context()
    for k, v in opts.items():
        c.set_option(k, v)
    try:
        pws, want = [], []
        for f in GOLD:
            d = dict(np.load(f))
            alp = params_from_golden(d)
            ga, gb = groups_from_golden(d, alp)
            pws.append(op.PwdM([ga, gb], alp)); want.append(d)
        res = op.align2_batch(c, pws)
        fs = op.calcSpScore_batch(c, pws, [skl for (_, skl, _) in res], stats=True)
        for d, f in zip(want, fs):
            assert f[2] == 0 and f[0] == d["fstat_val"][0] and f[1] == d["fstat_gap"][0]
            assert tuple(f[4:7]) == (d["fstat_mch"][0], d["fstat_mmc"][0], d["fstat_unp"][0])
        fam = make_family(40, 160, 23)
        sw = sweep.Sweep(fam, op.AlnParam())
        res = op.align2_batch(c, sw.pwds)
        fs = op.calcSpScore_batch(c, sw.pwds, [skl for (_, skl, _) in res])
        L = oraclelib.load()
        for pw, (scr, skl, st), (val, gap, fst, _raw) in zip(sw.pwds, res, fs):
            class H:
                c = pw.problem
            rc, oval, ogap = oraclelib.spscore(L, H, op.spparams(pw), skl)
            assert (fst == 0) == (rc == 0)
            if rc == 0:
                assert val == oval and gap == ogap
    finally:
        c.close()


@pytest.mark.parametrize("general", [False, True], ids=["one_resident_batch", "general_route"])
def test_align2_score_batch_equals_the_separate_calls(general):
    """g2g_align2_score_batch (what a window of g2g_refine calls: DPs + calcSpScore of the current and of the new alignment on
    one resident batch, both sets of walks in one launch) against g2g_align2_batch + g2g_spscore_batch on the same pairs -- the
    goldens (every mode) and a sweep of larger divisions whose 'current' alignments are the start MSA's own."""
    c = engine.Context()
    if general:
        c.set_option("NO_SCORE_BATCH", "1")
    try:
        pws = []
        for f in GOLD:
            d = dict(np.load(f))
            alp = params_from_golden(d)
            ga, gb = groups_from_golden(d, alp)
            pws.append(op.PwdM([ga, gb], alp))
        fam = make_family(24, 120, 5)
        sw = sweep.Sweep(fam, op.AlnParam())
        pws += list(sw.pwds)
        ref = op.align2_batch(c, pws)
        cur = [skl for (_, skl, _) in ref]                       # any valid skeleton serves as the 'current' one: the DP's own
        fs = op.calcSpScore_batch(c, pws, cur)
        got = op.align2_score_batch(c, pws, cur)
        assert len(got) == len(pws)
        for (scr, skl, st), f, (gscr, gskl, gst, fc, fn) in zip(ref, fs, got):
            assert gst == st and gscr == scr and np.array_equal(gskl, skl)
            assert fc == tuple(f) and fn == tuple(f)
    finally:
        c.close()


def test_calcspscore_sweep_vs_oracle(ctx):
    """... and on a sweep of larger divisions against the CPU restatement."""
    fam = make_family(40, 160, 23)
    sw = sweep.Sweep(fam, op.AlnParam())
    res = op.align2_batch(ctx, sw.pwds)
    fs = op.calcSpScore_batch(ctx, sw.pwds, [skl for (_, skl, _) in res])
    L = oraclelib.load()
    for pw, (scr, skl, st), (val, gap, fst, _raw) in zip(sw.pwds, res, fs):
        class H:
            c = pw.problem
        rc, oval, ogap = oraclelib.spscore(L, H, op.spparams(pw), skl)
        assert (fst == 0) == (rc == 0)
        if rc == 0:
            assert val == oval and gap == ogap


def test_calcspscore_noll3_sweep_vs_oracle(ctx):
    """... the same with -yl3 (Noll 3: Gep1st rings per member in an HBM workspace), DNA and protein."""
    from prrn_aln_amd.synth import DNA
    L = oraclelib.load()
    for fam, alp in ((make_family(24, 120, 31, alphabet=DNA, indel=0.03, max_indel=30), op.AlnParam(ls=3, molc=op.DNA, max_code=17)),
                     (make_family(24, 120, 32, indel=0.03, max_indel=30), op.AlnParam(ls=3))):
        sw = sweep.Sweep(fam, alp)
        res = op.align2_batch(ctx, sw.pwds)
        fs = op.calcSpScore_batch(ctx, sw.pwds, [skl for (_, skl, _) in res])
        n = 0
        for pw, (scr, skl, st), (val, gap, fst, _raw) in zip(sw.pwds, res, fs):
            class H:
                c = pw.problem
            rc, oval, ogap = oraclelib.spscore(L, H, op.spparams(pw), skl)
            assert (fst == 0) == (rc == 0), (pw.alnmode, fst, rc)
            if rc == 0:
                assert val == oval and gap == ogap, (pw.alnmode, val, oval)
                n += 1
        assert n > 10


def test_calcspscore_tiny_families_vs_oracle(ctx):
    """the naive units (NTV modes: SPunit_nv / _w11 / _w21 / _w22) on families of 3-6 sequences, weighted or not, Noll 2 and 3"""
    L = oraclelib.load()
    seen = set()
    for n in (3, 4, 5, 6):
        for weighted in (True, False):
            for ls in (1, 3):
                fam = make_family(n, 70, 40 + n, indel=0.05, max_indel=12)
                if min(len(r.replace("-", "")) for r in fam.msa) == 0:
                    continue
                sw = sweep.Sweep(fam, op.AlnParam(ls=ls), weighted=weighted)
                res = op.align2_batch(ctx, sw.pwds)
                fs = op.calcSpScore_batch(ctx, sw.pwds, [skl for (_, skl, _) in res])
                for pw, (scr, skl, st), (val, gap, fst, _raw) in zip(sw.pwds, res, fs):
                    class H:
                        c = pw.problem
                    rc, oval, ogap = oraclelib.spscore(L, H, op.spparams(pw), skl)
                    assert st == 0 and fst == 0 and rc == 0, (n, weighted, ls, pw.alnmode, st, fst, rc)
                    assert val == oval and gap == ogap, (n, weighted, ls, pw.alnmode, val, oval)
                    seen.add((pw.alnmode, pw.problem.noll, weighted))
    assert {m for (m, _, _) in seen} >= {10}, seen


def test_sweep_batch_vs_oracle_and_properties(ctx):
    """A whole (small) sweep as one batch: every DP bit-equal to the oracle; skeleton properties hold."""
    fam = make_family(40, 160, 21)
    alp = op.AlnParam()
    sw = sweep.Sweep(fam, alp)
    res = op.align2_batch(ctx, sw.pwds)
    L = oraclelib.load()
    for pw, (scr, skl, st) in zip(sw.pwds, res):
        assert st == 0

        class H:
            c = pw.problem
        oscr, ocells, otr = oraclelib.forward(L, H)
        assert scr == oscr
        assert np.array_equal(skl, oraclelib.stdskl(L, otr))
        q = pw.problem
        # skeleton: starts at (left,left), ends at (right,right), monotone, each step diagonal or pure gap
        assert tuple(skl[0]) == (q.a.left, q.b.left) and tuple(skl[-1]) == (q.a.right, q.b.right)
        dm, dn = np.diff(skl[:, 0]), np.diff(skl[:, 1])
        assert (dm >= 0).all() and (dn >= 0).all()
        assert ((dm == dn) | (dm == 0) | (dn == 0)).all()


def test_bench_size_dp_properties(ctx):
    """BASELINE-size DPs (256 x 1024 family): too slow for the oracle in a test, so check properties:
    identical replicas in one batch agree bit for bit, and the score does not depend on batch order."""
    fam = make_family(256, 1024, 1)
    alp = op.AlnParam()
    sw = sweep.Sweep(fam, alp, limit=6)
    r1 = op.align2_batch(ctx, sw.pwds)
    r2 = op.align2_batch(ctx, list(reversed(sw.pwds)))[::-1]
    r3 = op.align2_batch(ctx, [sw.pwds[0]] * 3)
    for (s1, k1, st1), (s2, k2, st2) in zip(r1, r2):
        assert st1 == 0 and st2 == 0
        assert s1 == s2 and np.array_equal(k1, k2)
    assert all(r[0] == r1[0][0] and np.array_equal(r[1], r1[0][1]) for r in r3)
    q = sw.pwds[0].problem
    assert tuple(r1[0][1][-1]) == (q.a.right, q.b.right)
