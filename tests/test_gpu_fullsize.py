"""Oracle parity at BASELINE sizes (not self-comparison): divisions of the 256 x 1024 aa bench family (configs[2]) and
DPs of configs[4]'s per-DP shape (DNA, -yl3 = Noll 3, 4096 nt, static gap-profile lists longer than 16 entries).

The oracle (oracle/g2g_oracle.c, pinned on the reference's goldens) fills a 6e6-cell DP in under a second, so full-size
parity is affordable for a handful of divisions: score bit for bit, traceback record by record, standardised skeleton."""
import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import DNA, make_family, tree_branches, tree_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


def _division(codes, weights, side, alp):
    a, b, ia, ib = sweep.division_groups(codes, side)
    ga, gb = op.mSeq(a, alp, weights[ia]), op.mSeq(b, alp, weights[ib])
    return (ga, gb), op.PwdM([ga, gb], alp)


def _maxlist(q):
    mx = 0
    for s in (q.a, q.b):
        if s.has_gfq:
            for v in range(3):
                off = np.ctypeslib.as_array(s.gfq.off[v], shape=(s.len + 2,))
                mx = max(mx, int(np.diff(off).max()))
    return mx


def _check_vs_oracle(ctx, pwds):
    """twice: as the engine would run so small a batch (its _pf DPs on v2: shortest critical path), and with v6 forced (what a
    full sweep uses)"""
    L = oraclelib.load()
    hs = []
    for pw in pwds:
        class H:
            c = pw.problem
        hs.append(H)
    want = [oraclelib.forward(L, H) for H in hs]
    for opts in ({}, {"V6_MIN_STRIPS": 0}):
        ctx.reset_options()
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            res = op.align2_batch(ctx, pwds)
            raw = ctx.forward_batch(hs)
        finally:
            ctx.reset_options()
        for pw, (oscr, ocells, otr), (scr, skl, st), (rscr, rcells, rtr, rst) in zip(pwds, want, res, raw):
            assert st == 0 and rst == 0
            assert rcells == ocells
            assert scr == oscr and rscr == oscr, (opts, pw.alnmode, scr, oscr)
            assert np.array_equal(rtr, otr), (opts, pw.alnmode)
            assert np.array_equal(skl, oraclelib.stdskl(L, otr))


def test_bench_family_divisions_vs_oracle(ctx):
    """256 proteins x 1024 aa (the bench workload): smallest, median and largest division of BOTH engines
    (HLF/RHF = DPunit_hf, GPF = DPunit_pf), 3.5e6 .. 1.2e7 cells each, in one batch."""
    fam = make_family(256, 1024, 1)
    alp = op.AlnParam()
    sw = sweep.Sweep(fam, alp, weighted=True)
    hf = [k for k in sw.order if sw.pwds[k].alnmode in (7, 8)]
    pf = [k for k in sw.order if sw.pwds[k].alnmode == 9]
    assert len(hf) > 3 and len(pf) > 3
    pick = [hf[0], hf[len(hf) // 2], hf[-1], pf[0], pf[1], pf[len(pf) // 2], pf[-1]]
    assert min(sw.cells[k] for k in pick) > 1e6
    _check_vs_oracle(ctx, [sw.pwds[k] for k in pick])


def test_progressive_start_divisions_vs_oracle(ctx):
    """The same family from the REFERENCE's progressive MSA (tests/golden/msa/prog256x1024.npz: where prrn's refinement really
    starts): rougher than the true alignment, so the records' dynamic gap-state lists grow longer -- 0.06 % of the cells of
    these DPs read a list of eight entries or more, i.e. one that no longer fits its inline part in LDS: the scans, the long
    newdelta results and the strip hand-over of such lists (g2g_kernels_v6.hip, LS6) all run here, against the oracle."""
    import os
    fam = make_family(256, 1024, 1)
    alp = op.AlnParam()
    codes = np.load(os.path.join(os.path.dirname(__file__), "golden", "msa", "prog256x1024.npz"))["codes"]
    sw = sweep.Sweep(fam, alp, weighted=True, codes=codes)
    pf = [k for k in sw.order if sw.pwds[k].alnmode == 9]
    pick = [pf[0], pf[2], pf[len(pf) // 3], pf[len(pf) // 2], pf[-1]]
    _check_vs_oracle(ctx, [sw.pwds[k] for k in pick])


def test_dna_ls3_4096nt_divisions_vs_oracle(ctx):
    """configs[4]'s per-DP shape: DNA family of 4096-nt sequences, double-affine penalty (-yl3: Noll 3 kernels, codonk1 = 21),
    static gap-profile lists of more than 16 entries (beyond what the register-list kernel holds), 0.8-1.8e8 cells per DP."""
    n = 224
    fam = make_family(n, 4096, 2, alphabet=DNA, indel=0.012, max_indel=8)
    alp = op.AlnParam(ls=3, molc=op.DNA, max_code=17)
    codes = op.encode(fam.msa, alp.molc)
    w = np.asarray(tree_weights(fam.tree, n))
    br = sorted(tree_branches(fam.tree), key=lambda s: -min(len(s), n - len(s)))
    sides = [[s for s in br if min(len(s), n - len(s)) == 6][0], br[-1]]
    keep, pwds = [], []
    for s in sides:
        g, pw = _division(codes, w, s, alp)
        keep.append(g); pwds.append(pw)
    assert {pw.alnmode for pw in pwds} == {8, 9} or {pw.alnmode for pw in pwds} == {7, 9}
    assert all(pw.problem.noll == 3 and pw.problem.codonk1 == 21 for pw in pwds)
    assert max(_maxlist(pw.problem) for pw in pwds) > 16
    _check_vs_oracle(ctx, pwds)


def test_dna_ls3_resident_sweep_properties_and_sampled_parity(ctx):
    """BASELINE configs[4] as far as ONE GPU goes: a whole randiv sweep of a DNA family under the double-affine penalty (-yl3:
    Noll 3 kernels, codonk1 = 21) RESIDENT in HBM -- every tree-branch division of 160 sequences x 2048 nt in one batch (the full
    2048 x 4096 nt sweep is 4093 such divisions sharded over 8 GPUs; unmeasured on hardware).  Size-independent properties on
    EVERY division (a skeleton from (0, 0) to (ra, rb) made of diagonal and gap segments, the score of the sweep the same when
    run again), oracle parity on a sample across the size range."""
    n = 160
    fam = make_family(n, 2048, 2, alphabet=DNA, indel=0.012, max_indel=8)
    alp = op.AlnParam(ls=3, molc=op.DNA, max_code=17)
    sw = sweep.Sweep(fam, alp, weighted=True)
    assert len(sw) == 2 * n - 3 and all(pw.problem.noll == 3 for pw in sw.pwds)
    assert {pw.alnmode for pw in sw.pwds} >= {9} and sw.cells.sum() > 2e9

    class H:
        def __init__(self, q): self.c = q
    hs = [H(pw.problem) for pw in sw.pwds]
    b = ctx.prepare(hs)                                   # inputs of all 317 divisions resident
    b.run(); r1 = b.fetch()
    b.run(); r2 = b.fetch()
    assert b.recovery() == (0, 0, 0)
    b.free()
    from prrn_aln_amd import engine as eng
    for pw, (scr, cells, tr, st), (scr2, _, tr2, st2) in zip(sw.pwds, r1, r2):
        q = pw.problem
        assert st == 0 and st2 == 0 and scr == scr2 and np.array_equal(tr, tr2) and np.isfinite(scr)
        skl = eng.stdskl(tr)
        assert tuple(skl[0]) == (q.a.left, q.b.left) and tuple(skl[-1]) == (q.a.right, q.b.right)
        d = np.diff(skl, axis=0)
        assert (d >= 0).all() and ((d[:, 0] == d[:, 1]) | (d[:, 0] == 0) | (d[:, 1] == 0)).all()
    L = oraclelib.load()
    order = list(sw.order)
    for k in (order[0], order[len(order) // 2], order[-1]):
        oscr, ocells, otr = oraclelib.forward(L, hs[k])
        scr, cells, tr, st = r1[k]
        assert scr == oscr and cells == ocells and np.array_equal(tr, otr), (k, sw.pwds[k].alnmode, scr, oscr)
