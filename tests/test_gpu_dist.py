"""f3 on the GPU: g2g_alnscored_batch (Fwd2d::forwardD, one wave per pair) against the reference's alnScoreD / dpscore values
of the committed fixtures, and against the CPU restatement on larger seeded families (ragged lengths, both state homes)."""
import glob
import os

import numpy as np
import pytest

import distlib
from prrn_aln_amd import engine, guide
from prrn_aln_amd.synth import make_family, DNA

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dist", "*.npz")))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.mark.parametrize("hbm", [False, True], ids=["lds", "hbm"])
def test_alnscored_matches_reference_goldens(ctx, hbm):
    ctx.reset_options()
    if hbm:
        ctx.set_option("DIST_HBM", 1)
    try:
        assert len(GOLD) >= 7
        for path in GOLD:
            d = dict(np.load(path))
            seqs = distlib.split(d)
            got, st = guide.alnscored_batch(ctx, distlib.params(d), seqs, d["ia"], d["ib"])
            assert (st == 0).all(), path
            assert np.array_equal(got, d["alnscored"]), (path, np.abs(got - d["alnscored"]).max())
    finally:
        ctx.reset_options()


def test_distance_matrix_matches_reference_dpscore(ctx):
    for path in GOLD:
        d = dict(np.load(path))
        seqs = distlib.split(d)
        dist = guide.distance_matrix(ctx, distlib.params(d), seqs, d["simmtx"])
        assert np.array_equal(dist, 100.0 * d["dist"]), path


@pytest.mark.parametrize("case", ["prot", "dna", "ragged", "ragged_tgapf05", "ragged_uv_tgapf03", "dna_tgapf05_ls3"])
def test_larger_families_vs_oracle(ctx, case):
    """(the last three take their parameters from the fixtures with discounted terminal gaps / non-dyadic penalties: walks of
    several hundred cells through dist_tail_walk and the accumulating branch of dist_edge, reference src/fwd2d1.cc:58-134)"""
    pick = {"ragged_tgapf05": "prot8_tgapf05", "ragged_uv_tgapf03": "prot8_u21_v93_tgapf03", "dna_tgapf05_ls3": "dna8_tgapf05_ls3"}.get(case)
    if pick:
        g = dict(np.load([p for p in GOLD if os.path.basename(p) == pick + ".npz"][0]))
        assert float(g["tgapf"][0]) < 1
        case = "dna" if "dna" in case else "ragged"
    else:
        g = dict(np.load([p for p in GOLD if ("dna" in p) == (case == "dna") and float(np.load(p)["tgapf"][0]) == 1 and float(np.load(p)["u"][0]) == 2][0]))        # parameters + matrix of the molecule type
    if case == "dna":
        fam = make_family(40, 700, 51, alphabet=DNA, indel=0.03, max_indel=30)
    else:
        fam = make_family(40, 600, 52, indel=0.04, max_indel=40)
    from prrn_aln_amd import operator as op
    rows = [r.replace("-", "") for r in fam.msa]
    if case == "ragged":
        rows = [r[: 30 + 14 * k] for k, r in enumerate(rows)]
    alp = op.AlnParam(molc=op.DNA, max_code=17) if case == "dna" else op.AlnParam()
    seqs = [op.encode([r], alp.molc)[:, 0].copy() for r in rows]
    ia, ib = guide.all_pairs(len(seqs))
    d = dict(g); d["ia"], d["ib"] = ia, ib
    got, st = guide.alnscored_batch(ctx, distlib.params(g), seqs, ia, ib)
    want = distlib.oracle_scores(d, seqs)
    assert (st == 0).all()
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_bad_sequences_fail_their_pairs_only(ctx):
    d = dict(np.load(GOLD[0]))
    seqs = distlib.split(d)
    seqs[2] = np.array([250, 3, 4], np.uint8)                      # a code outside the matrix
    got, st = guide.alnscored_batch(ctx, distlib.params(d), seqs, d["ia"], d["ib"])
    hit = (d["ia"] == 2) | (d["ib"] == 2)
    assert (st[hit] != 0).all() and (st[~hit] == 0).all()
    assert np.array_equal(got[~hit], d["alnscored"][~hit])
    got, st = guide.alnscored_batch(ctx, distlib.params(d), seqs, [], [])
    assert len(got) == 0


def _alp_params(d):
    p = distlib.params(d)
    p.k1, p.u1, p.molc = 7, 0.6, int(d["molc"][0])
    return p


@pytest.mark.parametrize("hbm", [False, True], ids=["lds", "hbm"])
def test_alignb_ng_matches_reference_goldens(ctx, hbm):
    """g2g_alignb_ng_batch: score and standardised skeleton of every pair equal the reference's alignB_ng (Noll 2 and 3,
    banded and not, terminal gaps discounted and not)"""
    ctx.reset_options()
    if hbm:
        ctx.set_option("DIST_HBM", 1)
    try:
        for path in GOLD:
            d = dict(np.load(path))
            seqs = distlib.split(d)
            res = guide.alignb_ng_batch(ctx, _alp_params(d), seqs, d["ia"], d["ib"])
            off = np.concatenate([[0], np.cumsum(d["alignb_nskl"])])
            for k, (scr, skl, st) in enumerate(res):
                assert st == 0, (path, k, st)
                assert scr == d["alignb_scr"][k], (path, k, scr, d["alignb_scr"][k])
                assert np.array_equal(skl, d["alignb_skl"][off[k]:off[k + 1]]), (path, k)
    finally:
        ctx.reset_options()


@pytest.mark.parametrize("ls", [1, 3])
def test_alignb_ng_larger_vs_oracle(ctx, ls):
    g = dict(np.load([p for p in GOLD if "prot12" in p][0]))
    g["ls"] = np.array([ls])
    from prrn_aln_amd import operator as op
    fam = make_family(16, 500, 61 + ls, indel=0.05, max_indel=60)
    rows = [r.replace("-", "")[: 200 + 25 * k] for k, r in enumerate(fam.msa)]
    seqs = [op.encode([r], op.PROTEIN)[:, 0].copy() for r in rows]
    ia, ib = guide.all_pairs(len(seqs))
    d = dict(g); d["ia"], d["ib"] = ia, ib
    got = guide.alignb_ng_batch(ctx, _alp_params(g), seqs, ia, ib)
    want = distlib.oracle_alignb(d, seqs)
    for k, ((scr, skl, st), (oscr, oskl, _, _)) in enumerate(zip(got, want)):
        assert st == 0 and scr == oscr, (k, st, scr, oscr)
        assert np.array_equal(skl, oskl), k


LSP = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lsp", "*.npz")))


@pytest.mark.parametrize("path", LSP, ids=[os.path.basename(p)[:-4] for p in LSP])
def test_lspb_ng_matches_reference_goldens(ctx, path):
    """The linear-space recursion on the GPU (g2g_centerb_kernel + g2g_centerb_pick_kernel + the traced leaves, driven level by
    level): score and standardised skeleton of every pair equal what the reference's alignB_ng returned with the same MaxVmfSpace
    -- lowered (option MAX_VMF_SPACE <-> setVmfSpace) so that small pairs recurse up to seven centres deep, and the default 16 Mi
    on 4500-nt DNA with -yl3 and a 6000-aa protein pair."""
    d = dict(np.load(path))
    seqs = distlib.split(d)
    ctx.reset_options()
    if int(d["maxvmf"][0]) != 16 * 1024 * 1024:
        ctx.set_option("MAX_VMF_SPACE", int(d["maxvmf"][0]))
    try:
        res = guide.alignb_ng_batch(ctx, _alp_params(d), seqs, d["ia"], d["ib"])
    finally:
        ctx.reset_options()
    off = np.concatenate([[0], np.cumsum(d["alignb_nskl"])])
    for k, (scr, skl, st) in enumerate(res):
        assert st == 0, (path, k, st)
        assert scr == d["alignb_scr"][k], (path, k, scr, d["alignb_scr"][k])
        assert np.array_equal(skl, d["alignb_skl"][off[k]:off[k + 1]]), (path, k)


@pytest.mark.parametrize("ls,threads", [(1, 64), (3, 256)])
def test_lspb_ng_larger_vs_oracle(ctx, ls, threads):
    """ragged seeded family, every pair through the recursion (MaxVmfSpace 20000 cells), against the CPU restatement"""
    g = dict(np.load([p for p in GOLD if "prot12" in p][0]))
    g["ls"] = np.array([ls])
    from prrn_aln_amd import operator as op
    fam = make_family(12, 700, 71 + ls, indel=0.05, max_indel=80)
    rows = [r.replace("-", "")[: 300 + 35 * k] for k, r in enumerate(fam.msa)]
    seqs = [op.encode([r], op.PROTEIN)[:, 0].copy() for r in rows]
    ia, ib = guide.all_pairs(len(seqs))
    d = dict(g); d["ia"], d["ib"] = ia, ib
    ctx.reset_options()
    ctx.set_option("MAX_VMF_SPACE", 20000)
    ctx.set_option("CENTER_THREADS", threads)
    try:
        got = guide.alignb_ng_batch(ctx, _alp_params(g), seqs, ia, ib)
    finally:
        ctx.reset_options()
    want = distlib.oracle_alignb(d, seqs, maxvmf=20000)
    assert sum(w[3] for w in want) > len(want)
    undefined = 0
    for k, ((scr, skl, st), (oscr, oskl, _, _)) in enumerate(zip(got, want)):
        if oscr is None:                                   # centerB_ng handed back a part that is no DP (-yl3: the reference itself crashes
            assert st == -2, (k, st)                       # on this pair): G2G_ERR_MODE, nothing read or written out of bounds
            undefined += 1
            continue
        assert st == 0 and scr == oscr, (k, st, scr, oscr)
        assert np.array_equal(skl, oskl), k
    assert undefined <= 2
