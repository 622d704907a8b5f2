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


@pytest.mark.parametrize("case", ["prot", "dna", "ragged"])
def test_larger_families_vs_oracle(ctx, case):
    g = dict(np.load([p for p in GOLD if ("dna" in p) == (case == "dna")][0]))        # parameters + matrix of the molecule type
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
