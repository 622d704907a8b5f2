"""f3, CPU side: the restatement of Fwd2d::forwardD (oracle/g2g_oracle.c: g2g_oracle_alnscored) against the reference's
alnScoreD on every pair of every committed fixture (tests/golden/dist, made by tools/make_dist_golden.py), and the host
mirror of alnscore2dist / dpscore (prrn_aln_amd/guide.py) against the reference's distances."""
import glob
import os

import numpy as np
import pytest

import distlib

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dist", "*.npz")))


def test_fixtures_present():
    assert len(GOLD) >= 12
    # the branches of Fwd2d's boundary / lastD (fwd2d1.cc:58-134) need fixtures of their own: discounted terminal gaps
    # (tgapf < 1, incl. 0) and penalties whose partial sums round
    tg = {os.path.basename(p): float(np.load(p)["tgapf"][0]) for p in GOLD}
    assert sum(1 for v in tg.values() if 0 < v < 1) >= 3 and any(v == 0 for v in tg.values())
    assert any(float(np.load(p)["u"][0]) not in (2.0,) for p in GOLD)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_alnscored_matches_reference(path):
    d = dict(np.load(path))
    seqs = distlib.split(d)
    got = distlib.oracle_scores(d, seqs)
    assert np.array_equal(got, d["alnscored"]), np.abs(got - d["alnscored"]).max()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_host_distance_formula_matches_reference(path):
    """alnscore2dist (aln2.cc:325-333) + dpscore's denominator (phyl.cc:229) from the reference's own scores"""
    from prrn_aln_amd import guide
    d = dict(np.load(path))
    seqs = distlib.split(d)
    selfs = np.array([guide.self_score(s, d["simmtx"]) for s in seqs])
    assert np.array_equal(selfs, d["selfscr"])
    got = guide.scores_to_dist(d["alnscored"], d["ia"], d["ib"], d["lens"], selfs, float(d["u"][0]))
    assert np.array_equal(got, d["dist"]), np.abs(got - d["dist"]).max()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_alignb_ng_matches_reference(path):
    """g2g_oracle_alignb_ng (forwardB_ng + initB_ng + lastB_ng + the Vmf chain + stdskl) against the reference's alignB_ng: score,
    skeleton and PwdB's constants for every pair"""
    d = dict(np.load(path))
    seqs = distlib.split(d)
    res = distlib.oracle_alignb(d, seqs)
    off = np.concatenate([[0], np.cumsum(d["alignb_nskl"])])
    for k, (scr, skl, pw, nc) in enumerate(res):
        assert scr == d["alignb_scr"][k], (k, scr, d["alignb_scr"][k])
        assert np.array_equal(skl, d["alignb_skl"][off[k]:off[k + 1]]), k
        assert pw == list(d["pwdb"]) and nc == 0


LSP = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lsp", "*.npz")))


def test_lsp_goldens_present():
    assert len(LSP) >= 8


@pytest.mark.parametrize("path", LSP, ids=[os.path.basename(p)[:-4] for p in LSP])
def test_oracle_lspb_ng_matches_reference(path):
    """The linear-space recursion (lspB_ng -> centerB_ng with binitB_ng / finitB_ng, the narrowed windows, diagonalB_ng, the traced
    leaves; fwd2b1.cc:382-782, 990-1095) against the reference's alignB_ng run with the same MaxVmfSpace: score and skeleton of every
    pair, and the recursion really ran (centerB_ng calls counted)."""
    d = dict(np.load(path))
    seqs = distlib.split(d)
    res = distlib.oracle_alignb(d, seqs, maxvmf=int(d["maxvmf"][0]))
    off = np.concatenate([[0], np.cumsum(d["alignb_nskl"])])
    centers = 0
    for k, (scr, skl, pw, nc) in enumerate(res):
        assert scr == d["alignb_scr"][k], (k, scr, d["alignb_scr"][k])
        assert np.array_equal(skl, d["alignb_skl"][off[k]:off[k + 1]]), k
        centers += nc
    assert centers >= len(res), centers
