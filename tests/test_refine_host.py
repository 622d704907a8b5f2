"""f2, host side (prrn_aln_amd/refine.py) against traces of the reference's own refinement loop (tests/golden/refine_*.json,
made by tools/make_refine_golden.py from oracle/_ref/prrn5_trace).  No GPU here: the generator, the tree weights, and the
matrix form of divideseq / synthgap (split_columns / join_columns) replayed over the moves the reference accepted."""
import ctypes
import glob
import json
import os

import numpy as np
import pytest

from prrn_aln_amd import operator as op
from prrn_aln_amd.refine import GlibcRand, KTree, TreeDivisions, join_columns, lt0, split_columns

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_*.json")))


def _load(path):
    f = json.load(open(path))
    t = f["tree"]
    return f, KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])


def test_glibc_rand_emulation():
    try:
        libc = ctypes.CDLL("libc.so.6")
    except OSError:
        pytest.skip("no glibc")
    g = GlibcRand()
    for seed in (1, 2, 1804289383, 123456789, 3000000000):
        libc.srand(seed); g.srand(seed)
        assert [libc.rand() for _ in range(50)] == [g.rand() for _ in range(50)]


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_branch_sequence_is_the_references(path):
    f, tree = _load(path)
    td = TreeDivisions(tree, 1)
    assert td.cycle == f["cycle"]
    assert [td.next() for _ in f["branches"]] == f["branches"]


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_replay_of_accepted_moves_gives_the_final_msa(path):
    """split_columns / join_columns = delcommongap + gather / synthgap in matrix form: applying the skeletons the reference
    accepted, in order, to the start MSA must end in the reference's final MSA."""
    f, tree = _load(path)
    codes = op.encode(f["rows"], f["molc"])
    td = TreeDivisions(tree, 1)
    for mv in f["accepted"]:
        a, b, skl0 = split_columns(codes, mv["lst0"], mv["lst1"])
        skl = np.asarray(mv["skl"], np.int32)
        assert tuple(skl[0]) == (0, 0) and tuple(skl[-1]) == (len(a), len(b))
        assert tuple(skl0[0]) == (0, 0) and tuple(skl0[-1]) == (len(a), len(b))
        # the current alignment read off the matrix is itself a valid skeleton: applying it changes nothing
        assert np.array_equal(join_columns(a, b, skl0, mv["lst0"], mv["lst1"], codes.shape[1]), codes)
        codes = join_columns(a, b, skl, mv["lst0"], mv["lst1"], codes.shape[1])
        assert not (codes == 1).all(axis=1).any()           # no all-gap column appears
    assert np.array_equal(codes, op.encode(f["final_rows"], f["molc"]))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_member_lists_and_weights(path):
    f, tree = _load(path)
    td = TreeDivisions(tree, 1)
    n = tree.n_leaves
    for t in range(td.cycle):
        la, lb = td.members(t)
        assert sorted(la + lb) == list(range(n)) and len(la) >= len(lb) and la == sorted(la) and lb == sorted(lb)
        pwt, w = tree.calcfact(t)
        assert pwt == tree.cur[t] and (w > 0).all()
    # accepted moves use exactly those lists
    seen = {(tuple(td.members(t)[0]), tuple(td.members(t)[1])) for t in range(td.cycle)}
    for mv in f["accepted"]:
        assert (tuple(mv["lst0"]), tuple(mv["lst1"])) in seen


def test_lt0():
    assert lt0(1e-3) and not lt0(0.0) and not lt0(5e-8) and not lt0(-1.0) and not lt0(float("-inf"))
