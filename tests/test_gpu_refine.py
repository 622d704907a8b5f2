"""f2 end to end on the GPU: prrn_aln_amd.refine.Refiner (divisions batched, DPs and calcSpScore on the GPU through the C
ABI) must walk the reference's own trajectory -- tests/golden/refine_*.json, traces of Prrn::rir taken by
oracle/_ref/prrn5_trace: the same branch sequence, bit-identical DP score and fstat.val at every align2(), the same accepted
moves (member lists and skeletons) and the same final MSA."""
import glob
import json
import os

import numpy as np
import pytest

from prrn_aln_amd import engine, operator as op
from prrn_aln_amd.refine import KTree, Refiner

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_*.json")))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_refinement_walks_the_reference_trajectory(ctx, path):
    f = json.load(open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    r = Refiner(ctx, op.encode(f["rows"], f["molc"]), tree, alp, seed=1, maxitr=10, window=16)
    final = r.run()
    steps = r.steps
    assert [s.branch for s in steps] == f["branches"][:len(steps)] and len(steps) == len(f["branches"])
    dps = [s for s in steps if s.delta != float("-inf")]            # (divisions with nothing to re-align make no align2 call)
    assert len(dps) == len(f["align2"])
    for s, ref in zip(dps, f["align2"]):
        assert (max(s.na, s.nb), min(s.na, s.nb), int(s.swp)) == (max(ref["na"], ref["nb"]), min(ref["na"], ref["nb"]), ref["swp"])
        assert s.scr == ref["scr"] and s.val_new == ref["val"], (s.branch, s.scr, ref["scr"], s.val_new, ref["val"])
    acc = [s for s in steps if s.accepted]
    assert len(acc) == len(f["accepted"])
    for s, ref in zip(acc, f["accepted"]):
        assert list(s.lst[0]) == ref["lst0"] and list(s.lst[1]) == ref["lst1"]
        assert np.array_equal(s.skl, np.asarray(ref["skl"], np.int32))
    assert np.array_equal(final, op.encode(f["final_rows"], f["molc"]))
    assert r.batches < len(dps)                                       # the DPs really ran in batches
