"""The refinement trajectory of the reference (tests/golden/refine_*.json) re-walked on the CPU with the ORACLE doing the DPs
and the sum-of-pairs scores: host builders + oracle forwardB / stdskl / calcSpScore + the division logic of
prrn_aln_amd.refine must give bit-identical (DP score, fstat.val) at every align2() of the reference's loop, accept exactly
the moves it accepted and end in its final MSA.  (The GPU walks the same trajectory in tests/test_gpu_refine.py.)"""
import glob
import json
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import operator as op
from prrn_aln_amd.refine import KTree, TreeDivisions, join_columns, lt0, split_columns

FIX = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_*.json"))) if "48x300" not in p]      # (the 48 x 300 trace is the GPU tests': 930 DPs are minutes on the CPU oracle)


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_oracle_walks_the_reference_trajectory(path):
    L = oraclelib.load()
    f = json.load(open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    codes = op.encode(f["rows"], f["molc"])
    td = TreeDivisions(tree, 1)
    calls, acc = 0, 0
    for br in f["branches"]:
        assert td.next() == br
        la, lb = td.members(br)
        pwt, w = tree.calcfact(br)
        a, b, skl0 = split_columns(codes, la, lb)
        if len(a) == len(codes) and len(b) == len(codes):
            continue                                          # nothing to re-align: the reference makes no align2 call
        wa = w[la] if len(la) > 1 else np.ones(1)
        wb = w[lb] if len(lb) > 1 else np.ones(1)
        pw = op.PwdM([op.mSeq(a, alp, wa), op.mSeq(b, alp, wb)], alp)

        class H:
            c = pw.problem
        scr, cells, tr = oraclelib.forward(L, H)
        skl = oraclelib.stdskl(L, tr)
        old = skl0[:, ::-1].copy() if pw.swp else skl0
        sp = op.spparams(pw)
        rc0, _, _, raw_old = oraclelib.spscore_raw(L, H, sp, old)
        rc1, val_new, _ = oraclelib.spscore(L, H, sp, skl)
        assert rc0 == 0 and rc1 == 0
        ref = f["align2"][calls]
        calls += 1
        assert scr == ref["scr"] and val_new == ref["val"] and int(pw.swp) == ref["swp"], (br, scr, ref)
        same = skl.shape == old.shape and np.array_equal(skl, old)
        delta = 0.0 if same else pwt * (val_new - raw_old)    # Prrn::onecycle, src/prrn5.cc:535 (the old score is not rescaled)
        if lt0(delta):
            skl1 = skl[:, ::-1].copy() if pw.swp else skl
            mv = f["accepted"][acc]
            acc += 1
            assert mv["lst0"] == list(la) and mv["lst1"] == list(lb) and np.array_equal(np.asarray(mv["skl"]), skl1)
            codes = join_columns(a, b, skl1, la, lb, codes.shape[1])
    assert calls == len(f["align2"]) and acc == len(f["accepted"])
    assert np.array_equal(codes, op.encode(f["final_rows"], f["molc"]))
