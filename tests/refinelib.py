"""Test-side helpers for the refinement loop (g2g_refine): loading the reference's trajectory fixtures, and a scorer callback
that puts the CPU CHECKER (oracle/) in the GPU's seat so that the C++ loop -- window logic, sharding, exchange, error paths --
runs on a machine without a GPU.  Tests only: the product never sets a scorer."""
from __future__ import annotations

import ctypes as C
import gzip
import json
import os

import numpy as np

import oraclelib
from prrn_aln_amd import _abi
from prrn_aln_amd import operator as op
from prrn_aln_amd._lib import lib
from prrn_aln_amd.refine import KTree

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures(small_only: bool = False):
    out = []
    for n in sorted(os.listdir(GOLD)):
        if n.startswith("refine_") and (n.endswith(".json") or n.endswith(".json.gz")):
            if small_only and ("48x300" in n or "256x1024" in n):
                continue
            out.append(os.path.join(GOLD, n))
    return out


def fixture_id(path: str) -> str:
    return os.path.basename(path)[7:].split(".json")[0]


def load(path: str):
    """(fixture dict, KTree, AlnParam, start codes)"""
    f = json.load(gzip.open(path, "rt") if path.endswith(".gz") else open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    return f, tree, alp, op.encode(f["rows"], f["molc"])


def oracle_scorer(fail_after: int = -1, record=None):
    """An _abi.SCORE_FN: forwardB + stdskl + calcSpScore of every PwdM of a window by the CPU checker.  fail_after >= 0: the
    call with that index (0-based) returns an error instead -- a rank-local failure for the exchange tests."""
    OL = oraclelib.load()
    L = lib()
    L.g2g_pwdm_spparams.argtypes = [C.c_void_p, C.POINTER(_abi.SpParams)]
    calls = [0]

    def cb(user, n, pw, cur, ncur, scr, skl, nskl, raw_cur, val_new):
        k = calls[0]
        calls[0] += 1
        if fail_after >= 0 and k >= fail_after:
            return 7
        try:
            for i in range(n):
                prob = L.g2g_pwdm_problem(pw[i])
                sp = _abi.SpParams()
                if L.g2g_pwdm_spparams(pw[i], C.byref(sp)) != 0:
                    return 2
                res = _abi.Result()
                if OL.g2g_oracle_forward(prob, C.byref(res)) != 0:
                    return 3
                nout = C.c_int(0)
                s = OL.g2g_oracle_stdskl(res.trace, res.ntrace, C.byref(nout))          # malloc'ed: handed to the library, which frees it
                OL.g2g_oracle_free(res.trace)
                out = (C.c_double * 3)()
                if OL.g2g_oracle_spscore(prob, C.byref(sp), cur[i], ncur[i], out) != 0:
                    return 4
                raw_cur[i] = out[2]
                if OL.g2g_oracle_spscore(prob, C.byref(sp), s, nout.value, out) != 0:
                    return 5
                val_new[i] = out[0]
                scr[i] = res.score
                skl[i] = s
                nskl[i] = nout.value
                if record is not None:
                    record.append(res.cells)
            return 0
        except Exception:
            return 9
    return _abi.SCORE_FN(cb)


def check_against_trace(f, final, steps, stats, molc_codes=None):
    """g2g_refine's outputs against the reference's trace: the branch sequence, (DP score, fstat.val) of every align2() call, the
    accepted moves (member lists and skeletons, when recorded) and the final MSA."""
    assert [s["branch"] for s in steps] == f["branches"]
    dps = [s for s in steps if not s["skipped"]]
    assert len(dps) == len(f["align2"])
    for s, ref in zip(dps, f["align2"]):
        assert (max(s["na"], s["nb"]), min(s["na"], s["nb"]), int(s["swp"])) == (max(ref["na"], ref["nb"]), min(ref["na"], ref["nb"]), ref["swp"])
        assert s["scr"] == ref["scr"] and s["val_new"] == ref["val"], (s["branch"], s["scr"], ref["scr"], s["val_new"], ref["val"])
    assert sum(1 for s in steps if s["accepted"]) == len(f["accepted"]) == stats["accepted"]
    if "moves" in stats:
        assert len(stats["moves"]) == len(f["accepted"])
        import zlib
        for (br, la, lb, skl), ref in zip(stats["moves"], f["accepted"]):
            assert lb == ref["lst1"], br
            if "lst0" in ref:
                assert la == ref["lst0"], br
            else:                                  # (the big trace keeps the smaller group only: the larger one is its complement)
                assert sorted(la + lb) == list(range(len(f["rows"]))), br
            if "skl" in ref:
                assert np.array_equal(skl, np.asarray(ref["skl"], np.int32)), br
            else:                                  # ... and a checksum of the skeleton instead of its corners
                assert len(skl) == ref["ncorners"] and zlib.crc32(np.ascontiguousarray(skl, np.int32).tobytes()) == ref["skl_crc32"], br
    assert np.array_equal(final, op.encode(f["final_rows"], f["molc"]))
