"""f1, Ssrel::pairsum_ss (reference src/fspscore.cc:896-922), CPU side: the restatement of the naive branch (families of up to
ndesc_thr = 60 members: Ktree::recalcpw pair weights + Msap::ps_nml, with -yl3 the long-gap count) against values traced out of the
reference (tests/golden/pairsum, tools/make_pairsum_golden.py), weighted and unweighted, before and after a refinement pass."""
import ctypes as C
import glob
import json
import os

import pytest

import oraclelib
import pairsumlib
from prrn_aln_amd import _abi

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pairsum", "*.json")))


def test_fixtures_present():
    assert len(FIX) >= 6


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-5] for p in FIX])
def test_oracle_pairsum_small_trees(path):
    L = oraclelib.load()
    L.g2g_oracle_pairsum.argtypes = [C.POINTER(_abi.Params), C.c_int, C.c_int, _abi.c_u8p, C.POINTER(_abi.Tree), C.c_int, C.POINTER(C.c_double)]
    f = json.load(open(path))
    prm, _ = pairsumlib.alp_of(f).to_c()
    for c in f["cases"]:
        T = pairsumlib.tree_struct(c["tree"])
        codes = pairsumlib.case_codes(c)
        out = C.c_double()
        rc = L.g2g_oracle_pairsum(C.byref(prm), codes.shape[1], codes.shape[0], codes.ctypes.data_as(_abi.c_u8p), C.byref(T), c["use_pw"], C.byref(out))
        if codes.shape[1] > 60:
            assert rc != 0                          # the tree recursion is the product's (host builders + calcSpScore): tests/test_gpu_pairsum.py
            continue
        assert rc == 0 and out.value == c["value"], (f["name"], c["use_pw"], out.value, c["value"])
