"""The CPU restatement oracle (oracle/g2g_oracle.c) against golden vectors produced by the reference
itself (tools/make_golden.py -> tests/golden/*.npz).  Bit-exact: IEEE double score, integer traceback."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.fixture(scope="module")
def L():
    return oraclelib.load()


def test_goldens_present():
    assert len(GOLD) >= 40
    modes = {int(np.load(f)["alnmode"][0]) for f in GOLD}
    assert {6, 8, 9, 10} <= modes          # DPunit, _hf, _pf, _nv engines
    nolls = {int(np.load(f)["Noll"][0]) for f in GOLD}
    assert nolls == {2, 3}


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_reference(L, path):
    d = dict(np.load(path))
    h = _abi.problem_from_arrays(d)
    scr, cells, tr = oraclelib.forward(L, h)
    assert scr == d["scr"][0]                       # bit-exact double (fwd2c.h:481)
    assert np.array_equal(tr, d["vmf_trace"])       # Vmf::traceback order and coordinates
    skl = oraclelib.stdskl(L, tr)
    assert np.array_equal(skl, d["align2_skl"])     # stdskl corners == align2() result
    assert d["align2_scr"][0] == scr


@pytest.mark.parametrize("path", GOLD[::5], ids=[os.path.basename(p)[:-4] for p in GOLD[::5]])
def test_oracle_homscore(L, path):
    import ctypes as C
    d = dict(np.load(path))
    h = _abi.problem_from_arrays(d)
    rr = (C.c_long * 2)()
    s = L.g2g_oracle_homscore(C.byref(h.c), rr)
    assert s == d["homscore"][0]
    assert [rr[0], rr[1]] == d["homscore_rr"].tolist()
