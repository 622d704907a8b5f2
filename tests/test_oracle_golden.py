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


BANDED = [f for f in GOLD if not os.path.basename(f).startswith("rect_")]      # (rect_*: the rectangular engine forwardA, `-A`)


@pytest.mark.parametrize("path", BANDED[::5], ids=[os.path.basename(p)[:-4] for p in BANDED[::5]])
def test_oracle_homscore(L, path):
    import ctypes as C
    d = dict(np.load(path))
    h = _abi.problem_from_arrays(d)
    rr = (C.c_long * 2)()
    s = L.g2g_oracle_homscore(C.byref(h.c), rr)
    assert s == d["homscore"][0]
    assert [rr[0], rr[1]] == d["homscore_rr"].tolist()


SP_GOLD = [f for f in GOLD if int(np.load(f)["alnmode"][0]) in (6, 7, 8, 9, 10)]     # Noll 2 and 3 (Gep1st); naive units


def sp_from_golden(d):
    return _abi.SpParams(float(d["Vab"][0]), float(d["BasicGEP"][0]), float(d["LongGEP"][0]) - float(d["BasicGEP"][0]),
                         float(d["diff_u"][0]))


@pytest.mark.parametrize("path", SP_GOLD, ids=[os.path.basename(p)[:-4] for p in SP_GOLD])
def test_oracle_spscore(L, path):
    """f1: SpScore::calcSkl + rescale along the reference's own skeleton == the reference's Gsinfo.fstat (bit-exact)."""
    d = dict(np.load(path))
    h = _abi.problem_from_arrays(d)
    rc, val, gap = oraclelib.spscore(L, h, sp_from_golden(d), d["align2_skl"])
    assert rc == 0
    assert val == d["fstat_val"][0] and gap == d["fstat_gap"][0]
    # the FSTAT counters (PwdM::stt?? / the naive calcstats): matched, mismatched, unpaired member pairs
    rc, val, gap, raw, mch, mmc, unp = oraclelib.spscore_stats(L, h, sp_from_golden(d), d["align2_skl"])
    assert rc == 0 and (mch, mmc, unp) == (d["fstat_mch"][0], d["fstat_mmc"][0], d["fstat_unp"][0])


def test_oracle_spscore_scope(L):
    """every reference golden pins calcSpScore: plain, half- and full-profile units with Noll 2 and 3, and the naive units
    SPunit_nv / _w11 / _w21 / _w22"""
    assert len(SP_GOLD) == len(BANDED) >= 59


def test_rectangular_goldens_present():
    """Fwd2c::forwardA (reference src/fwd2c.h:232-356): 21 goldens made with algmode.bnd = 0, every record type, Noll 2 and 3 --
    the restatement reproduces all of them (score and Vmf chain) INCLUDING the engines whose gap-state arrays the reference
    aliases at the start of every row (`*hdiag = *h`, fwd2c.h:247)"""
    rect = [f for f in GOLD if os.path.basename(f).startswith("rect_")]
    modes = {int(np.load(f)["alnmode"][0]) for f in rect}
    assert len(rect) >= 21 and modes == {1, 3, 4, 5}
