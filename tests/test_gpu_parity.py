"""Parity of the HIP engine (through the C ABI, libg2g.so) against
  * the committed reference-generated goldens (bit-exact score, identical traceback records), and
  * the CPU restatement oracle on the same inputs.
Runs only on the GPU box (-m gpu)."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi, engine

pytestmark = pytest.mark.gpu

GOLD = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))) if not os.path.basename(f).startswith("rect_")]   # (rect_*: the rectangular engine, tests/test_gpu_rect.py)


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def L():
    return oraclelib.load()


def test_native_library_loaded():
    # the driver checks which .so files the test process mapped: make sure it is ours
    from prrn_aln_amd import _lib
    _lib.lib()
    with open("/proc/self/maps") as fd:
        assert "libg2g.so" in fd.read()


def test_goldens_one_batch(ctx, L):
    """All goldens as ONE batch (mixed engines, Noll 2 and 3) -- the shape a refinement sweep has."""
    ds = [dict(np.load(f)) for f in GOLD]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    res = ctx.forward_batch(hs)
    bad = []
    for f, d, h, (scr, cells, tr, st) in zip(GOLD, ds, hs, res):
        name = os.path.basename(f)
        if st != 0:
            bad.append((name, "status", st)); continue
        oscr, ocells, otr = oraclelib.forward(L, h)
        if scr != d["scr"][0] or scr != oscr:
            bad.append((name, "score", scr, float(d["scr"][0])))
        if cells != ocells:
            bad.append((name, "cells", cells, ocells))
        if not np.array_equal(tr, d["vmf_trace"]):
            bad.append((name, "trace"))
        skl = engine.stdskl(tr)
        if not np.array_equal(skl, d["align2_skl"]):
            bad.append((name, "skl"))
    assert not bad, bad


@pytest.mark.parametrize("path", GOLD[::7], ids=[os.path.basename(p)[:-4] for p in GOLD[::7]])
def test_golden_single(ctx, path):
    d = dict(np.load(path))
    h = _abi.problem_from_arrays(d)
    (scr, cells, tr, st), = ctx.forward_batch([h])
    assert st == 0
    assert scr == d["scr"][0]
    assert np.array_equal(tr, d["vmf_trace"])


def test_intron_position_bonus_goldens(ctx, L):
    """Inputs with exon-boundary annotations (reference sample/pas/ce13a*; BASELINE configs[0]'s pair among them): the bonus
    PfqItr::match_score adds in forwardB (fwd2c.h:446-452) and the Iiinfo term of calcSkl (gsinfo.cc:622-684), level 0 of the
    ABI: DP score, traceback and fstat.val / gap equal the reference's, and differ from a run with the annotations removed."""
    gold = [f for f in GOLD if os.path.basename(f).startswith("intron_")]
    assert len(gold) >= 7
    ds = [dict(np.load(f)) for f in gold]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    assert all(h.c.a.npfq > 0 and h.c.b.npfq > 0 and h.c.spb_fact > 0 for h in hs)
    batch = ctx.prepare(hs)
    batch.run()
    res = batch.fetch()
    for f, d, (scr, cells, tr, st) in zip(gold, ds, res):
        assert st == 0 and scr == d["scr"][0] and np.array_equal(tr, d["vmf_trace"]), (os.path.basename(f), st, scr, float(d["scr"][0]))
    sps = [_abi.SpParams(float(d["Vab"][0]), float(d["BasicGEP"][0]), float(d["LongGEP"][0]) - float(d["BasicGEP"][0]), float(d["diff_u"][0])) for d in ds]
    fs = batch.spscore(sps, [d["align2_skl"] for d in ds])
    for f, d, r, stt in zip(gold, ds, fs, batch.last_stats):
        assert r[2] == 0 and r[0] == d["fstat_val"][0] and r[1] == d["fstat_gap"][0], (os.path.basename(f), r, float(d["fstat_val"][0]))
        assert stt == (d["fstat_mch"][0], d["fstat_mmc"][0], d["fstat_unp"][0]), (os.path.basename(f), stt)
    batch.free()
    plain = []
    for d in ds:
        e = {k: v for k, v in d.items() if "pfq" not in k}
        plain.append(_abi.problem_from_arrays(e))
    res0 = ctx.forward_batch(plain)
    assert any(r0[0] != r[0] for r0, r in zip(res0, res))             # the annotation does change the score
