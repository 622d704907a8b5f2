"""Edge cases of the DP path on the GPU: the smallest possible DPs (1 x 1 cells, single columns/rows), tiny and
ragged groups, heavy-indel families, and the error behaviour of the boundary (bad arguments are reported per
problem, the rest of the batch still runs).  Checker: the CPU oracle on the same flattened problem."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi, engine, operator as op, sweep
from prrn_aln_amd.synth import make_family, DNA

pytestmark = pytest.mark.gpu
GOLD = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))) if not os.path.basename(f).startswith("rect_")]   # (rect_*: the rectangular engine, tests/test_gpu_rect.py)


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


TINY = [
    ("2x1", dict(n_seq=2, length=1, seed=1), {}),
    ("3x1", dict(n_seq=3, length=1, seed=2), {}),
    ("2x3", dict(n_seq=2, length=3, seed=3), {}),
    ("5x2", dict(n_seq=5, length=2, seed=4, indel=0.3), {}),
    ("9x8_ls3", dict(n_seq=9, length=8, seed=5, indel=0.2), dict(ls=3)),
    ("24x3", dict(n_seq=24, length=3, seed=6, indel=0.25), {}),
    ("30x12_gappy", dict(n_seq=30, length=12, seed=7, indel=0.2, max_indel=6), {}),
    ("16x40_dna_ls3", dict(n_seq=16, length=40, seed=8, alphabet=DNA, indel=0.1, max_indel=8), dict(ls=3, molc=op.DNA, max_code=17)),
]


@pytest.mark.parametrize("name,fkw,akw", TINY, ids=[t[0] for t in TINY])
def test_tiny_and_ragged_families(ctx, name, fkw, akw):
    sw = None
    for bump in range(20):                             # (a leaf that lost every residue is not a valid input: next seed)
        fam = make_family(**dict(fkw, seed=fkw["seed"] + 100 * bump))
        if min(len(s.replace("-", "")) for s in fam.msa) == 0:
            continue
        sw = sweep.Sweep(fam, op.AlnParam(**akw))
        break
    assert sw is not None
    res = op.align2_batch(ctx, sw.pwds)
    L = oraclelib.load()
    for pw, (scr, skl, st) in zip(sw.pwds, res):
        assert st == 0, (name, pw.alnmode, st)

        class H:
            c = pw.problem
        oscr, ocells, otr = oraclelib.forward(L, H)
        assert scr == oscr, (name, pw.alnmode, scr, oscr)
        assert np.array_equal(skl, oraclelib.stdskl(L, otr))


def test_bad_arguments_are_reported_per_problem(ctx):
    """One broken problem does not poison the batch: it gets its own status, the others their results."""
    ds = [dict(np.load(f)) for f in GOLD[:3]]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    hs[1].c.lw, hs[1].c.up = 5, 4                      # empty band
    res = ctx.forward_batch(hs)
    assert res[1][3] != 0                              # G2G_ERR_ARG
    for i in (0, 2):
        scr, cells, tr, st = res[i]
        assert st == 0 and scr == ds[i]["scr"][0] and np.array_equal(tr, ds[i]["vmf_trace"])
    hs[1].c.lw, hs[1].c.up = int(ds[1]["wdw_lw"].ravel()[0]), int(ds[1]["wdw_up"].ravel()[0])
    hs[1].c.alnmode = 4                                # rectangular GPF_ALN: not on this path (tests/test_gpu_rect.py says why)
    res = ctx.forward_batch(hs)
    assert res[1][3] != 0 and res[0][3] == 0


def test_empty_batch(ctx):
    assert ctx.forward_batch([]) == []


def test_batches_are_cut_to_the_memory_budget(ctx, monkeypatch):
    """A tiny arena budget forces g2g_forward_batch to run the goldens in many chunks: same results."""
    ds = [dict(np.load(f)) for f in GOLD]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    monkeypatch.setenv("G2G_ARENA_LIMIT_GB", "0.002")
    res = ctx.forward_batch(hs)
    for d, (scr, cells, tr, st) in zip(ds, res):
        assert st == 0 and scr == d["scr"][0] and np.array_equal(tr, d["vmf_trace"])


def test_many_runs_of_one_batch(ctx):
    """A resident batch can be run any number of times: the sweep-mode progress counters carry an 11-bit generation,
    so run across the wrap (2048) and check that the results never change."""
    names = ["syn24x120_k3", "syn24x120_k1", "prot16x100_ls3_k2", "syn64x96_balanced"]
    ds = [dict(np.load(os.path.join(os.path.dirname(__file__), "golden", n + ".npz"))) for n in names]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    batch = ctx.prepare(hs)
    for it in range(2100):
        batch.run()
        if it in (0, 1, 2046, 2047, 2048, 2049, 2099):
            for d, (scr, cells, tr, st) in zip(ds, batch.fetch()):
                assert st == 0 and scr == d["scr"][0] and np.array_equal(tr, d["vmf_trace"]), it
    batch.free()


def test_injected_stall_costs_one_dp_and_is_recovered(ctx, monkeypatch):
    """A wait that can never be satisfied (test hook G2G_INJECT_STALL: the victim's first strip depends on a flag nobody
    writes) must run into the WALL-CLOCK limit, cost only that DP, and g2g_batch_run must re-run the DP on the non-polling
    kernel in the same call: every result of the batch, the victim's included, still equals the reference golden."""
    import glob, os, time
    from prrn_aln_amd import _abi
    gold = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "syn24x120_k*.npz")))
    ds = [dict(np.load(f)) for f in gold]
    ds = [d for d in ds if d["alnmode"][0] in (7, 8, 9)]
    assert len(ds) >= 4
    hs = [_abi.problem_from_arrays(d) for d in ds]
    for victim in (0, 2):
        before = ctx.counters()
        ctx.set_option("INJECT_STALL", victim)
        ctx.set_option("WAIT_LIMIT_MS", 300)
        t0 = time.time()
        res = ctx.forward_batch(hs)
        dt = time.time() - t0
        ctx.set_option("INJECT_STALL", None)
        ctx.set_option("WAIT_LIMIT_MS", None)
        assert 0.25 < dt < 20, dt                      # it did wait for the limit, and for not much longer
        for d, (scr, cells, tr, st) in zip(ds, res):
            assert st == 0 and scr == d["scr"][0] and np.array_equal(tr, d["vmf_trace"])
        # the event is VISIBLE: the context counts the waits that gave up and the DP that was re-run (g2g_ctx_counters; bench.py
        # and g2g_refine report the same counters for every ordinary run, where they must be zero)
        after = ctx.counters()
        # (victim 0 is also DP 0 of the one-DP retry batch, where the hook fires again: that DP is counted twice, the second
        # time as a DP that needed the non-polling kernel)
        assert after["wait_timeouts"] > before["wait_timeouts"]
        assert after["recovered_dps"] - before["recovered_dps"] == (2 if victim == 0 else 1)
        assert after["recovered_on_v1"] - before["recovered_on_v1"] == (1 if victim == 0 else 0)
    quiet = ctx.counters()
    res = ctx.forward_batch(hs)                        # and the context is fine afterwards
    assert all(st == 0 and scr == d["scr"][0] for d, (scr, cells, tr, st) in zip(ds, res))
    now = ctx.counters()
    assert now["wait_timeouts"] == quiet["wait_timeouts"] and now["recovered_dps"] == quiet["recovered_dps"] and now["runs"] == quiet["runs"] + 1


def test_cu_shares_do_not_change_results(ctx):
    """Option CU_SHARES: 2 forces a share of the CUs per persistent launch (hipExtStreamCreateWithCUMask) even on a run that does
    not fill the machine, 0 turns the shares off; the default decides by the run's size.  Same bits either way."""
    import glob, os
    from prrn_aln_amd import _abi
    gold = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))) if not os.path.basename(f).startswith("rect_")]
    ds = [dict(np.load(f)) for f in gold]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    for mode in ("2", "0"):
        ctx.set_option("CU_SHARES", mode)
        ctx.set_option("V6_MIN_STRIPS", 0)
        try:
            res = ctx.forward_batch(hs)
        finally:
            ctx.reset_options()
        for f, d, (scr, cells, tr, st) in zip(gold, ds, res):
            assert st == 0 and scr == d["scr"][0] and np.array_equal(tr, d["vmf_trace"]), (mode, os.path.basename(f))


def test_context_options(ctx, monkeypatch):
    """g2g_set_option is per context and beats the environment; NULL turns a switch off; reset goes back to the defaults"""
    monkeypatch.setenv("G2G_V3_COLS", "32")
    other = engine.Context(options={"V3_COLS": 128})
    try:
        assert ctx.get_option("V3_COLS") == "32" and other.get_option("G2G_V3_COLS") == "128"
        ctx.set_option("V3_COLS", None)
        assert ctx.get_option("V3_COLS") is None and other.get_option("V3_COLS") == "128"
        ctx.reset_options()
        assert ctx.get_option("V3_COLS") == "32"
    finally:
        other.close()
        ctx.reset_options()


def test_context_creation_leaves_the_hosts_rand_sequence_alone():
    """HIP's initialisation perturbs glibc's random() state; the reference seeds its division order from rand() (randiv.cc:41),
    so g2g_create must hand the state back untouched (fresh process: the runtime initialises only once)."""
    import subprocess, sys
    code = ("import ctypes, sys; libc = ctypes.CDLL('libc.so.6'); libc.srand(1); a = libc.rand();\n"
            "from prrn_aln_amd import engine; c = engine.Context(); b = libc.rand(); d = libc.rand(); c.close(); print(a, b, d)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    assert out.stdout.split() == ["1804289383", "846930886", "1681692777"], out.stdout
