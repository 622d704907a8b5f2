"""Every forward-kernel generation and tile geometry against the reference goldens and the oracle.

The engine picks a kernel per problem (g2g_engine.hip: v3r = one lane per cell with the rows' static lists in
registers, v3 = the same with the lists in LDS, v6 = one lane per cell with rank-form merges (_pf), v2 = 8-lane teams
in 4-wave workgroups, v7 / v8 = one lane per cell for the records without gap profiles (DPunit, DPunit_nv), v1 = anti-diagonal sweep with the state in HBM).  The selection can be forced per context (g2g_set_option; the environment G2G_* supplies the defaults), read at batch-prepare time; each
forced configuration must reproduce the reference bit for bit.  Narrow tiles (G2G_V3_COLS) make even the small
golden DPs span several column blocks and strips."""
import glob
import os

import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import _abi, engine, operator as op, sweep
from prrn_aln_amd.synth import make_family

pytestmark = pytest.mark.gpu

GOLD = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))) if not os.path.basename(f).startswith("rect_")]   # (rect_*: the rectangular engine, tests/test_gpu_rect.py)
CONFIGS = {
    "default": {},
    "prologue_kernel": {"G2G_NO_CHAINQ": "1"},                       # boundary chains in their own kernel, lists staged in LDS
    "prologue_kernel_hbm": {"G2G_NO_CHAINQ": "1", "G2G_NO_PROSTAGE": "1"},   # ... lists straight from HBM
    "chainq_hbm": {"G2G_NO_PROSTAGE": "1"},                          # chains as queue entries, lists from HBM
    "v2_chainq_hbm": {"G2G_FORCE_V2": "1", "G2G_NO_PROSTAGE": "1"},
    "v3r_cols32": {"G2G_V3_COLS": "32", "G2G_V3_SWEEP": "0"},
    "v3r_tiles": {"G2G_V3_SWEEP": "0"},
    "v3lds_all": {"G2G_NO_AREG": "1"},
    "v3lds_all_cols32": {"G2G_NO_AREG": "1", "G2G_V3_COLS": "32", "G2G_V3_SWEEP": "0"},   # tile mode
    "no_v7": {"G2G_NO_V7": "1"},                                     # DPunit on v1 instead of the strip kernel
    "no_v8": {"G2G_NO_V8": "1"},                                     # DPunit_nv on v1 instead of the strip kernel
    "no_v6": {"G2G_NO_V6": "1"},                                     # _pf on the 8-lanes-per-cell kernel instead of v6
    "v6": {"G2G_V6_MIN_STRIPS": "0"},                                # v6 for _pf whatever the batch size (by default only batches that fill the GPU)
    "v6_publish4": {"G2G_V6_MIN_STRIPS": "0", "G2G_V2_SWEEP": "4"},  # progress counters published every 4 steps
    "v6_class_b": {"G2G_V6_MIN_STRIPS": "0", "G2G_V6_SMALL_KB": "53"},                         # the second footprint class (three strips per CU; off by default since v2 is faster there)
    "v6_class_c": {"G2G_V6_MIN_STRIPS": "0", "G2G_V6_LARGE_KB": "96"},                         # a third footprint class for v6 (off by default: slower than v2 there)
    "v2": {"G2G_FORCE_V2": "1"},
    "v2_t128": {"G2G_FORCE_V2": "1", "G2G_V2_THREADS": "128"},
    "v2_tiles": {"G2G_FORCE_V2": "1", "G2G_V2_SWEEP": "0"},
    "v2_tiles_t128": {"G2G_FORCE_V2": "1", "G2G_V2_SWEEP": "0", "G2G_V2_THREADS": "128", "G2G_V2_COLS": "64"},
    "v1": {"G2G_FORCE_V1": "1"},
}
ALLVARS = sorted({k for c in CONFIGS.values() for k in c})


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def L():
    return oraclelib.load()


@pytest.fixture(autouse=True)
def _fresh_options(ctx):
    ctx.reset_options()
    yield
    ctx.reset_options()


def _setenv(monkeypatch, cfg, ctx=None):
    """Force a configuration: through the context's own options (g2g_set_option) when a context is given -- the
    environment's G2G_* are cleared either way -- else through the environment defaults."""
    for k in ALLVARS:
        monkeypatch.delenv(k, raising=False)
    for k, v in cfg.items():
        if ctx is not None:
            ctx.set_option(k, v)
        else:
            monkeypatch.setenv(k, v)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_goldens_every_path(ctx, monkeypatch, name):
    _setenv(monkeypatch, CONFIGS[name], ctx)
    ds = [dict(np.load(f)) for f in GOLD]
    hs = [_abi.problem_from_arrays(d) for d in ds]
    res = ctx.forward_batch(hs)
    bad = []
    for f, d, (scr, cells, tr, st) in zip(GOLD, ds, res):
        if st != 0 or scr != d["scr"][0] or not np.array_equal(tr, d["vmf_trace"]):
            bad.append((os.path.basename(f), st, scr, float(d["scr"][0])))
    assert not bad, bad


FAMILIES = [
    ("prot_ls1", dict(n_seq=24, length=700, seed=11), dict()),
    ("prot_ls3", dict(n_seq=20, length=650, seed=12, max_indel=25), dict(ls=3)),
]


@pytest.mark.parametrize("name", ["default", "v6", "v6_publish4", "v6_class_c", "no_v6", "v3r_cols32", "v3r_tiles", "v3lds_all", "v2", "v2_t128", "v2_tiles", "v2_tiles_t128"])
@pytest.mark.parametrize("fam", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_large_divisions_every_path(ctx, L, monkeypatch, name, fam):
    """Group-vs-rest divisions of a 650-700 column family (many strips and blocks) per forced path."""
    _setenv(monkeypatch, CONFIGS[name])
    _, fkw, akw = fam
    sw = sweep.Sweep(make_family(**fkw), op.AlnParam(**akw), limit=10)
    res = op.align2_batch(ctx, sw.pwds)
    for pw, (scr, skl, st) in zip(sw.pwds, res):
        assert st == 0

        class H:
            c = pw.problem
        oscr, ocells, otr = oraclelib.forward(L, H)
        assert scr == oscr, (name, pw.alnmode, scr, oscr)
        assert np.array_equal(skl, oraclelib.stdskl(L, otr))


@pytest.mark.parametrize("name", ["default", "no_v8", "v6_publish4"])
def test_naive_mode_strips_vs_oracle(ctx, L, monkeypatch, name):
    """NTV_ALB (DPunit_nv: one gap length per member, PwdM::crg11 ... crg22w) on families of 3-6 sequences over several
    64-row strips, weighted and not, Noll 2 and 3: every division and every pairwise member count the mode is chosen for."""
    _setenv(monkeypatch, CONFIGS[name], ctx)
    seen = set()
    for n in (3, 4, 5, 6):
        for weighted in (True, False):
            for ls in (1, 3):
                fam = make_family(n, 260, 70 + n + ls, indel=0.05, max_indel=14)
                sw = sweep.Sweep(fam, op.AlnParam(ls=ls), weighted=weighted)
                res = op.align2_batch(ctx, sw.pwds)
                for pw, (scr, skl, st) in zip(sw.pwds, res):
                    assert st == 0, (n, weighted, ls, pw.alnmode)

                    class H:
                        c = pw.problem
                    oscr, ocells, otr = oraclelib.forward(L, H)
                    assert scr == oscr, (name, n, weighted, ls, pw.alnmode, pw.problem.crg2_kind, scr, oscr)
                    assert np.array_equal(skl, oraclelib.stdskl(L, otr))
                    if pw.alnmode == 10:
                        seen.add((pw.problem.crg2_kind, pw.problem.a.many, pw.problem.b.many, pw.problem.noll))
    kinds = {k for (k, _, _, _) in seen}
    assert kinds >= {121, 211, 221, 120, 210, 220} or len(kinds) >= 4, seen
