"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/g2g.h declares, and its
host-only entry points (stdskl) agree with the reference goldens.  No GPU compute here."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from prrn_aln_amd import build
    return build.build_lib()


def test_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "g2g.h")).read()
    declared = set(re.findall(r"\b(g2g_[a-z0-9_]+)\s*\(", hdr))
    L = C.CDLL(built)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing


def test_abi_version(built):
    L = C.CDLL(built)
    assert L.g2g_abi_version() == 5


def test_struct_sizes_match_header(built):
    # guards the ctypes mirror against drift: compile a probe against the header
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "g2g.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(g2g_problem), sizeof(g2g_side), sizeof(g2g_gapprof), sizeof(g2g_result), sizeof(g2g_params)); printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(g2g_refine_opts), sizeof(g2g_refine_step), sizeof(g2g_refine_stats), sizeof(g2g_tree), sizeof(g2g_spparams), sizeof(g2g_fstat));}'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "p"), os.path.join(td, "p.c")])
        out = subprocess.check_output([os.path.join(td, "p")]).decode().split()
    from prrn_aln_amd import _abi
    assert [int(x) for x in out] == [C.sizeof(_abi.Problem), C.sizeof(_abi.Side), C.sizeof(_abi.GapProf),
                                     C.sizeof(_abi.Result), C.sizeof(_abi.Params), C.sizeof(_abi.RefineOpts), C.sizeof(_abi.RefineStep),
                                     C.sizeof(_abi.RefineStats), C.sizeof(_abi.Tree), C.sizeof(_abi.SpParams), C.sizeof(_abi.Fstat)]


def test_stdskl_host_matches_reference(built):
    from prrn_aln_amd import engine
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        d = np.load(f)
        skl = engine.stdskl(d["vmf_trace"])
        assert np.array_equal(skl, d["align2_skl"]), f


def test_no_device_fails_loudly(built):
    """Without a GPU the product must refuse, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from prrn_aln_amd import engine
    from prrn_aln_amd._lib import G2GError
    with pytest.raises(G2GError):
        engine.Context()


def test_context_options_override_the_environment(built, monkeypatch):
    """g2g_set_option / g2g_get_option / g2g_reset_options: per-context switches, the environment supplies defaults only
    (no GPU needed: a context without a device still holds options)."""
    L = C.CDLL(built)
    L.g2g_create.restype = C.c_void_p
    L.g2g_get_option.restype = C.c_char_p
    L.g2g_get_option.argtypes = [C.c_void_p, C.c_char_p]
    L.g2g_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.g2g_reset_options.argtypes = [C.c_void_p]
    L.g2g_destroy.argtypes = [C.c_void_p]
    monkeypatch.setenv("G2G_V3_COLS", "48")
    assert L.g2g_get_option(None, b"V3_COLS") == b"48"                 # no context: the environment's default
    h = L.g2g_create(-1)
    if not h:
        pytest.skip("g2g_create gives no context without a device (the same test runs in tests/test_gpu_edges.py)")
    try:
        monkeypatch.delenv("G2G_V3_COLS", raising=False)
        assert L.g2g_get_option(h, b"V3_COLS") is None
        monkeypatch.setenv("G2G_V3_COLS", "32")
        assert L.g2g_get_option(h, b"V3_COLS") == b"32" and L.g2g_get_option(h, b"G2G_V3_COLS") == b"32"
        assert L.g2g_set_option(h, b"V3_COLS", b"64") == 0
        assert L.g2g_get_option(h, b"V3_COLS") == b"64"
        assert L.g2g_set_option(h, b"G2G_V3_COLS", None) == 0          # off for this context, whatever the environment says
        assert L.g2g_get_option(h, b"V3_COLS") is None
        L.g2g_reset_options(h)
        assert L.g2g_get_option(h, b"V3_COLS") == b"32"
        assert L.g2g_set_option(h, b"", b"1") != 0 and L.g2g_set_option(None, b"X", b"1") != 0
    finally:
        L.g2g_destroy(h)
