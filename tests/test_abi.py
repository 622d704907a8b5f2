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
    assert L.g2g_abi_version() == 2


def test_struct_sizes_match_header(built):
    # guards the ctypes mirror against drift: compile a probe against the header
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "g2g.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(g2g_problem), sizeof(g2g_side), sizeof(g2g_gapprof), sizeof(g2g_result), sizeof(g2g_params));}'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "p"), os.path.join(td, "p.c")])
        out = subprocess.check_output([os.path.join(td, "p")]).decode().split()
    from prrn_aln_amd import _abi
    assert [int(x) for x in out] == [C.sizeof(_abi.Problem), C.sizeof(_abi.Side), C.sizeof(_abi.GapProf),
                                     C.sizeof(_abi.Result), C.sizeof(_abi.Params)]


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
