"""CPU-side check of the reference-side binding (integration/g2g_bind.cc): without a GPU every align2() call must go to the
reference's own align2 and the wrapped programs must behave exactly like the unmodified ones.  (The GPU half of the
claim is tests/test_gpu_dropin.py.)  Skips when oracle/_ref was not built (no /root/reference)."""
import os
import re
import subprocess

import pytest

import refdump
from prrn_aln_amd.synth import make_family

REF = refdump.REF_DIR
BIN = {k: os.path.join(REF, k) for k in ("aln", "aln_g2g", "prrn5", "prrn5_g2g")}
pytestmark = pytest.mark.skipif(not all(os.path.exists(p) for p in BIN.values()) or not refdump.available(),
                                reason="oracle/_ref drop-in binaries not built (make -f oracle/Makefile.ref)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _run(exe, args, cwd, **env):
    e = dict(os.environ, ALN_TAB=os.path.join(REF, "table"), G2G_BIND_STATS="1", **env)
    p = subprocess.run([BIN[exe]] + args, cwd=cwd, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout, p.stderr


def test_binding_symbols_wrapped():
    """the wrapped programs import the C ABI from libg2g.so and define the wrapper of the mangled align2"""
    out = subprocess.run(["nm", "-D", "--undefined-only", BIN["prrn5_g2g"]], capture_output=True, text=True).stdout
    assert "g2g_forward_batch" in out and "g2g_create" in out
    syms = subprocess.run(["nm", BIN["prrn5_g2g"]], capture_output=True, text=True).stdout
    assert "__wrap__Z6align2PP4mSeqP4PwdMPdP6Gsinfo" in syms


@pytest.mark.parametrize("mode", ["off", None])
def test_prrn5_wrapped_equals_reference_on_cpu(tmp_path, mode):
    if mode is None and _gpu_present():
        pytest.skip("a GPU is present: covered by tests/test_gpu_dropin.py")
    tmp = str(tmp_path)
    fam = make_family(n_seq=12, length=80, seed=3, indel=0.03, max_indel=6)
    refdump.write_multi(os.path.join(tmp, "fam.msa"), ["s%02d" % i for i in range(12)], list(fam.msa), "fam")
    opts = ["-YH0", "-R1", "-O4", "fam.msa"]
    ref_out, _ = _run("prrn5", opts, tmp)
    env = {"G2G_BIND": mode} if mode else {}
    out, err = _run("prrn5_g2g", opts, tmp, **env)
    m = re.search(r"g2g_bind: (\d+) align2 calls, (\d+) on the GPU, (\d+) by the reference", err)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) == 0 and int(m.group(3)) == int(m.group(1)), err[-500:]
    assert out == ref_out


def test_aln_ce13a_pair_score():
    """BASELINE configs[0] (plumbing): `aln -s sample/pas ce13a1 ce13a2` through the wrapped program = the reference's output."""
    pas = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pas")
    ref_out, _ = _run("aln", ["-s", pas, "ce13a1", "ce13a2"], pas)
    out, err = _run("aln_g2g", ["-s", pas, "ce13a1", "ce13a2"], pas, G2G_BIND="off")
    assert "Score = 2325.0" in out and out == ref_out
