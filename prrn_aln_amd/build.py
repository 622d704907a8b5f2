"""Builds libg2g.so (hand-written HIP for gfx950 + the C-ABI host layer) in-tree with hipcc."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "g2g_engine.hip")]
EXTRA_CPP = [os.path.join(HERE, "csrc", f) for f in ("g2g_host.cpp", "g2g_refine.cpp")]
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("g2g_engine.hip", "g2g_kernels.hip", "g2g_kernels_v2.hip", "g2g_kernels_v3.hip", "g2g_kernels_v6.hip", "g2g_kernels_v7.hip", "g2g_kernels_v8.hip", "g2g_dist.hip", "g2g_pairaln.hip", "g2g_pairsum.hip", "g2g_device.h", "g2g_internal.h", "g2g_host.cpp", "g2g_refine.cpp")] + \
       [os.path.join(os.path.dirname(HERE), "include", "g2g.h")]
LIB = os.path.join(HERE, "libg2g.so")

# -ffp-contract=off: device AND host doubles must never be fused into FMAs (bit-exact parity with the
# reference CPU path, SURVEY.md §7 "Hard parts").
FLAGS = ["--offload-arch=gfx950", "-mllvm", "-amdgpu-promote-alloca-to-vector-limit=4096", "-DG2G_FWD_THREADS=512", "-DG2G_V2_THREADS=256", "-DG2G_V2_MINWAVES=3", "-DG2G_V2_TILE_COLS=512", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wno-unused-value", "-Wno-unused-result"]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = SRC + [f for f in EXTRA_CPP if os.path.exists(f)]
    cmd = [hipcc] + FLAGS + os.environ.get("G2G_EXTRA_FLAGS", "").split() + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
