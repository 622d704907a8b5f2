"""Builds libg2g.so (hand-written HIP for gfx950 + the C-ABI host layer) in-tree with hipcc.

The library is made of several translation units compiled side by side (csrc/g2g_device.h says which kernels each one emits):
a unit is recompiled only when one of the files it includes changed, so a host-side edit costs seconds and a kernel edit the
compile of its own unit."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libg2g.so")
HDR = [os.path.join(os.path.dirname(HERE), "include", "g2g.h"), os.path.join(CSRC, "g2g_device.h"), os.path.join(CSRC, "g2g_internal.h")]

_K1, _K2, _K3 = "g2g_kernels.hip", "g2g_kernels_v2.hip", "g2g_kernels_v3.hip"
# unit -> (source, files it includes besides the headers)
UNITS = {
    "engine": ("g2g_engine.hip", [_K1, _K2, _K3, "g2g_kernels_v6.hip", "g2g_kernels_v7.hip", "g2g_kernels_v8.hip", "g2g_dist.hip", "g2g_pairaln.hip", "g2g_pairsum.hip", "g2g_build.hip", "g2g_group.h"]),
    "v2": ("g2g_tu_v2.hip", [_K1, _K2]),
    "v3": ("g2g_tu_v3.hip", [_K1, _K2, _K3]),
    "v6": ("g2g_tu_v6.hip", [_K1, _K2, _K3, "g2g_kernels_v6.hip"]),
    "v78": ("g2g_tu_v78.hip", [_K1, _K2, _K3, "g2g_kernels_v6.hip", "g2g_kernels_v7.hip", "g2g_kernels_v8.hip"]),
    "host": ("g2g_host.cpp", ["g2g_group.h"]),
    "refine": ("g2g_refine.cpp", []),
}

# -ffp-contract=off: device AND host doubles must never be fused into FMAs (bit-exact parity with the
# reference CPU path, SURVEY.md §7 "Hard parts").
FLAGS = ["--offload-arch=gfx950", "-mllvm", "-amdgpu-promote-alloca-to-vector-limit=4096", "-DG2G_FWD_THREADS=512", "-DG2G_V2_THREADS=256", "-DG2G_V2_MINWAVES=3", "-DG2G_V2_TILE_COLS=512", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wno-unused-value", "-Wno-unused-result"]


def _extra():
    return os.environ.get("G2G_EXTRA_FLAGS", "").split()


def _deps(unit):
    src, inc = UNITS[unit]
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in inc] + HDR


def _obj(unit):
    return os.path.join(OBJ, unit + ".o")


def _flags_stamp():
    return os.path.join(OBJ, "flags.txt")


def _unit_stale(unit) -> bool:
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps(unit))


def _flags_changed() -> bool:
    stamp = " ".join(FLAGS + _extra())
    return not os.path.exists(_flags_stamp()) or open(_flags_stamp()).read() != stamp


def stale() -> bool:
    if not os.path.exists(LIB) or _flags_changed():     # (G2G_EXTRA_FLAGS=... python -m prrn_aln_amd.build must rebuild)
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for u in UNITS for d in _deps(u))


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    flags = FLAGS + _extra()
    stamp = " ".join(flags)
    if _flags_changed():
        force = True                                        # other flags than the objects were built with: everything again
    todo = [u for u in UNITS if force or _unit_stale(u)]

    def compile_unit(u):
        cmd = [hipcc] + flags + ["-c", "-o", _obj(u), os.path.join(CSRC, UNITS[u][0])]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return u, r.returncode, r.stdout + r.stderr

    jobs = int(os.environ.get("G2G_BUILD_JOBS", "0")) or min(len(todo) or 1, max(1, (os.cpu_count() or 2)))
    failed = []
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for u, rc, out in ex.map(compile_unit, todo):
            if rc != 0:
                failed.append(u)
                sys.stderr.write(out)
            elif verbose and out.strip():
                sys.stderr.write(out)
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc (units: %s)" % ", ".join(failed))
    open(_flags_stamp(), "w").write(stamp)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
