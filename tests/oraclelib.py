"""ctypes access to the CPU restatement oracle (oracle/libg2g_oracle.so).  Tests only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from prrn_aln_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libg2g_oracle.so")


def load():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(SO)
    L.g2g_oracle_forward.argtypes = [C.POINTER(_abi.Problem), C.POINTER(_abi.Result)]
    L.g2g_oracle_forward_cells.argtypes = [C.POINTER(_abi.Problem), C.POINTER(_abi.Result),
                                           _abi.c_f64p, C.POINTER(C.c_int8)]
    L.g2g_oracle_homscore.restype = C.c_double
    L.g2g_oracle_homscore.argtypes = [C.POINTER(_abi.Problem), C.POINTER(C.c_long)]
    L.g2g_oracle_spscore.restype = C.c_int
    L.g2g_oracle_spscore.argtypes = [C.POINTER(_abi.Problem), C.POINTER(_abi.SpParams), C.POINTER(_abi.Skl), C.c_int,
                                     C.POINTER(C.c_double)]
    L.g2g_oracle_stdskl.restype = C.POINTER(_abi.Skl)
    L.g2g_oracle_stdskl.argtypes = [C.POINTER(_abi.Skl), C.c_int, C.POINTER(C.c_int)]
    L.g2g_oracle_free.argtypes = [C.c_void_p]
    return L


def skl_to_np(ptr, n) -> np.ndarray:
    out = np.zeros((n, 2), np.int32)
    for i in range(n):
        out[i, 0] = ptr[i].m
        out[i, 1] = ptr[i].n
    return out


def forward(L, holder):
    res = _abi.Result()
    rc = L.g2g_oracle_forward(C.byref(holder.c), C.byref(res))
    if rc:
        raise RuntimeError("oracle rc=%d" % rc)
    tr = skl_to_np(res.trace, res.ntrace)
    L.g2g_oracle_free(res.trace)
    return res.score, res.cells, tr


def stdskl(L, trace: np.ndarray) -> np.ndarray:
    n = len(trace)
    buf = (_abi.Skl * n)()
    for i in range(n):
        buf[i].m = int(trace[i, 0]); buf[i].n = int(trace[i, 1])
    nout = C.c_int(0)
    p = L.g2g_oracle_stdskl(buf, n, C.byref(nout))
    out = skl_to_np(p, nout.value)
    L.g2g_oracle_free(p)
    return out


def spscore(L, holder, sp: "_abi.SpParams", skl: np.ndarray):
    """SpScore::calcSkl on a standardised skeleton ((n,2) int array): (rc, val, gap)."""
    n = len(skl)
    buf = (_abi.Skl * n)()
    for i in range(n):
        buf[i].m, buf[i].n = int(skl[i][0]), int(skl[i][1])
    out = (C.c_double * 3)()
    rc = L.g2g_oracle_spscore(C.byref(holder.c), C.byref(sp), buf, n, out)
    return rc, out[0], out[1]


def spscore_raw(L, holder, sp: "_abi.SpParams", skl: np.ndarray):
    """the same, plus the score before PwdM::rescale (what PreSpScore::calcSpScore(SKL*) returns): (rc, val, gap, raw)"""
    n = len(skl)
    buf = (_abi.Skl * n)()
    for i in range(n):
        buf[i].m, buf[i].n = int(skl[i][0]), int(skl[i][1])
    out = (C.c_double * 3)()
    rc = L.g2g_oracle_spscore(C.byref(holder.c), C.byref(sp), buf, n, out)
    return rc, out[0], out[1], out[2]


def spscore_stats(L, holder, sp: "_abi.SpParams", skl: np.ndarray):
    """the same with the FSTAT counters: (rc, val, gap, raw, mch, mmc, unp)"""
    L.g2g_oracle_spscore6.restype = C.c_int
    L.g2g_oracle_spscore6.argtypes = [C.POINTER(_abi.Problem), C.POINTER(_abi.SpParams), C.POINTER(_abi.Skl), C.c_int, C.POINTER(C.c_double)]
    n = len(skl)
    buf = (_abi.Skl * n)()
    for i in range(n):
        buf[i].m, buf[i].n = int(skl[i][0]), int(skl[i][1])
    out = (C.c_double * 6)()
    rc = L.g2g_oracle_spscore6(C.byref(holder.c), C.byref(sp), buf, n, out)
    return (rc,) + tuple(out)
