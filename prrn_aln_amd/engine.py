"""Thin Python glue over level 0 of the C ABI (include/g2g.h): run batches of flattened DP problems on
the GPU.  All compute happens in libg2g.so (HIP); this module only marshals."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _abi
from ._lib import G2GError, last_error, lib


class Context:
    def __init__(self, device: int = -1, options=None):
        L = lib()
        self._h = L.g2g_create(device)
        if not self._h:
            raise G2GError("g2g_create failed: " + last_error())
        if not L.g2g_device_ok(self._h):
            msg = last_error()
            L.g2g_destroy(self._h)
            self._h = None
            raise G2GError("no usable MI355X / gfx950 kernel image: " + msg)
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name: str, value=None):
        """g2g_set_option: a tuning / diagnostic switch of this context (None: off, whatever the environment says)"""
        v = None if value is None else str(value).encode()
        if lib().g2g_set_option(self._h, name.encode(), v) != 0:
            raise G2GError(last_error())

    def reset_options(self):
        lib().g2g_reset_options(self._h)

    def get_option(self, name: str):
        v = lib().g2g_get_option(self._h, name.encode())
        return None if v is None else v.decode()

    def counters(self) -> dict:
        """g2g_ctx_counters: batch runs, waits that ran into the wall-clock limit, DPs re-run after one, DPs that needed v1"""
        out = (C.c_longlong * 4)()
        lib().g2g_ctx_counters(self._h, out)
        return {"runs": out[0], "wait_timeouts": out[1], "recovered_dps": out[2], "recovered_on_v1": out[3]}

    def last_timeout(self) -> str:
        """g2g_ctx_last_timeout: the report of the last recovered time-out of the scheduler's waits ("" if none)"""
        return lib().g2g_ctx_last_timeout(self._h).decode()

    def wait_gaps(self):
        """g2g_ctx_wait_gaps: (gaps of more than 4 ms that waiting waves found in their own running time, the longest in ms)"""
        n = C.c_longlong(); ms = C.c_double()
        lib().g2g_ctx_wait_gaps(self._h, C.byref(n), C.byref(ms))
        return n.value, ms.value

    def mem_counters(self) -> dict:
        """g2g_ctx_mem_counters: hipMalloc / hipFree calls made for the context's device-memory pool, requests served from it,
        bytes free in it"""
        out = (C.c_longlong * 4)()
        lib().g2g_ctx_mem_counters(self._h, out)
        return {"hip_malloc": out[0], "hip_free": out[1], "pool_hits": out[2], "pool_free_bytes": out[3]}

    def close(self):
        if self._h:
            lib().g2g_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _pp(holders: Sequence[_abi.ProblemHolder]):
        n = len(holders)
        arr = (C.POINTER(_abi.Problem) * n)()
        for i, h in enumerate(holders):
            arr[i] = C.pointer(h.c)
        return arr

    @staticmethod
    def _results(res, n) -> List[Tuple[float, int, np.ndarray, int]]:
        L = lib()
        out = []
        for i in range(n):
            r = res[i]
            tr = np.zeros((r.ntrace, 2), np.int32)
            if r.ntrace and r.trace:
                buf = np.ctypeslib.as_array(C.cast(r.trace, C.POINTER(C.c_int32)), shape=(r.ntrace * 2,))
                tr[:] = buf.reshape(-1, 2)
                L.g2g_free(r.trace)
            out.append((r.score, r.cells, tr, r.status))
        return out

    def forward_batch(self, holders: Sequence[_abi.ProblemHolder]):
        """alignC<recd_t> for every problem: [(score, cells, raw traceback (end->start), status)]."""
        L = lib()
        n = len(holders)
        res = (_abi.Result * n)()
        rc = L.g2g_forward_batch(self._h, n, self._pp(holders), res)
        if rc:
            raise G2GError("g2g_forward_batch rc=%d: %s" % (rc, last_error()))
        return self._results(res, n)

    def prepare(self, holders: Sequence[_abi.ProblemHolder]) -> "Batch":
        L = lib()
        h = C.c_void_p()
        rc = L.g2g_batch_prepare(self._h, len(holders), self._pp(holders), C.byref(h))
        if rc:
            raise G2GError("g2g_batch_prepare rc=%d: %s" % (rc, last_error()))
        return Batch(self, h, len(holders), keep=list(holders))


class Batch:
    """A sweep resident in HBM (inputs uploaded once); run() launches the kernels."""

    def __init__(self, ctx: Context, h, n: int, keep=None):
        self.ctx, self._h, self.n = ctx, h, n
        self._keep = keep                       # the library re-reads the problem descriptions when it re-runs a DP (g2g.h)

    def run(self):
        rc = lib().g2g_batch_run(self._h)
        if rc:
            raise G2GError("g2g_batch_run rc=%d: %s" % (rc, last_error()))

    def times_ms(self) -> Tuple[float, float]:
        a, b = C.c_float(0), C.c_float(0)
        lib().g2g_batch_times(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def cells(self) -> int:
        return lib().g2g_batch_cells(self._h)

    def recovery(self) -> Tuple[int, int, int]:
        """(waits that gave up in the last run, DPs re-run in the last run, DPs re-run over the batch's life)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().g2g_batch_recovery(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def arena_bytes(self) -> int:
        return lib().g2g_batch_arena_bytes(self._h)

    def fetch(self):
        res = (_abi.Result * self.n)()
        rc = lib().g2g_batch_fetch(self._h, res)
        if rc:
            raise G2GError("g2g_batch_fetch rc=%d: %s" % (rc, last_error()))
        return Context._results(res, self.n)

    def spscore(self, sps, skls):
        """f1 on the resident batch: PreSpScore::calcSpScore of problem i along skeleton skls[i] ((n,2) int arrays),
        sps[i] = _abi.SpParams.  Returns [(val, gap, status)]."""
        n = self.n
        sp = (_abi.SpParams * max(1, n))(*sps)
        bufs = []
        ptrs = (C.POINTER(_abi.Skl) * max(1, n))()
        cnt = (C.c_int * max(1, n))()
        for i, s in enumerate(skls):
            s = np.ascontiguousarray(s, np.int32).reshape(-1, 2)
            buf = (_abi.Skl * max(1, len(s)))()
            C.memmove(buf, s.ctypes.data, s.nbytes)
            bufs.append(buf)
            ptrs[i] = C.cast(buf, C.POINTER(_abi.Skl))
            cnt[i] = len(s)
        out = (_abi.Fstat * max(1, n))()
        rc = lib().g2g_batch_spscore(self._h, sp, ptrs, cnt, out)
        if rc:
            raise G2GError("g2g_batch_spscore rc=%d: %s" % (rc, last_error()))
        self.last_stats = [(out[i].mch, out[i].mmc, out[i].unp) for i in range(n)]      # FSTAT::mch / mmc / unp of the same call
        return [(out[i].val, out[i].gap, out[i].status) for i in range(n)]

    def free(self):
        if self._h:
            lib().g2g_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def stdskl(trace: np.ndarray) -> np.ndarray:
    """stdskl() of the C ABI (host code in libg2g.so), reference src/gaps.cc:139."""
    L = lib()
    t = np.ascontiguousarray(trace, np.int32)
    n = len(t)
    nout = C.c_int(0)
    p = L.g2g_stdskl(C.cast(t.ctypes.data, C.POINTER(_abi.Skl)), n, C.byref(nout))
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(nout.value * 2,)).reshape(-1, 2).copy() \
        if nout.value else np.zeros((0, 2), np.int32)
    L.g2g_free(p)
    return out


def process_counters() -> dict:
    """g2g_process_counters: the scheduler's counters over every context this process has had (destroyed ones included)"""
    out = (C.c_longlong * 8)()
    lib().g2g_process_counters(out)
    return {"runs": out[0], "wait_timeouts": out[1], "recovered_dps": out[2], "recovered_on_v1": out[3], "wait_gaps": out[4],
            "injected_timeouts": out[5], "injected_recovered_dps": out[6]}


def process_last_timeout() -> str:
    """the report of the process's last time-out that no test hook injected ("" if none)"""
    n = lib().g2g_process_last_timeout(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().g2g_process_last_timeout(buf, n + 1)
    return buf.value.decode()
