"""Test-side access to the real reference built under oracle/_ref (see oracle/Makefile.ref).

Only tests/ and tools/make_golden.py import this.  It is absent-tolerant: `available()` is False when
oracle/_ref/libprrn_ref.so has not been built (e.g. a checkout without /root/reference); tests that
need the live reference skip, the committed goldens under tests/golden/ still pin the oracle.
"""
from __future__ import annotations

import ctypes
import os
import struct
import tempfile
from typing import Dict, List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_SO = os.path.join(REF_DIR, "libprrn_ref.so")

PROTEIN, DNA = 1, 2          # reference src/cmn.h:107

_DT = {0: np.uint8, 1: np.int32, 2: np.int64, 3: np.float64}


def available() -> bool:
    return os.path.exists(REF_SO) and os.path.exists(os.path.join(REF_DIR, "table", "mdm_mtx"))


def parse_dump(path: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as fd:
        buf = fd.read()
    assert buf[:8] == b"G2GD0001", "bad dump magic"
    p = 8
    while p < len(buf):
        name = buf[p:p + 32].split(b"\0", 1)[0].decode()
        p += 32
        dtype, ndim = struct.unpack_from("<ii", buf, p)
        p += 8
        n0, n1 = struct.unpack_from("<qq", buf, p)
        p += 16
        dt = np.dtype(_DT[dtype])
        cnt = n0 * n1
        arr = np.frombuffer(buf, dtype=dt, count=cnt, offset=p).copy()
        p += cnt * dt.itemsize
        out[name] = arr.reshape(n0, n1) if ndim == 2 else arr
    return out


def write_multi(path: str, names: Sequence[str], rows: Sequence[str], title: str = "grp") -> None:
    """Write aligned rows in the reference's sequential multi-sequence format
    (the layout of reference sample/pas/Multi_A)."""
    with open(path, "w") as fd:
        if len(rows) == 1:      # a lone sequence: plain FASTA (the multi header would be read as residues)
            fd.write(">%s\n" % names[0])
            for i in range(0, len(rows[0]), 60):
                fd.write(rows[0][i:i + 60] + "\n")
            return
        fd.write("%5d %5d\t%s\n" % (len(rows), len(rows[0]), title))
        for nm, r in zip(names, rows):
            fd.write(">%s\n" % nm)
            for i in range(0, len(r), 60):
                fd.write(r[i:i + 60] + "\n")
            fd.write("/\n")


class RefLib:
    """One per process: the reference keeps its parameters in globals (alprm, algmode, simmtxes)."""

    def __init__(self, molc: int = PROTEIN, ls: int = 0, sh: int = 0, tgapf: Optional[float] = None,
                 band: Optional[bool] = None):
        if not available():
            raise RuntimeError("oracle/_ref is not built (make -f oracle/Makefile.ref)")
        os.environ["ALN_TAB"] = os.path.join(REF_DIR, "table")
        self.lib = ctypes.CDLL(REF_SO)
        L = self.lib
        L.ref_group_read.restype = ctypes.c_void_p
        L.ref_group_read.argtypes = [ctypes.c_char_p]
        L.ref_group_free.argtypes = [ctypes.c_void_p]
        L.ref_group_many.argtypes = [ctypes.c_void_p]
        L.ref_group_len.argtypes = [ctypes.c_void_p]
        L.ref_group_set_weight.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.ref_align_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        L.ref_align_timed.restype = ctypes.c_double
        L.ref_align_timed.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
        L.ref_forward_timed.restype = ctypes.c_double
        L.ref_forward_timed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
        L.ref_set_tgapf.argtypes = [ctypes.c_double]
        if tgapf is not None:
            L.ref_set_tgapf(tgapf)
        L.ref_init_prrn(molc, ls, sh)
        if band is not None:
            L.ref_set_band(1 if band else 0)

    def group(self, names, rows, weights=None, path=None):
        own = path is None
        if own:
            fd, path = tempfile.mkstemp(suffix=".mfa")
            os.close(fd)
        write_multi(path, names, rows)
        g = self.lib.ref_group_read(path.encode())
        if own:
            os.unlink(path)
        if not g:
            raise RuntimeError("reference reader rejected the group")
        if weights is not None:
            w = (ctypes.c_double * len(weights))(*weights)
            self.lib.ref_group_set_weight(g, w)
        return g

    def group_file(self, path, weights=None):
        g = self.lib.ref_group_read(path.encode())
        if not g:
            raise RuntimeError("reference reader rejected %s" % path)
        if weights is not None:
            w = (ctypes.c_double * len(weights))(*weights)
            self.lib.ref_group_set_weight(g, w)
        return g

    def free(self, g):
        self.lib.ref_group_free(g)

    def align_dump(self, ga, gb) -> Dict[str, np.ndarray]:
        fd, path = tempfile.mkstemp(suffix=".g2gd")
        os.close(fd)
        try:
            rc = self.lib.ref_align_dump(ga, gb, path.encode())
            if rc < 0:
                raise RuntimeError("ref_align_dump failed: %d" % rc)
            return parse_dump(path)
        finally:
            os.unlink(path)

    def forward_timed(self, ga, gb):
        """(seconds, cells, alnmode, score) of the reference's alignC<recd_t> alone."""
        cells = ctypes.c_int64(0)
        mode = ctypes.c_int(0)
        scr = ctypes.c_double(0)
        sec = self.lib.ref_forward_timed(ga, gb, ctypes.byref(cells), ctypes.byref(mode), ctypes.byref(scr))
        return sec, cells.value, mode.value, scr.value

    def align_fstat(self, ga, gb):
        """(DP score, fstat.val, fstat.gap) of the reference's align2 with a Gsinfo."""
        if not hasattr(self.lib, "ref_align_fstat"):
            raise RuntimeError("oracle/_ref predates ref_align_fstat: rebuild it")
        self.lib.ref_align_fstat.restype = ctypes.c_double
        self.lib.ref_align_fstat.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        val, gap = ctypes.c_double(0), ctypes.c_double(0)
        scr = self.lib.ref_align_fstat(ga, gb, ctypes.byref(val), ctypes.byref(gap))
        return scr, val.value, gap.value

    def align_timed(self, ga, gb):
        cells = ctypes.c_int64(0)
        mode = ctypes.c_int(0)
        scr = self.lib.ref_align_timed(ga, gb, ctypes.byref(cells), ctypes.byref(mode))
        return scr, cells.value, mode.value
