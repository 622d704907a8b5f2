"""ctypes mirror of include/g2g.h (the C ABI of libg2g.so).  Plumbing only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)


class Skl(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32)]


class GapProf(C.Structure):
    _fields_ = [("hetero", C.c_int32),
                ("off", c_i32p * 3), ("glen", c_i32p * 3), ("freq", c_f64p * 3)]


class Side(C.Structure):
    _fields_ = [("many", C.c_int32), ("len", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
                ("nils", C.c_int32), ("dels", C.c_int32),
                ("seq", c_u8p), ("weight", c_f64p),
                ("nelm", C.c_int32), ("felm", C.c_int32),
                ("pseq", c_f64p), ("thk", c_f64p),
                ("has_gfq", C.c_int32), ("gfq", GapProf),
                ("gapdens", c_f64p), ("postgapdens", c_f64p),
                ("npfq", C.c_int32), ("pfq_step", C.c_int32), ("pfq_pos", C.POINTER(C.c_int32)), ("pfq_dns", c_f64p),
                ("sumwt", C.c_double),
                ("dev", C.c_void_p)]                # ABI 5: device-resident twins (g2g_side_dev), NULL for host-only sides


class Problem(C.Structure):
    _fields_ = [("alnmode", C.c_int32), ("sim2_kind", C.c_int32), ("noll", C.c_int32),
                ("codonk1", C.c_int32), ("lw", C.c_int32), ("up", C.c_int32),
                ("crg2_kind", C.c_int32), ("dvsp", C.c_int32),
                ("basic_gop", C.c_double), ("weighted_gop", C.c_double), ("u", C.c_double),
                ("u2divu1", C.c_double), ("v2divv1", C.c_double),
                ("simmtx", c_f64p), ("simdim", C.c_int32), ("simrows", C.c_int32),
                ("a", Side), ("b", Side), ("spb_fact", C.c_double)]


class Result(C.Structure):
    _fields_ = [("score", C.c_double), ("cells", C.c_int64), ("ntrace", C.c_int32),
                ("status", C.c_int32), ("trace", C.POINTER(Skl)), ("rr", C.c_int64 * 2)]


class Params(C.Structure):
    _fields_ = [("u", C.c_double), ("v", C.c_double), ("u0", C.c_double), ("u1", C.c_double),
                ("tgapf", C.c_double), ("scale", C.c_double), ("gamma", C.c_double),
                ("k1", C.c_int32), ("ls", C.c_int32), ("sh", C.c_int32), ("banded", C.c_int32),
                ("molc", C.c_int32),
                ("simmtx", c_f64p), ("simdim", C.c_int32), ("simrows", C.c_int32),
                ("max_code", C.c_int32)]


class DSeq(C.Structure):
    """g2g_dseq: one single sequence of the guide-tree DPs (f3)"""
    _fields_ = [("res", c_u8p), ("len", C.c_int32), ("left", C.c_int32), ("right", C.c_int32)]


def dseq(codes: np.ndarray) -> "DSeq":
    x = np.ascontiguousarray(codes, np.uint8)
    d = DSeq()
    d.res = x.ctypes.data_as(c_u8p)
    d.len, d.left, d.right = len(x), 0, len(x)
    d._keep = x
    return d


class Tree(C.Structure):
    """g2g_tree: the weighting tree (Ktree::lead[]) of g2g_refine"""
    _fields_ = [("n_nodes", C.c_int32), ("left", C.POINTER(C.c_int32)), ("right", C.POINTER(C.c_int32)),
                ("parent", C.POINTER(C.c_int32)), ("vol", c_f64p), ("cur", c_f64p)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32))


SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(Skl)), C.POINTER(C.c_int),
                       C.POINTER(C.c_double), C.POINTER(C.POINTER(Skl)), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double))


ACCEPT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(Skl))


class RefineOpts(C.Structure):
    _fields_ = [("seed", C.c_int32), ("maxitr", C.c_int32), ("window", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("slot_cap", C.c_int32), ("exchange", EXCHANGE_FN), ("exchange_user", C.c_void_p),
                ("scorer", SCORE_FN), ("scorer_user", C.c_void_p), ("on_accept", ACCEPT_FN), ("on_accept_user", C.c_void_p),
                ("window_min", C.c_int32), ("reserved", C.c_int32)]


class RefineStep(C.Structure):
    _fields_ = [("branch", C.c_int32), ("na", C.c_int32), ("nb", C.c_int32), ("swp", C.c_int32), ("accepted", C.c_int32),
                ("skipped", C.c_int32), ("scr", C.c_double), ("val_new", C.c_double), ("val_old", C.c_double), ("delta", C.c_double),
                ("t_ms", C.c_double)]


class RefineStats(C.Structure):
    _fields_ = [("divisions", C.c_int32), ("accepted", C.c_int32), ("batches", C.c_int32), ("divisions_scored_here", C.c_int32),
                ("divisions_wasted", C.c_int32), ("wait_timeouts", C.c_int32), ("recovered_dps", C.c_int32), ("reserved", C.c_int32)]


def _ptr(arr: np.ndarray, typ):
    return arr.ctypes.data_as(typ)


class ProblemHolder:
    """Owns the numpy arrays a Problem points into."""

    def __init__(self):
        self.keep: List[np.ndarray] = []
        self.c = Problem()

    def arr(self, a, dtype):
        x = np.ascontiguousarray(a, dtype=dtype)
        self.keep.append(x)
        return x


def problem_from_arrays(d: Dict[str, np.ndarray]) -> ProblemHolder:
    """Build a g2g_problem from a dict of flattened arrays (the layout tests/refdump.py parses out of
    the reference dump and that tests/golden/*.npz store)."""
    h = ProblemHolder()
    p = h.c
    sc = lambda k: d[k].reshape(-1)[0]
    p.alnmode = int(sc("alnmode"))
    p.sim2_kind = int(sc("sim2_kind"))
    p.noll = int(sc("Noll"))
    p.codonk1 = int(sc("codonk1"))
    p.crg2_kind = int(sc("crg2_kind")) if "crg2_kind" in d else 0
    p.dvsp = int(sc("DvsP")) if "DvsP" in d else 3
    p.lw = int(sc("wdw_lw"))
    p.up = int(sc("wdw_up"))
    p.basic_gop = float(sc("Basic_GOP"))
    p.weighted_gop = float(sc("Weighted_GOP"))
    p.u = float(sc("alnprm_u"))
    bgep, bgop = float(sc("BasicGEP")), float(sc("BasicGOP"))
    # Fwd2c ctor, reference src/fwd2c.h:85-86
    p.u2divu1 = float(sc("LongGEP")) / bgep if bgep < 0 else 0.0
    p.v2divv1 = float(sc("LongGOP")) / bgop if bgop < 0 else 0.0
    sm = h.arr(d["simmtx"], np.float64)
    p.simmtx = _ptr(sm, c_f64p)
    p.simrows, p.simdim = sm.shape
    for pfx, side, wkey in (("a_", p.a, "wta"), ("b_", p.b, "wtb")):
        g = lambda k: d[pfx + k]
        side.many = int(g("many")[0]); side.len = int(g("len")[0])
        side.left = int(g("left")[0]); side.right = int(g("right")[0])
        side.nils = int(g("nils")[0]); side.dels = int(g("dels")[0])
        side.sumwt = float(g("sumwt")[0]) if (pfx + "sumwt") in d else float(side.many)
        side.seq = _ptr(h.arr(g("seq"), np.uint8), c_u8p)
        if wkey in d:
            side.weight = _ptr(h.arr(d[wkey], np.float64), c_f64p)
        elif pfx + "weight" in d:
            side.weight = _ptr(h.arr(g("weight"), np.float64), c_f64p)
        side.nelm = int(g("nelm")[0]); side.felm = int(g("felm")[0])
        if pfx + "pseq" in d:
            side.pseq = _ptr(h.arr(g("pseq"), np.float64), c_f64p)
        side.thk = _ptr(h.arr(g("thk"), np.float64), c_f64p)
        side.has_gfq = 1 if (pfx + "hetero") in d else 0
        if side.has_gfq:
            side.gfq.hetero = int(g("hetero")[0])
            for v, nm in enumerate(("sfq", "tfq", "rfq")):
                side.gfq.off[v] = _ptr(h.arr(g(nm + "_off"), np.int32), c_i32p)
                side.gfq.glen[v] = _ptr(h.arr(g(nm + "_glen"), np.int32), c_i32p)
                side.gfq.freq[v] = _ptr(h.arr(g(nm + "_freq"), np.float64), c_f64p)
        if pfx + "gapdens" in d:
            side.gapdens = _ptr(h.arr(g("gapdens"), np.float64), c_f64p)
            side.postgapdens = _ptr(h.arr(g("postgapdens"), np.float64), c_f64p)
        if pfx + "pfq_pos" in d and len(g("pfq_pos")):
            side.npfq = len(g("pfq_pos")); side.pfq_step = int(g("pfq_step")[0])
            side.pfq_pos = _ptr(h.arr(g("pfq_pos"), np.int32), c_i32p)
            side.pfq_dns = _ptr(h.arr(g("pfq_dns"), np.float64), c_f64p)
    p.spb_fact = float(sc("spb_fact")) if "spb_fact" in d else 0.0
    return h


class SpParams(C.Structure):
    _fields_ = [("vab", C.c_double), ("basic_gep", C.c_double), ("diffu", C.c_double), ("diff_u", C.c_double),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


class Fstat(C.Structure):
    _fields_ = [("val", C.c_double), ("gap", C.c_double), ("status", C.c_int32), ("reserved", C.c_int32), ("raw", C.c_double),
                ("mch", C.c_double), ("mmc", C.c_double), ("unp", C.c_double)]
