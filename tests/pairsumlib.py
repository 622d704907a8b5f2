"""helpers of the pairsum_ss tests: fixtures (tests/golden/pairsum, tools/make_pairsum_golden.py) -> ctypes structures"""
import ctypes as C

import numpy as np

from prrn_aln_amd import _abi, operator as op


def tree_struct(t):
    arr = lambda x, ty: np.ascontiguousarray(x, ty)
    keep = [arr(t["left"], np.int32), arr(t["right"], np.int32), arr(t["parent"], np.int32), arr(t["vol"], np.float64), arr(t["cur"], np.float64)]
    T = _abi.Tree()
    T.n_nodes = len(keep[0])
    i32p = C.POINTER(C.c_int32)
    T.left, T.right, T.parent = (k.ctypes.data_as(i32p) for k in keep[:3])
    T.vol, T.cur = keep[3].ctypes.data_as(_abi.c_f64p), keep[4].ctypes.data_as(_abi.c_f64p)
    T._keep = keep
    return T


def case_codes(case):
    """rows of a fixture case are strings of 'A' + residue code, one per member -> (len, many) uint8"""
    return np.array([[ord(ch) - 65 for ch in r] for r in case["rows"]], np.uint8).T.copy()


def alp_of(f):
    return op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
