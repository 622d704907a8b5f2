"""helpers of the f3 tests: fixtures -> ctypes structures, the oracle's alnScoreD"""
import ctypes as C

import numpy as np

import oraclelib
from prrn_aln_amd import _abi


def split(d):
    off = np.concatenate([[0], np.cumsum(d["lens"])])
    return [np.ascontiguousarray(d["codes"][off[k]:off[k + 1]]) for k in range(len(d["lens"]))]


def params(d):
    p = _abi.Params()
    p.u, p.v, p.scale, p.tgapf = float(d["u"][0]), float(d["v"][0]), float(d["scale"][0]), float(d["tgapf"][0])
    p.sh = int(d["sh"][0])
    p.ls = int(d["ls"][0])
    m = np.ascontiguousarray(d["simmtx"], np.float64)
    p.simmtx = m.ctypes.data_as(_abi.c_f64p)
    p.simrows, p.simdim = m.shape
    p._keep = m
    return p


def oracle_scores(d, seqs):
    L = oraclelib.load()
    L.g2g_oracle_alnscored.argtypes = [C.POINTER(_abi.Params), C.POINTER(_abi.DSeq), C.POINTER(_abi.DSeq), C.POINTER(C.c_double)]
    p = params(d)
    ds = [_abi.dseq(s) for s in seqs]
    out = np.zeros(len(d["ia"]))
    for k, (i, j) in enumerate(zip(d["ia"], d["ib"])):
        v = C.c_double()
        rc = L.g2g_oracle_alnscored(C.byref(p), C.byref(ds[i]), C.byref(ds[j]), C.byref(v))
        assert rc == 0, rc
        out[k] = v.value
    return out


def oracle_alignb(d, seqs, k1=7, u1=0.6, maxvmf=0, pairs=None):
    """g2g_oracle_alignb_ng_lsp for every pair of a fixture: [(score, skeleton (n, 2) int32, pwd constants, centerB_ng calls)];
    maxvmf 0 = the reference's default MaxVmfSpace (16 Mi cells)"""
    L = oraclelib.load()
    L.g2g_oracle_alignb_ng_lsp.argtypes = [C.POINTER(_abi.Params), C.POINTER(_abi.DSeq), C.POINTER(_abi.DSeq), C.c_long, C.POINTER(C.c_double),
                                           C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    p = params(d)
    p.k1, p.u1, p.molc = k1, u1, int(d["molc"][0])
    ds = [_abi.dseq(s) for s in seqs]
    out = []
    for k, (i, j) in enumerate(zip(d["ia"], d["ib"])):
        if pairs is not None and k not in pairs:
            out.append(None)
            continue
        v = C.c_double(); sk = C.POINTER(_abi.Skl)(); n = C.c_int(); pw = (C.c_double * 6)(); nc = C.c_int()
        rc = L.g2g_oracle_alignb_ng_lsp(C.byref(p), C.byref(ds[i]), C.byref(ds[j]), int(maxvmf), C.byref(v), C.byref(sk), C.byref(n), pw, C.byref(nc))
        if rc == -2 and maxvmf:                               # G2G_ERR_MODE: centerB_ng left a part that is no DP (the reference crashes there)
            out.append((None, None, list(pw), nc.value))
            continue
        assert rc == 0, rc
        out.append((v.value, oraclelib.skl_to_np(sk, n.value), list(pw), nc.value))
        L.g2g_oracle_free(sk)
    return out
