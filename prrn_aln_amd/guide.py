"""The guide-tree stage on the GPU: pairwise alignment-score distances between the single
sequences of a family, as the reference's dpscore() computes them (src/phyl.cc:222-252):

    scr1[i]   = selfAlnScr(seq i)                                   (src/aln2.cc:54-64)
    score     = alnScoreD(seq i, seq j)                             (src/fwd2d1.cc:324-338 -> Fwd2d::forwardD)  <- the GPU part
    dist[i,j] = 100 * (1 - (score + u * |len_i - len_j| / 2) / sqrt(scr1[i] * scr1[j]))   (alnscore2dist, src/aln2.cc:325-333)

All DPs of the N (N - 1) / 2 pairs go to the device as ONE batch (g2g_alnscored_batch: the sequences are uploaded once, a
pair is two indices).  There is no CPU fallback: without libg2g.so / a GPU the calls raise."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _abi
from ._lib import G2GError, last_error, lib


def self_score(codes: np.ndarray, simmtx: np.ndarray) -> float:
    """selfAlnScr of a single sequence: the diagonal entries summed in sequence order"""
    c = np.asarray(codes, np.intp)
    if len(c) == 0:
        return 0.0
    return float(np.cumsum(simmtx[c, c])[-1])          # cumsum adds left to right, like the reference's loop


def scores_to_dist(scores: np.ndarray, ia: Sequence[int], ib: Sequence[int], lens: Sequence[int], selfs: np.ndarray, u: float) -> np.ndarray:
    """alnscore2dist for the global mode with dpscore's denominator; returns 1 - score / denominator (dpscore stores 100 x)"""
    lens = np.asarray(lens, np.int64)
    ia = np.asarray(ia, np.intp); ib = np.asarray(ib, np.intp)
    dlen = np.abs(lens[ia] - lens[ib]).astype(np.float32)
    corr = (np.float32(u) * dlen / np.float32(2)).astype(np.float64)       # float arithmetic in the reference (alprm.u is a float)
    denome = np.sqrt(selfs[ia] * selfs[ib])
    return 1.0 - (np.asarray(scores, np.float64) + corr) / denome


def all_pairs(n: int) -> Tuple[np.ndarray, np.ndarray]:
    ia, ib = np.triu_indices(n, 1)
    return ia.astype(np.int32), ib.astype(np.int32)


def alnscored_batch(ctx, prm: "_abi.Params", seqs: Sequence[np.ndarray], ia: Sequence[int], ib: Sequence[int]):
    """alnScoreD for every index pair: (scores float64[npairs], status int32[npairs])"""
    L = lib()
    ds = (_abi.DSeq * len(seqs))()
    keep = []
    for k, s in enumerate(seqs):
        x = np.ascontiguousarray(s, np.uint8)
        keep.append(x)
        ds[k].res = x.ctypes.data_as(_abi.c_u8p)
        ds[k].len, ds[k].left, ds[k].right = len(x), 0, len(x)
    ia = np.ascontiguousarray(ia, np.int32); ib = np.ascontiguousarray(ib, np.int32)
    out = np.zeros(len(ia), np.float64)
    st = np.zeros(len(ia), np.int32)
    rc = L.g2g_alnscored_batch(ctx._h, C.byref(prm), len(seqs), ds, len(ia), ia.ctypes.data_as(C.POINTER(C.c_int32)),
                               ib.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(_abi.c_f64p), st.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise G2GError("g2g_alnscored_batch rc=%d: %s" % (rc, last_error()))
    return out, st


def alignb_ng_batch(ctx, prm: "_abi.Params", seqs: Sequence[np.ndarray], ia: Sequence[int], ib: Sequence[int]):
    """alignB_ng for every index pair: [(score, skeleton (n, 2) int32, status)]"""
    L = lib()
    ds = (_abi.DSeq * len(seqs))()
    keep = []
    for k, s in enumerate(seqs):
        x = np.ascontiguousarray(s, np.uint8)
        keep.append(x)
        ds[k].res = x.ctypes.data_as(_abi.c_u8p)
        ds[k].len, ds[k].left, ds[k].right = len(x), 0, len(x)
    ia = np.ascontiguousarray(ia, np.int32); ib = np.ascontiguousarray(ib, np.int32)
    n = len(ia)
    scr = np.zeros(n, np.float64); st = np.zeros(n, np.int32)
    skl = (C.POINTER(_abi.Skl) * max(n, 1))(); nskl = (C.c_int * max(n, 1))()
    rc = L.g2g_alignb_ng_batch(ctx._h, C.byref(prm), len(seqs), ds, n, ia.ctypes.data_as(C.POINTER(C.c_int32)),
                               ib.ctypes.data_as(C.POINTER(C.c_int32)), scr.ctypes.data_as(_abi.c_f64p), skl, nskl,
                               st.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise G2GError("g2g_alignb_ng_batch rc=%d: %s" % (rc, last_error()))
    out = []
    for k in range(n):
        a = np.zeros((nskl[k], 2), np.int32)
        if nskl[k] and skl[k]:
            a[:] = np.ctypeslib.as_array(C.cast(skl[k], C.POINTER(C.c_int32)), shape=(nskl[k] * 2,)).reshape(-1, 2)
            L.g2g_free(skl[k])
        out.append((float(scr[k]), a, int(st[k])))
    return out


def distance_matrix(ctx, prm: "_abi.Params", seqs: Sequence[np.ndarray], simmtx: np.ndarray) -> np.ndarray:
    """dpscore over all pairs: the condensed distance vector (pair order of all_pairs), 100 x as the reference stores it"""
    ia, ib = all_pairs(len(seqs))
    scores, st = alnscored_batch(ctx, prm, seqs, ia, ib)
    if (st != 0).any():
        raise G2GError("alnScoreD failed for %d pairs (first status %d)" % (int((st != 0).sum()), int(st[st != 0][0])))
    selfs = np.array([self_score(s, simmtx) for s in seqs])
    return 100.0 * scores_to_dist(scores, ia, ib, [len(s) for s in seqs], selfs, prm.u)
