"""Python glue over g2g_refine / g2g_pairsum (include/g2g.h): prrn's randomised iterative
refinement (Prrn::rir, reference src/prrn5.cc:633-666) runs in C++ behind the C ABI (csrc/g2g_refine.cpp); this module only
marshals the MSA, the weighting tree and the options, and offers an exchange callback on torch.distributed for runs sharded
over ranks.  Nothing here computes."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List

import numpy as np

from . import _abi
from . import operator as op
from ._lib import G2GError, last_error, lib


@dataclass
class KTree:
    """The weighting tree as the reference's Ktree::lead[] holds it: node id = tid (leaves 0..n-1 = members), -1 where a link is
    absent, vol / cur = the Kirchhoff weights the reference computed (building the tree is phylogeny code, outside this path)."""
    left: List[int]
    right: List[int]
    parent: List[int]
    vol: List[float]
    cur: List[float]

    @property
    def n_leaves(self) -> int:
        return (len(self.left) + 1) // 2

    def to_c(self):
        """(g2g_tree, arrays to keep alive)"""
        arr = lambda x, t: np.ascontiguousarray(x, t)
        keep = (arr(self.left, np.int32), arr(self.right, np.int32), arr(self.parent, np.int32), arr(self.vol, np.float64), arr(self.cur, np.float64))
        T = _abi.Tree()
        T.n_nodes = len(keep[0])
        i32p = C.POINTER(C.c_int32)
        T.left, T.right, T.parent = (k.ctypes.data_as(i32p) for k in keep[:3])
        T.vol, T.cur = keep[3].ctypes.data_as(_abi.c_f64p), keep[4].ctypes.data_as(_abi.c_f64p)
        return T, keep


def torch_exchange(device=None):
    """An exchange callback for g2g_refine on torch.distributed (RCCL when `device` is a GPU, gloo on the CPU): all-gathers the
    ranks' buffers.  Returns (callback object to keep alive, rank, world).  A rank whose part of a window failed still calls it
    (its error code travels in its buffer, g2g.h), so the collective is entered by every rank in every window."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()

    def cb(user, mine, n_ints, out):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(mine, shape=(n_ints,)).copy())
            if device is not None:
                t = t.to(device)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            dst = np.ctypeslib.as_array(out, shape=(world * n_ints,))
            for r, part in enumerate(parts):
                dst[r * n_ints:(r + 1) * n_ints] = part.cpu().numpy()
            return 0
        except Exception:
            return 1
    return _abi.EXCHANGE_FN(cb), rank, world


def refine_native(ctx, codes: np.ndarray, tree: KTree, alp: op.AlnParam, seed: int = 1, maxitr: int = 10, window: int = 32,
                  exchange=None, scorer=None, window_min: int = 0, want_moves: bool = False, slot_cap: int = 0):
    """g2g_refine: (refined MSA (len, many) uint8, [step dicts], stats dict).  `exchange`: the triple torch_exchange returns, for
    a run sharded over ranks.  `scorer`: an _abi.SCORE_FN standing in for the GPU (tests; ctx may then be None).  With
    want_moves every accepted move is recorded as stats["moves"] = [(branch, la, lb, skl (n, 2))]."""
    L = lib()
    codes = np.ascontiguousarray(codes, np.uint8)
    ln, many = codes.shape
    prm, _sm = alp.to_c()
    T, _keep = tree.to_c()
    O = _abi.RefineOpts()
    O.seed, O.maxitr, O.window, O.window_min, O.slot_cap = seed, maxitr, window, window_min, slot_cap
    if exchange is not None:
        O.exchange, O.rank, O.world = exchange
    if scorer is not None:
        O.scorer = scorer
    moves = []
    if want_moves:
        def on_accept(user, branch, na, la, nb, lb, nskl, skl):
            moves.append((branch, [la[i] for i in range(na)], [lb[i] for i in range(nb)],
                          np.array([(skl[i].m, skl[i].n) for i in range(nskl)], np.int32).reshape(-1, 2)))
        acb = _abi.ACCEPT_FN(on_accept)
        O.on_accept = acb
    out = _abi.c_u8p(); olen = C.c_int(); steps = C.POINTER(_abi.RefineStep)(); ns = C.c_int(); st = _abi.RefineStats()
    rc = L.g2g_refine(ctx._h if ctx is not None else None, C.byref(prm), many, ln, codes.ctypes.data_as(_abi.c_u8p), C.byref(T), C.byref(O),
                      C.byref(out), C.byref(olen), C.byref(steps), C.byref(ns), C.byref(st))
    if rc != 0:
        raise G2GError("g2g_refine rc=%d: %s" % (rc, last_error()))
    final = np.ctypeslib.as_array(out, shape=(olen.value * many,)).reshape(olen.value, many).copy()
    L.g2g_free(out)
    log = [dict(branch=s.branch, na=s.na, nb=s.nb, swp=bool(s.swp), accepted=bool(s.accepted), skipped=bool(s.skipped), scr=s.scr,
                val_new=s.val_new, val_old=s.val_old, delta=s.delta, t_ms=s.t_ms) for s in (steps[i] for i in range(ns.value))]
    L.g2g_free(steps)
    stats = {k: getattr(st, k) for k, _ in _abi.RefineStats._fields_ if k != "reserved"}
    if want_moves:
        stats["moves"] = moves
    return final, log, stats


def pairsum(ctx, codes: np.ndarray, tree: KTree, alp: op.AlnParam, use_pw: bool = True) -> float:
    """g2g_pairsum: Ssrel::pairsum_ss of an MSA -- the sum-of-pairs score prrn reports"""
    codes = np.ascontiguousarray(codes, np.uint8)
    ln, many = codes.shape
    prm, _sm = alp.to_c()
    T, _keep = tree.to_c()
    out = C.c_double()
    rc = lib().g2g_pairsum(ctx._h, C.byref(prm), many, ln, codes.ctypes.data_as(_abi.c_u8p), C.byref(T), 1 if use_pw else 0, C.byref(out))
    if rc != 0:
        raise G2GError("g2g_pairsum rc=%d: %s" % (rc, last_error()))
    return out.value
