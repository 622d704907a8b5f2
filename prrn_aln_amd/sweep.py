"""One refinement sweep as a batch of independent group-vs-rest re-alignments.

prrn's Randiv (reference src/randiv.cc:158-239, TREEDIV) walks the 2N-3 branches of a tree; every branch
splits the current MSA into two groups which are re-aligned by align2() (Prrn::onecycle, reference
src/prrn5.cc:516-543).  Given the current MSA all divisions are independent -- that is what batches on one
GPU and shards across GPUs.  This module builds the divisions (host side), deals them to
ranks and packs results into fixed-size slots for the all-gather between sweeps."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import operator as op
from .synth import Family, tree_branches, tree_weights

GAP = 1


def division_groups(codes: np.ndarray, side: Sequence[int]) -> Tuple[np.ndarray, np.ndarray, List[int], List[int]]:
    """Split an MSA ((len, N) residue codes) into the two groups of a division and drop the columns that
    became all-gap in each (GapsList::delcommongap, reference src/mgaps.cc:181)."""
    n = codes.shape[1]
    mask = np.zeros(n, bool)
    mask[list(side)] = True
    ia, ib = np.flatnonzero(mask), np.flatnonzero(~mask)
    a, b = codes[:, ia], codes[:, ib]
    a = a[(a != GAP).any(axis=1)]
    b = b[(b != GAP).any(axis=1)]
    return np.ascontiguousarray(a), np.ascontiguousarray(b), ia.tolist(), ib.tolist()


class Sweep:
    """All divisions of one sweep over a family's current MSA, as PwdM objects (host-side builders run in
    libg2g.so).  `order` lists division ids by decreasing DP size (longest-processing-time first)."""

    def __init__(self, fam: Family, alp: op.AlnParam, weighted: bool = True, limit: Optional[int] = None,
                 workers: Optional[int] = None, codes: Optional[np.ndarray] = None, ctx=None):
        """`codes`: start from this MSA ((columns, members) residue codes of the family's members) instead of fam.msa.
        `ctx`: an engine.Context -- the derived arrays of all divisions' groups (thickness, vectors, gap profiles) are then built on
        the device in one batch (g2g_pwdm_create_batch) instead of on host threads; same objects either way."""
        self.fam, self.alp = fam, alp
        self.codes = op.encode(fam.msa, alp.molc) if codes is None else np.ascontiguousarray(codes, np.uint8)
        n = len(fam.msa)
        self.weights = np.asarray(tree_weights(fam.tree, n)) if weighted else None
        self.branches = tree_branches(fam.tree)
        if limit:
            self.branches = self.branches[:limit]
        # The builders of a division (aggregate, thickness, profiles, gap profiles: g2g_host.cpp) touch only their own
        # objects and run outside the GIL (ctypes): divisions are built by a pool of host threads (5.9 s -> <1 s for the
        # 509 divisions of the bench sweep on 16 cores).
        def build(side):
            a, b, ia, ib = division_groups(self.codes, side)
            wa = None if self.weights is None else self.weights[ia]
            wb = None if self.weights is None else self.weights[ib]
            ga, gb = op.mSeq(a, alp, wa), op.mSeq(b, alp, wb)
            return (ga, gb), (op.PwdM([ga, gb], alp) if ctx is None else None)

        import time
        t0 = time.perf_counter()
        nthr = workers if workers is not None else min(16, os.cpu_count() or 1)
        if nthr > 1 and len(self.branches) > 1:
            with ThreadPoolExecutor(max_workers=nthr) as pool:
                built = list(pool.map(build, self.branches))
        else:
            built = [build(side) for side in self.branches]
        self.groups = [g for g, _ in built]
        t1 = time.perf_counter()
        self.pwds: List[op.PwdM] = [p for _, p in built] if ctx is None else op.PwdM.batch(ctx, self.groups, alp)
        # seconds: splitting the MSA into groups (+ the PwdMs when they are built on the host threads, ctx None); the device batch
        self.t_split, self.t_batch = t1 - t0, time.perf_counter() - t1
        self.cells = np.array([band_cells(p.problem) for p in self.pwds], np.int64)
        self.order = np.argsort(-self.cells, kind="stable")

    def __len__(self):
        return len(self.pwds)


def band_cells(q) -> int:
    """In-band cells of Fwd2c::forwardB: sum over rows of (n9 - n), reference src/fwd2c.h:373-374,393."""
    al, ar, bl, br = q.a.left, q.a.right, q.b.left, q.b.right
    m = np.arange(al, ar, dtype=np.int64)
    lo = np.maximum(m + q.lw, bl)
    hi = np.minimum(m + q.up + 1, br)
    return int(np.clip(hi - lo, 0, None).sum())


def shard(order: Sequence[int], world: int, rank: int) -> List[int]:
    """Deal the size-ordered divisions round-robin: rank r takes items r, r+world, ...."""
    return [int(k) for k in list(order)[rank::world]]


# ---- fixed-size result slots for the exchange between sweeps --------------------------------------
# slot = [division id, status, nskl, score (2 x int32 = the double's bits), corners (cap x 2)]
SLOT_HDR = 5


def pack_slots(ids: Sequence[int], results, cap: int, nslots: int) -> np.ndarray:
    """results[i] = (score, skl (n,2) int32, status) for division ids[i]; padded to nslots slots."""
    out = np.full((nslots, SLOT_HDR + 2 * cap), -1, np.int32)
    for row, (k, (scr, skl, st)) in enumerate(zip(ids, results)):
        n = min(len(skl), cap)
        out[row, 0] = k
        out[row, 1] = st if len(skl) <= cap else -99
        out[row, 2] = n
        out[row, 3:5] = np.frombuffer(np.float64(scr).tobytes(), np.int32)
        out[row, SLOT_HDR:SLOT_HDR + 2 * n] = skl[:n].reshape(-1)
    return out


def unpack_slots(slots: np.ndarray):
    """Inverse of pack_slots over the gathered array: {division id: (score, skl, status)}."""
    out = {}
    for row in slots.reshape(-1, slots.shape[-1]):
        k = int(row[0])
        if k < 0:
            continue
        n = int(row[2])
        scr = float(np.frombuffer(row[3:5].astype(np.int32).tobytes(), np.float64)[0])
        out[k] = (scr, row[SLOT_HDR:SLOT_HDR + 2 * n].reshape(n, 2).copy(), int(row[1]))
    return out
