"""Host side of prrn's randomised iterative refinement -- SURVEY.md §8 row f2 -- on top of level 1 of the C ABI.

What the reference does (Prrn::rir / onecycle / divideseq / gather / calcfact, reference src/prrn5.cc:414-666; Randiv and
McRand, src/randiv.cc:34-239; synthgap / delcommongap / aggregate, src/mgaps.cc:181-369; gap2skl, src/gaps.cc:274):
draw a branch of the weighting tree, split the MSA into the two groups on either side of it, drop the columns that became
all-gap in each, re-align the two groups (PwdM + align2), and keep the new alignment if its weighted sum-of-pairs score
(PreSpScore::calcSpScore of the new skeleton minus that of the old one, times the pair weight of the branch) is better.
One accepted move changes the MSA every later division is taken from: a sequential hill climb.

Here the same trajectory is produced with the DPs BATCHED: the branch sequence does not depend on the outcomes (a mixed
congruential generator), so the next `window` divisions are all built from the current MSA and evaluated in one
g2g_align2_batch + two g2g_spscore_batch calls; they are then looked at in generator order, the first improving one is
applied, and the divisions behind it -- computed on an MSA that no longer exists -- are thrown away and drawn again.
(The reference's own parallel form, best_of_n, uses a different acceptance rule and reaches a different MSA)

Own formulation, not the reference's data structures: the MSA is a (columns x members) matrix of residue codes instead of
per-member gap run lists; a division's "current" skeleton is read off that matrix; an accepted skeleton is applied by
interleaving the two groups' columns.  The tree (topology, Kirchhoff `vol` / `cur` per node) is an INPUT: building it is
the reference's phylogeny code (src/phyl.cc), outside this path -- fixtures carry the tree the reference used."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import operator as op

GAP = 1
FEPS = 1.0e-7                                   # reference src/cmn.h:54


# ---- glibc rand(), the generator McRand seeds itself from (src/randiv.cc:41-51) --------------------------------
class GlibcRand:
    """rand()/srand() of glibc (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]); a process that never
    called srand() runs on seed 1."""

    def __init__(self, seed: int = 1):
        self.srand(seed)

    def srand(self, seed: int) -> None:
        seed &= 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed if seed < 0x80000000 else seed - (1 << 32)
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1], 127773)            # C division truncates toward zero
            if r[i - 1] < 0 and lo:
                hi += 1
                lo -= 127773
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r[i] = w
        for i in range(31, 34):
            r[i] = r[i - 31]
        self.r = [x & 0xFFFFFFFF for x in r]
        for _ in range(310):
            self._next()

    def _next(self) -> int:
        v = (self.r[-31] + self.r[-3]) & 0xFFFFFFFF
        self.r.append(v)
        if len(self.r) > 64:
            del self.r[:-34]
        return v

    def rand(self) -> int:
        return self._next() >> 1


class McRand:
    """Mixed congruential generator over [0, 2^p), reference src/randiv.cc:34-56, randiv.h:36-47."""

    def __init__(self, p: int, rn: int = 1, libc: Optional[GlibcRand] = None):
        libc = libc or GlibcRand()
        self.mod = 1 << p
        if rn == 0:
            self.coef, self.val = 1, self.mod - 1
        else:
            val = libc.rand() if rn == 1 else rn
            libc.srand(val)
            self.coef = (libc.rand() // 4 * 4 + 5) % self.mod
            self.val = val % self.mod

    def next(self) -> int:
        self.val = (self.coef * self.val + 1) % self.mod
        return self.val


@dataclass
class KTree:
    """The weighting tree as the reference's Ktree::lead[] holds it: node id = tid (leaves 0..n-1 = members)."""
    left: List[int]
    right: List[int]
    parent: List[int]
    vol: List[float]
    cur: List[float]

    @property
    def n_leaves(self) -> int:
        return (len(self.left) + 1) // 2

    def leaves(self, tid: int) -> List[int]:
        out, stack = [], [tid]
        while stack:
            k = stack.pop()
            if self.left[k] < 0 and self.right[k] < 0:
                out.append(k)
            else:
                stack.append(self.right[k])
                stack.append(self.left[k])
        return sorted(out)

    def calcfact(self, tid: int) -> Tuple[float, np.ndarray]:
        """Prrn::calcfact / childfact (src/prrn5.cc:414-441): weight of every member when the tree is cut above node
        `tid`, and the pair weight of the cut."""
        w = np.zeros(self.n_leaves)

        def child(node: int, fact: float) -> None:
            for leaf in self.leaves(node):
                w[leaf] = self.vol[leaf] * fact

        node = tid
        child(node, 1.0 / self.vol[node])
        fact = 1.0
        while self.parent[node] >= 0:
            father = self.parent[node]
            other = self.left[father] if self.left[father] != node else self.right[father]
            child(other, fact / self.vol[father])
            node = father
            fact *= self.cur[node]
        return self.cur[tid], w


class TreeDivisions:
    """Randiv in TREEDIV mode (src/randiv.cc:158-178,217-226): branch ids in generator order."""

    def __init__(self, tree: KTree, seed: int = 1):
        n = tree.n_leaves
        self.tree = tree
        self.cycle = 2 * n - 3
        p, x = 0, 1
        while x < self.cycle:
            x <<= 1
            p += 1
        self.mcr = McRand(p, seed)

    def next(self) -> int:
        while True:
            r = self.mcr.next()
            if r < self.cycle:
                return r

    def members(self, tid: int) -> Tuple[List[int], List[int]]:
        """bin2lst2 + the swap of Prrn::divideseq: (larger group, smaller group); ties keep the complement first."""
        inside = self.tree.leaves(tid)
        s = set(inside)
        outside = [i for i in range(self.tree.n_leaves) if i not in s]
        if len(outside) < len(inside):
            return inside, outside
        return outside, inside


# ---- the MSA as a matrix ---------------------------------------------------------------------------------------
def split_columns(codes: np.ndarray, ia: Sequence[int], ib: Sequence[int]):
    """The two groups of a division without their all-gap columns, and the skeleton of their CURRENT alignment
    (what delcommongap + gap2skl give the reference): corners (m, n) where the heading changes."""
    a, b = codes[:, ia], codes[:, ib]
    ka, kb = (a != GAP).any(axis=1), (b != GAP).any(axis=1)
    m = np.concatenate([[0], np.cumsum(ka)])
    n = np.concatenate([[0], np.cumsum(kb)])
    keep = ka | kb                                          # (an MSA has no all-gap column; be safe)
    pts = np.stack([m, n], axis=1)
    pts = pts[np.concatenate([[True], keep])]
    d = np.diff(pts, axis=0)
    corner = np.ones(len(pts), bool)
    if len(d) > 1:
        corner[1:-1] = (d[1:] != d[:-1]).any(axis=1)
    skl = pts[corner].astype(np.int32)
    return np.ascontiguousarray(a[ka]), np.ascontiguousarray(b[kb]), skl


def join_columns(a: np.ndarray, b: np.ndarray, skl: np.ndarray, ia: Sequence[int], ib: Sequence[int], n_members: int) -> np.ndarray:
    """Apply a skeleton (synthgap, src/mgaps.cc:350): interleave the columns of the two groups."""
    total = int(sum(max(int(skl[k + 1][0] - skl[k][0]), int(skl[k + 1][1] - skl[k][1])) for k in range(len(skl) - 1)))
    out = np.full((total, n_members), GAP, np.uint8)
    col = 0
    for k in range(len(skl) - 1):
        m0, n0 = int(skl[k][0]), int(skl[k][1])
        dm, dn = int(skl[k + 1][0]) - m0, int(skl[k + 1][1]) - n0
        if dm == dn:
            out[col:col + dm, ia] = a[m0:m0 + dm]
            out[col:col + dm, ib] = b[n0:n0 + dn]
            col += dm
        elif dn == 0:
            out[col:col + dm, ia] = a[m0:m0 + dm]
            col += dm
        elif dm == 0:
            out[col:col + dn, ib] = b[n0:n0 + dn]
            col += dn
        else:
            raise ValueError("skeleton segment is neither diagonal nor a gap")
    return out[:col]


def lt0(delta: float) -> bool:
    """lt(0, delta), src/cmn.h:63"""
    return 0.0 < delta - FEPS * max(1.0, abs(delta))


@dataclass
class Step:
    branch: int
    na: int
    nb: int
    swp: bool
    scr: float
    val_new: float
    val_old: float
    delta: float
    accepted: bool
    lst: Tuple[List[int], List[int]]
    skl: Optional[np.ndarray]


def gpu_scorer(ctx):
    """Scores a list of divisions on the GPU through the C ABI: g2g_align2_batch for the DPs, g2g_spscore_batch for the
    sum-of-pairs scores of the current and the new alignment.  Returns [(DP score, new skeleton, raw score of the current
    alignment, fstat.val of the new one)]."""
    def score(divs):
        pwds = [d["pw"] for d in divs]
        res = op.align2_batch(ctx, pwds)
        old = [d["old"] for d in divs]
        new = [skl for (_, skl, _) in res]
        fs = op.calcSpScore_batch(ctx, pwds + pwds, old + new)
        k = len(divs)
        out = []
        for i, d in enumerate(divs):
            scr, skl, st = res[i]
            if st != 0 or fs[i][2] != 0 or fs[k + i][2] != 0:
                raise RuntimeError("division %d: status %d / %d / %d" % (d["branch"], st, fs[i][2], fs[k + i][2]))
            out.append((scr, skl, fs[i][3], fs[k + i][0]))
        return out
    return score


class Exchange:
    """The one exchange step of a sharded window: rank r scores divisions r, r + world, ... of the
    size-ordered window, packs (DP score, the two sum-of-pairs scores, new skeleton) into fixed-size slots, and an
    all-gather (RCCL when the tensors live on the GPU, gloo in the CPU tests) leaves every rank with every result --
    every rank then takes the same accept / reject decisions and holds the same MSA, no broadcast needed."""

    def __init__(self, cap: int = 4096):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.cap = cap

    def share(self, order: Sequence[int]) -> List[int]:
        return [int(k) for k in list(order)[self.rank::self.world]]

    def gather(self, mine: Sequence[int], results, n_total: int, device=None):
        import torch
        nslots = (n_total + self.world - 1) // self.world
        width = 9 + 2 * self.cap
        buf = np.full((nslots, width), -1, np.int32)
        for row, (k, (scr, skl, raw_old, val_new)) in enumerate(zip(mine, results)):
            if len(skl) > self.cap:
                raise RuntimeError("skeleton of %d corners exceeds the slot capacity %d" % (len(skl), self.cap))
            buf[row, 0], buf[row, 1], buf[row, 2] = k, 0, len(skl)
            buf[row, 3:9] = np.frombuffer(np.array([scr, raw_old, val_new], np.float64).tobytes(), np.int32)
            buf[row, 9:9 + 2 * len(skl)] = np.asarray(skl, np.int32).reshape(-1)
        t = torch.from_numpy(buf)
        if device is not None:
            t = t.to(device)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        out = {}
        for part in parts:
            for row in part.cpu().numpy():
                k = int(row[0])
                if k < 0:
                    continue
                n = int(row[2])
                scr, raw_old, val_new = np.frombuffer(row[3:9].astype(np.int32).tobytes(), np.float64)
                out[k] = (float(scr), row[9:9 + 2 * n].reshape(n, 2).copy(), float(raw_old), float(val_new))
        return [out[k] for k in range(n_total)]


class Refiner:
    """Prrn::rir over a matrix MSA, DPs batched `window` divisions at a time (and, with an Exchange, sharded over ranks).
    `scorer`: callable(divisions) -> [(DP score, new skeleton, raw current score, new fstat.val)]; default: the GPU."""

    def __init__(self, ctx, codes: np.ndarray, tree: KTree, alp: op.AlnParam, seed: int = 1, maxitr: int = 10,
                 window: int = 32, scorer=None, exchange: Optional[Exchange] = None, device=None):
        self.codes, self.tree, self.alp = np.ascontiguousarray(codes, np.uint8), tree, alp
        self.div = TreeDivisions(tree, seed)
        self.maxitr, self.window = maxitr, window
        self.scorer = scorer if scorer is not None else gpu_scorer(ctx)
        self.exchange, self.device = exchange, device
        self.steps: List[Step] = []
        self.batches = 0
        self.wasted = 0
        self.scored_here = 0

    def _evaluate(self, branches: Sequence[int]) -> List[dict]:
        todo = []
        for t in branches:
            la, lb = self.div.members(t)
            pwt, w = self.tree.calcfact(t)
            a, b, skl0 = split_columns(self.codes, la, lb)
            if len(a) == len(self.codes) and len(b) == len(self.codes):
                # neither group has a column to drop: Prrn::divideseq returns no skeleton and onecycle no score
                # (src/prrn5.cc:497-498,518-521): the division counts as a failure without a DP
                todo.append(dict(branch=t, la=la, lb=lb, skip=True))
                continue
            # a group of one member is the member itself in the reference (aliaseq): weight 1
            wa = w[la] if len(la) > 1 else np.ones(1)
            wb = w[lb] if len(lb) > 1 else np.ones(1)
            ga, gb = op.mSeq(a, self.alp, wa), op.mSeq(b, self.alp, wb)
            pw = op.PwdM([ga, gb], self.alp)
            old = skl0[:, ::-1].copy() if pw.swp else skl0
            todo.append(dict(branch=t, la=la, lb=lb, pwt=pwt, a=a, b=b, old=old, groups=(ga, gb), pw=pw))
        live = [d for d in todo if not d.get("skip")]
        if live:
            if self.exchange is None:
                res = self.scorer(live)
                self.scored_here += len(live)
            else:                                              # my share of the window, largest rectangles first
                order = sorted(range(len(live)), key=lambda i: -(len(live[i]["a"]) * len(live[i]["b"])))
                mine = self.exchange.share(order)
                part = self.scorer([live[i] for i in mine]) if mine else []
                self.scored_here += len(mine)
                res = self.exchange.gather(mine, part, len(live), self.device)
            for d, (scr, skl, raw_old, val_new) in zip(live, res):
                same = skl.shape == d["old"].shape and np.array_equal(skl, d["old"])
                # Prrn::onecycle (src/prrn5.cc:523,535): the NEW alignment enters with Gsinfo.fstat.val (rescaled by PwdM::Vab),
                # the CURRENT one with the return value of calcSpScore(SKL*), which is not rescaled -- kept as the reference has it
                d["scr"], d["val_old"], d["val_new"] = scr, raw_old, val_new
                d["delta"] = 0.0 if same else d["pwt"] * (val_new - raw_old)
                d["skl1"] = skl[:, ::-1].copy() if d["pw"].swp else skl       # back to (larger group, smaller group)
        self.batches += 1
        return todo

    def run(self) -> np.ndarray:
        """Returns the refined MSA; self.steps holds the trajectory."""
        cycle = self.div.cycle
        maxi = self.maxitr * cycle
        nrep, i = 0, 0
        pending: List[int] = []                              # branch ids drawn but not yet consumed
        win = 2
        while i < maxi:
            while len(pending) < min(win, maxi - i):
                pending.append(self.div.next())
            batch = self._evaluate(pending[:win])
            consumed = 0
            accepted = False
            for d in batch:
                consumed += 1
                i += 1
                if d.get("skip"):
                    self.steps.append(Step(d["branch"], len(d["la"]), len(d["lb"]), False, 0.0, 0.0, 0.0, float("-inf"), False, (d["la"], d["lb"]), None))
                    nrep += 1
                    if nrep >= cycle or i >= maxi:
                        break
                    continue
                ok = lt0(d["delta"])
                self.steps.append(Step(d["branch"], len(d["la"]), len(d["lb"]), d["pw"].swp, d["scr"], d["val_new"], d["val_old"],
                                       d["delta"], ok, (d["la"], d["lb"]), d["skl1"] if ok else None))
                if ok:
                    self.codes = join_columns(d["a"], d["b"], d["skl1"], d["la"], d["lb"], self.codes.shape[1])
                    nrep = 1
                    accepted = True
                    break
                nrep += 1
                if nrep >= cycle or i >= maxi:
                    break
            self.wasted += len(batch) - consumed
            del pending[:consumed]
            if nrep >= cycle:
                break
            win = 2 if accepted else min(self.window, 2 * win)
        return self.codes


# ---- the same loop in C++ behind the C ABI (g2g_refine, csrc/g2g_refine.cpp) ------------------------------------------
def torch_exchange(device=None):
    """An exchange callback for g2g_refine on torch.distributed (RCCL when `device` is a GPU, gloo on the CPU): all-gathers the
    ranks' slot buffers.  Returns (callback object to keep alive, rank, world)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _abi
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
                  exchange=None):
    """g2g_refine: (refined MSA (len, many) uint8, [RefineStep-like dicts], stats dict).  `exchange`: the triple torch_exchange
    returns, for a run sharded over ranks."""
    import ctypes as C
    from . import _abi
    from ._lib import G2GError, last_error, lib
    L = lib()
    codes = np.ascontiguousarray(codes, np.uint8)
    ln, many = codes.shape
    prm, _sm = alp.to_c()
    arr = lambda x, t: np.ascontiguousarray(x, t)
    left, right, parent = arr(tree.left, np.int32), arr(tree.right, np.int32), arr(tree.parent, np.int32)
    vol, cur = arr(tree.vol, np.float64), arr(tree.cur, np.float64)
    T = _abi.Tree()
    T.n_nodes = len(left)
    i32p = C.POINTER(C.c_int32)
    T.left, T.right, T.parent = left.ctypes.data_as(i32p), right.ctypes.data_as(i32p), parent.ctypes.data_as(i32p)
    T.vol, T.cur = vol.ctypes.data_as(_abi.c_f64p), cur.ctypes.data_as(_abi.c_f64p)
    O = _abi.RefineOpts()
    O.seed, O.maxitr, O.window = seed, maxitr, window
    if exchange is not None:
        O.exchange, O.rank, O.world = exchange
    out = _abi.c_u8p(); olen = C.c_int(); steps = C.POINTER(_abi.RefineStep)(); ns = C.c_int(); st = _abi.RefineStats()
    rc = L.g2g_refine(ctx._h, C.byref(prm), many, ln, codes.ctypes.data_as(_abi.c_u8p), C.byref(T), C.byref(O), C.byref(out),
                      C.byref(olen), C.byref(steps), C.byref(ns), C.byref(st))
    if rc != 0:
        raise G2GError("g2g_refine rc=%d: %s" % (rc, last_error()))
    final = np.ctypeslib.as_array(out, shape=(olen.value * many,)).reshape(olen.value, many).copy()
    L.g2g_free(out)
    log = [dict(branch=s.branch, na=s.na, nb=s.nb, swp=bool(s.swp), accepted=bool(s.accepted), skipped=bool(s.skipped), scr=s.scr,
                val_new=s.val_new, val_old=s.val_old, delta=s.delta) for s in (steps[i] for i in range(ns.value))]
    L.g2g_free(steps)
    stats = {k: getattr(st, k) for k, _ in _abi.RefineStats._fields_ if k != "reserved"}
    return final, log, stats


def pairsum(ctx, codes: np.ndarray, tree: KTree, alp: op.AlnParam, use_pw: bool = True) -> float:
    """g2g_pairsum: Ssrel::pairsum_ss of an MSA -- the sum-of-pairs score prrn reports"""
    import ctypes as C
    from . import _abi
    from ._lib import G2GError, last_error, lib
    codes = np.ascontiguousarray(codes, np.uint8)
    ln, many = codes.shape
    prm, _sm = alp.to_c()
    arr = lambda x, t: np.ascontiguousarray(x, t)
    left, right, parent = arr(tree.left, np.int32), arr(tree.right, np.int32), arr(tree.parent, np.int32)
    vol, cur = arr(tree.vol, np.float64), arr(tree.cur, np.float64)
    T = _abi.Tree()
    T.n_nodes = len(left)
    i32p = C.POINTER(C.c_int32)
    T.left, T.right, T.parent = left.ctypes.data_as(i32p), right.ctypes.data_as(i32p), parent.ctypes.data_as(i32p)
    T.vol, T.cur = vol.ctypes.data_as(_abi.c_f64p), cur.ctypes.data_as(_abi.c_f64p)
    out = C.c_double()
    rc = lib().g2g_pairsum(ctx._h, C.byref(prm), many, ln, codes.ctypes.data_as(_abi.c_u8p), C.byref(T), 1 if use_pw else 0, C.byref(out))
    if rc != 0:
        raise G2GError("g2g_pairsum rc=%d: %s" % (rc, last_error()))
    return out.value
