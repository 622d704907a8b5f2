"""Synthetic sequence families for parity tests and bench.py (SURVEY.md §8(d) "Synthetic inputs").

A family is evolved from a random root sequence along a random binary tree: per branch every site is
substituted with probability `sub`, an indel opens with probability `indel` per site (insertion or
deletion with equal odds), indel length ~ U[1, 5]; residues are uniform over the alphabet.  Because the
generator tracks column identity through insertions and deletions it also returns the TRUE multiple
alignment, which bench.py uses as the start MSA of a refinement sweep (no reference code involved).

Everything is driven by `random.Random(seed)` so a (n_seq, length, seed) triple names the data exactly.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import List, Optional, Tuple

PROTEIN = "ARNDCQEGHILKMFPSTWYV"
DNA = "ACGT"


@dataclass
class TreeNode:
    left: Optional["TreeNode"] = None
    right: Optional["TreeNode"] = None
    leaf: int = -1          # leaf index, -1 for internal nodes
    length: float = 1.0     # branch length above this node (used for weights only)

    def leaves(self) -> List[int]:
        if self.leaf >= 0:
            return [self.leaf]
        return self.left.leaves() + self.right.leaves()


@dataclass
class Family:
    names: List[str]
    seqs: List[str]         # unaligned sequences
    msa: List[str]          # true alignment, '-' for gaps, all rows the same length
    tree: TreeNode
    alphabet: str


def random_tree(n: int, rng: random.Random) -> TreeNode:
    nodes = [TreeNode(leaf=i) for i in range(n)]
    while len(nodes) > 1:
        i = rng.randrange(len(nodes))
        a = nodes.pop(i)
        j = rng.randrange(len(nodes))
        b = nodes.pop(j)
        nodes.append(TreeNode(left=a, right=b, length=0.5 + rng.random()))
    return nodes[0]


def make_family(n_seq: int, length: int, seed: int = 1, alphabet: str = PROTEIN,
                sub: float = 0.08, indel: float = 0.01, max_indel: int = 5) -> Family:
    rng = random.Random(seed)
    tree = random_tree(n_seq, rng)
    # a sequence is a list of (column key, residue); column keys are floats kept sorted, a new
    # column is keyed between its neighbours so the global column order is always well defined
    root = [(float(i), rng.choice(alphabet)) for i in range(length)]
    leaves: List[Optional[List[Tuple[float, str]]]] = [None] * n_seq

    def evolve(seq):
        out = []
        i = 0
        n = len(seq)
        while i < n:
            key, res = seq[i]
            if rng.random() < indel:
                k = rng.randint(1, max_indel)
                if rng.random() < 0.5:          # deletion of k sites
                    i += k
                    continue
                nxt = seq[i + 1][0] if i + 1 < n else key + 1.0
                if rng.random() < sub:
                    res = rng.choice(alphabet)
                out.append((key, res))
                # independent insertions behind the same site share (left-justified) columns, as an
                # aligner would place them; nested insertions subdivide further
                step = (nxt - key) / (max_indel + 1)
                for t in range(1, k + 1):
                    out.append((key + step * t, rng.choice(alphabet)))
                i += 1
                continue
            if rng.random() < sub:
                res = rng.choice(alphabet)
            out.append((key, res))
            i += 1
        return out

    stack = [(tree, root)]
    while stack:
        node, seq = stack.pop()
        if node.leaf >= 0:
            leaves[node.leaf] = seq
            continue
        stack.append((node.right, evolve(seq)))
        stack.append((node.left, evolve(seq)))

    keys = sorted({k for s in leaves for k, _ in s})
    col = {k: i for i, k in enumerate(keys)}
    msa = []
    for s in leaves:
        row = ["-"] * len(keys)
        for k, r in s:
            row[col[k]] = r
        msa.append("".join(row))
    names = ["s%04d" % i for i in range(n_seq)]
    seqs = ["".join(r for _, r in s) for s in leaves]
    return Family(names, seqs, msa, tree, alphabet)


def drop_common_gaps(rows: List[str]) -> List[str]:
    """Remove columns that are gaps in every row (what GapsList::delcommongap does to a sub-group,
    reference src/mgaps.cc:181)."""
    if not rows:
        return rows
    keep = [j for j in range(len(rows[0])) if any(r[j] != "-" for r in rows)]
    return ["".join(r[j] for j in keep) for r in rows]


def tree_branches(tree: TreeNode) -> List[List[int]]:
    """All 2N-3 non-trivial bipartitions of an unrooted binary tree as leaf lists of one side
    (the TREEDIV partitions Randiv enumerates, reference src/randiv.cc:170)."""
    out: List[List[int]] = []

    def walk(node: TreeNode, is_root_child: bool, skip: bool):
        if not skip:
            out.append(sorted(node.leaves()))
        if node.leaf < 0:
            walk(node.left, False, False)
            walk(node.right, False, False)

    # the two root children define the same bipartition: emit only one of them
    walk(tree.left, True, False)
    walk(tree.right, True, True)
    return out


def tree_weights(tree: TreeNode, n: int) -> List[float]:
    """Simple deterministic per-leaf weights (not the reference's Kirchhoff weights): each branch
    length is shared equally among the leaves below it.  Only used to exercise the weighted scorers."""
    w = [0.0] * n

    def walk(node: TreeNode):
        lv = node.leaves()
        for i in lv:
            w[i] += node.length / len(lv)
        if node.leaf < 0:
            walk(node.left)
            walk(node.right)

    walk(tree)
    s = sum(w) / n
    return [x / s for x in w]
