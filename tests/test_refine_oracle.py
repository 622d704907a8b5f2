"""The refinement trajectory of the reference (tests/golden/refine_*.json) re-walked on the CPU: g2g_refine -- the product's
C++ loop behind the C ABI (csrc/g2g_refine.cpp: Randiv order, divisions read off the MSA matrix, windows of speculative
divisions, in-order acceptance) with the host builders -- and the ORACLE in the scorer's seat (forwardB / stdskl / calcSpScore
on the CPU; the GPU needs a GPU).  It must give bit-identical (DP score, fstat.val) at every align2() of the reference's loop,
accept exactly the moves it accepted (member lists and skeletons) and end in its final MSA.  The GPU walks the same
trajectories in tests/test_gpu_refine.py."""
import numpy as np
import pytest

import refinelib
from prrn_aln_amd._lib import G2GError
from prrn_aln_amd.refine import KTree, refine_native

FIX = refinelib.fixtures(small_only=True)      # (the 48 x 300 and 256 x 1024 traces are the GPU tests': minutes to hours on the CPU oracle)


@pytest.mark.parametrize("path", FIX, ids=[refinelib.fixture_id(p) for p in FIX])
def test_native_loop_with_oracle_scorer_walks_the_reference_trajectory(path):
    f, tree, alp, start = refinelib.load(path)
    final, steps, stats = refine_native(None, start, tree, alp, seed=1, maxitr=10, window=8, scorer=refinelib.oracle_scorer(), want_moves=True)
    refinelib.check_against_trace(f, final, steps, stats)
    assert stats["batches"] < len(f["align2"])                        # the DPs really were evaluated in windows
    assert stats["wait_timeouts"] == 0 and stats["recovered_dps"] == 0


def test_window_policy_does_not_change_the_trajectory():
    """Speculation only decides how many divisions are evaluated together: any (window_min, window) walks the same trajectory."""
    f, tree, alp, start = refinelib.load(FIX[0])
    for wmin, wmax in ((1, 1), (4, 4), (2, 32)):
        final, steps, stats = refine_native(None, start, tree, alp, window=wmax, window_min=wmin, scorer=refinelib.oracle_scorer())
        refinelib.check_against_trace(f, final, steps, stats)
        if wmax == 1:
            assert stats["divisions_wasted"] == 0


def _bad(tree, **kw):
    d = dict(left=list(tree.left), right=list(tree.right), parent=list(tree.parent), vol=list(tree.vol), cur=list(tree.cur))
    for k, (i, v) in kw.items():
        d[k][i] = v
    return KTree(**d)


def test_malformed_trees_are_refused():
    """The tree crosses the C ABI: index ranges, leaf / inner shape, parent-child consistency, a single root, no cycle, vol > 0."""
    f, tree, alp, start = refinelib.load(FIX[0])
    n = tree.n_leaves
    root = tree.parent.index(-1)
    inner = next(k for k in range(n, len(tree.left)) if k != root)
    cases = [
        _bad(tree, left=(inner, len(tree.left) + 5)),              # child index out of range
        _bad(tree, left=(0, 1)),                                    # a leaf with a child
        _bad(tree, parent=(inner, inner)),                          # a node that is its own parent (cycle)
        _bad(tree, parent=(inner, -1)),                             # two roots
        _bad(tree, vol=(inner, 0.0)),                               # a volume that would divide by zero
        _bad(tree, right=(inner, tree.left[inner])),                # both children the same node
    ]
    for t in cases:
        with pytest.raises(G2GError) as e:
            refine_native(None, start, t, alp, scorer=refinelib.oracle_scorer())
        assert "rc=-" in str(e.value)


def test_a_failing_scorer_fails_the_call():
    f, tree, alp, start = refinelib.load(FIX[0])
    with pytest.raises(G2GError):
        refine_native(None, start, tree, alp, window=4, scorer=refinelib.oracle_scorer(fail_after=3))
