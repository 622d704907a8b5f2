"""Live check of the oracle against the real reference (oracle/_ref) on seeded random families.
Skipped where oracle/_ref is not built (the GPU box normally has the prebuilt files; a bare checkout
does not).  One subprocess per parameter set: the reference keeps its parameters in globals."""
import os
import subprocess
import sys

import pytest

import refdump

pytestmark = pytest.mark.skipif(not refdump.available(), reason="oracle/_ref not built")

WORKER = r'''
import sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import refdump, oraclelib
from prrn_aln_amd import _abi
from prrn_aln_amd.synth import make_family, tree_branches, tree_weights, drop_common_gaps, DNA, PROTEIN
molc, ls, nseq, length, seed, weighted = {args!r}
R = refdump.RefLib(molc=molc, ls=ls)
L = oraclelib.load()
fam = make_family(nseq, length, seed, alphabet=(DNA if molc == 2 else PROTEIN), indel=0.02)
w = tree_weights(fam.tree, nseq) if weighted else None
bad = []
modes = set()
for b in tree_branches(fam.tree):
    o = [i for i in range(nseq) if i not in b]
    ra = drop_common_gaps([fam.msa[i] for i in b]); rb = drop_common_gaps([fam.msa[i] for i in o])
    ga = R.group([fam.names[i] for i in b], ra, [w[i] for i in b] if w else None)
    gb = R.group([fam.names[i] for i in o], rb, [w[i] for i in o] if w else None)
    d = R.align_dump(ga, gb)
    modes.add(int(d["alnmode"][0]))
    h = _abi.problem_from_arrays(d)
    scr, cells, tr = oraclelib.forward(L, h)
    skl = oraclelib.stdskl(L, tr)
    if scr != d["scr"][0] or not np.array_equal(tr, d["vmf_trace"]) or not np.array_equal(skl, d["align2_skl"]):
        bad.append(len(b))
    R.free(ga); R.free(gb)
print(json.dumps({{"bad": bad, "modes": sorted(modes)}}))
'''

CASES = [
    (1, 0, 20, 90, 101, True),
    (1, 0, 9, 70, 102, False),
    (1, 3, 14, 80, 103, True),
    (2, 3, 12, 120, 104, True),
    (2, 0, 6, 100, 105, False),
]


@pytest.mark.parametrize("args", CASES, ids=["prot20w", "prot9", "prot14_ls3", "dna12_ls3", "dna6"])
def test_live(args):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, "-c", WORKER.format(root=root, args=args)],
                                  stderr=subprocess.DEVNULL, timeout=600)
    import json
    r = json.loads(out.decode().strip().splitlines()[-1])
    assert r["bad"] == [], r
