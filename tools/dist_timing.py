#!/usr/bin/env python3
"""Timing aid for f3: the all-pairs alnScoreD batch of a family (default 256 x 1024 aa: 32640 pairs) on the GPU, and a sample of
the same pairs on the CPU restatement (one core) for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from prrn_aln_amd import engine, guide, operator as op
from prrn_aln_amd.synth import make_family
import distlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dist", "prot12_sh60.npz")))
fam = make_family(n, length, 1)
alp = op.AlnParam()
seqs = [op.encode([r.replace("-", "")], alp.molc)[:, 0].copy() for r in fam.msa]
ia, ib = guide.all_pairs(n)
ctx = engine.Context(0)
prm = distlib.params(g)
got, st = guide.alnscored_batch(ctx, prm, seqs, ia[:64], ib[:64])
for rep in range(2):
    t = time.perf_counter()
    got, st = guide.alnscored_batch(ctx, prm, seqs, ia, ib)
    dt = time.perf_counter() - t
    lens = np.array([len(s) for s in seqs], np.int64)
    cells = float((lens[ia] * lens[ib]).sum())
    print("GPU: %d pairs, %.3g full-matrix cells, %.1f ms (incl. upload, launch, download), %.3g cells/s, failed %d" % (len(ia), cells, 1e3 * dt, cells / dt, int((st != 0).sum())), flush=True)
k = 40
d = dict(g); d["ia"], d["ib"] = ia[:k], ib[:k]
t = time.perf_counter()
want = distlib.oracle_scores(d, seqs)
dt = time.perf_counter() - t
print("CPU restatement, 1 core: %d pairs in %.2f s -> %.1f s for all %d pairs; equal to the GPU's: %s" % (k, dt, dt * len(ia) / k, len(ia), bool(np.array_equal(want, got[:k]))))
