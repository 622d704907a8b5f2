#!/usr/bin/env python3
"""Timing aid: a batch of pairwise DPs between gap-free single sequences (alignment mode NGP_ALB, record type DPunit) on the
strip kernel (v7) and on the one-workgroup-per-DP kernel (v1: G2G_NO_V7=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prrn_aln_amd import engine, operator as op
from prrn_aln_amd.synth import make_family
alp = op.AlnParam()
pw, keep = [], []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 128):
    fam = make_family(2, int(sys.argv[2]) if len(sys.argv) > 2 else 1000, 100 + seed, sub=0.3, indel=0.02)
    rows = [r.replace("-", "") for r in fam.msa]
    g = [op.mSeq(op.encode([r], alp.molc), alp, None) for r in rows]
    keep.append(g); pw.append(op.PwdM(g, alp))
assert all(p.alnmode == 6 for p in pw)
ctx = engine.Context(0)
class H:
    def __init__(self, q): self.c = q
for tag, env in (("v7", None), ("v1", "1")):
    if env: os.environ["G2G_NO_V7"] = env
    b = ctx.prepare([H(p.problem) for p in pw])
    b.run(); r0 = b.fetch()
    t = time.perf_counter()
    for _ in range(5): b.run()
    dt = (time.perf_counter() - t) / 5
    print("%s: %d DPs, %.3g cells, %.2f ms per batch, %.3g cells/s" % (tag, len(pw), b.cells(), 1e3 * dt, b.cells() / dt), [round(x[0], 3) for x in r0[:2]], flush=True)
    del b
