#!/usr/bin/env python3
"""Timing aid: a batch of group-vs-group DPs of tiny families (alignment mode NTV_ALB, record type DPunit_nv) on the strip
kernel (v8) and on the one-workgroup-per-DP kernel (v1: option NO_V8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family
nfam = int(sys.argv[1]) if len(sys.argv) > 1 else 48
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ls = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pw, keep = [], []
for seed in range(nfam):
    sw = sweep.Sweep(make_family(3 + seed % 4, length, 300 + seed, indel=0.03, max_indel=20), op.AlnParam(ls=ls))
    keep.append(sw)
    pw += [p for p in sw.pwds if p.alnmode == 10]
ctx = engine.Context(0)
class H:
    def __init__(self, q): self.c = q
for tag in ("v8", "v1"):
    if tag == "v1": ctx.set_option("NO_V8", 1)
    b = ctx.prepare([H(p.problem) for p in pw])
    b.run(); r0 = b.fetch()
    t = time.perf_counter()
    for _ in range(5): b.run()
    dt = (time.perf_counter() - t) / 5
    print("%s: %d DPs, %.3g cells, %.2f ms per batch, %.3g cells/s" % (tag, len(pw), b.cells(), 1e3 * dt, b.cells() / dt), [round(x[0], 3) for x in r0[:2]], flush=True)
    del b
