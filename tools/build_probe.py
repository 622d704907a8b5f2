#!/usr/bin/env python3
"""Probe of the device builders (g2g_pwdm_create_batch): one small family's sweep built on host threads and on the device, timed;
prints where the device path spends its time (G2G_DEBUG_PREP=1).  usage: python tools/build_probe.py [nseq length]"""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.dump_traceback_later(100, exit=True)
pass
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
fam = make_family(n, L, 1) if n >= 100 else make_family(n, L, 3, indel=0.03)
alp = op.AlnParam()
ctx = engine.Context()
t = time.perf_counter(); a = sweep.Sweep(fam, alp, weighted=True); th = time.perf_counter() - t
print("host: %.1f ms (%d divisions)" % (1e3 * th, len(a)), flush=True)
b = None
for rep in range(3):
    b = None                        # (the previous sweep's device slabs go back to the pool first, as in a refinement loop)
    t = time.perf_counter(); b = sweep.Sweep(fam, alp, weighted=True, ctx=ctx); td = time.perf_counter() - t
    print("device: %.1f ms (split %.1f, batch %.1f)" % (1e3 * td, 1e3 * b.t_split, 1e3 * b.t_batch), flush=True)
from test_gpu_builders import same_problem
for k, (h, g) in enumerate(zip(a.pwds, b.pwds)):
    same_problem(g.problem, h.problem, k)
print("arrays identical", flush=True)
t = time.perf_counter(); r1 = op.align2_batch(ctx, a.pwds); t1 = time.perf_counter() - t
t = time.perf_counter(); r2 = op.align2_batch(ctx, b.pwds); t2 = time.perf_counter() - t
assert all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(r1, r2))
print("align2_batch: host-built inputs %.1f ms, device-resident inputs %.1f ms; results identical" % (1e3 * t1, 1e3 * t2), flush=True)
print(ctx.mem_counters(), flush=True)
