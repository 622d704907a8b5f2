#!/usr/bin/env python3
"""Debug aid: a context that ran CU-share launches, closed, and another one created."""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.dump_traceback_later(120, exit=True)
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family
fam = make_family(n_seq=24, length=700, seed=31)
sw = sweep.Sweep(fam, op.AlnParam())
ctx = engine.Context()
t = time.perf_counter(); res = op.align2_batch(ctx, sw.pwds); print("align2_batch %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
print(ctx.mem_counters(), ctx.counters(), flush=True)
t = time.perf_counter(); ctx.close(); print("close %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
t = time.perf_counter(); ctx = engine.Context(); print("create %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
res2 = op.align2_batch(ctx, sw.pwds)
assert all(a[0] == b[0] for a, b in zip(res, res2))
ctx.close(); print("done", flush=True)
