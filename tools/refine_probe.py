#!/usr/bin/env python3
"""Diagnostics: where a refinement's wall time goes.  Runs g2g_refine on the first `iters` cycles of the 256 x 1024 trajectory
(tests/golden/refine_prot256x1024_prog.json.gz) and prints the wall time, the library's own split (G2G_REFINE_TIMES: builders,
align2 batches, calcSpScore, the rest) and the window statistics.  Under `rocprofv3 --kernel-trace --stats -- python3
tools/refine_probe.py` the kernel table says which kernels the latency regime spends its GPU time in.
    python3 tools/refine_probe.py [iters=1] [window=16] [window_min=0] [OPTION=value ...]      (options: g2g_ctx_set_option)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("G2G_REFINE_TIMES", "1")
import refinelib
from prrn_aln_amd import engine
from prrn_aln_amd.refine import refine_native

kw = dict(iters=1, window=16, window_min=0)
opts = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k in kw:
        kw[k] = int(v)
    else:
        opts[k] = v
path = [p for p in refinelib.fixtures() if "256x1024" in p][0]
f, tree, alp, start = refinelib.load(path)
ctx = engine.Context(0)
for k, v in opts.items():
    ctx.set_option(k, v)
t = time.perf_counter()
final, steps, stats = refine_native(ctx, start, tree, alp, seed=1, maxitr=kw["iters"], window=kw["window"], window_min=kw["window_min"])
wall = time.perf_counter() - t
dps = [s for s in steps if not s["skipped"]]
ref = f["align2"][:len(dps)]
same = all(s["scr"] == r["scr"] and s["val_new"] == r["val"] for s, r in zip(dps, ref))
print("%d divisions (%d DPs, %d accepted) in %.2f s: %d batches, %d divisions wasted, %.1f ms per batch, %.1f ms per DP of the trajectory; "
      "scores identical to the reference's trace: %s; wait time-outs %d, recovered DPs %d"
      % (len(steps), len(dps), stats["accepted"], wall, stats["batches"], stats["divisions_wasted"], 1e3 * wall / max(1, stats["batches"]),
         1e3 * wall / max(1, len(dps)), same, stats["wait_timeouts"], stats["recovered_dps"]))
print("waiting waves off the machine for > 4 ms at a stretch: %d (longest %.1f ms)" % ctx.wait_gaps())
if stats["recovered_dps"]:
    print(ctx.last_timeout())
ctx.close()
