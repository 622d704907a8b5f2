#!/usr/bin/env python3
"""Diagnostics: what one small batch of full-size DPs costs (the regime g2g_refine runs in).  Divisions of the bench family from
the reference's progressive MSA; batches of 1, 2, 4, 8, 16 DPs through the resident-batch path (kernel time by HIP events) and
through g2g_align2_batch from host memory (wall), per record type."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family
fam = make_family(256, 1024, 1)
codes = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "msa", "prog256x1024.npz"))["codes"]
sw = sweep.Sweep(fam, op.AlnParam(), codes=codes)
ctx = engine.Context(0)
for k, v in (a.split("=") for a in sys.argv[1:]):
    ctx.set_option(k, v)
class H:
    def __init__(self, q): self.c = q
rng = np.random.RandomState(3)
for mode, name in ((9, "_pf"), (7, "_hf")):
    ids = [k for k in range(len(sw)) if sw.pwds[k].alnmode in ((mode,) if mode == 9 else (7, 8))]
    rng.shuffle(ids)
    for n in (1, 2, 4, 8, 16):
        pick = ids[:n]
        hs = [H(sw.pwds[k].problem) for k in pick]
        b = ctx.prepare(hs)
        b.run(); b.fetch()
        t = time.perf_counter(); b.run(); r = b.fetch(); wall = time.perf_counter() - t
        fwd, tb = b.times_ms()
        b.free()
        pw = [sw.pwds[k] for k in pick]
        op.align2_batch(ctx, pw)
        t = time.perf_counter(); op.align2_batch(ctx, pw); e2e = time.perf_counter() - t
        cells = sum(int(sw.cells[k]) for k in pick)
        print("%s n=%2d cells %.3g  resident run+fetch %6.1f ms (kernels %6.1f + traceback %4.1f)   align2_batch from host %6.1f ms" % (name, n, cells, 1e3 * wall, fwd, tb, 1e3 * e2e))
ctx.close()
