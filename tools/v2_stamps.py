#!/usr/bin/env python3
"""Diagnostics: time shares of the phases of a v2 step in the latency regime (a handful of full-size _pf DPs of the bench family).
Needs a library built with -DG2G_V2_STAMP in the v2 unit (G2G_LIB=... selects it).  s_memtime ticks of the first lane of every wave,
summed; slots: 1 sources, 2 merges, 3 decisions, 4 list updates, 5 record scalars + trace byte, 6 parking / boundary stores /
column ring, 7 the barrier that ends the step.   python3 tools/v2_stamps.py [n_dps=4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prrn_aln_amd import engine, operator as op, sweep, _lib
from prrn_aln_amd.synth import make_family
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fam = make_family(256, 1024, 1)
codes = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "msa", "prog256x1024.npz"))["codes"]
sw = sweep.Sweep(fam, op.AlnParam(), codes=codes)
ids = [k for k in range(len(sw)) if sw.pwds[k].alnmode == 9]
np.random.RandomState(3).shuffle(ids)
ctx = engine.Context(0)
class H:
    def __init__(self, q): self.c = q
b = ctx.prepare([H(sw.pwds[k].problem) for k in ids[:n]])
L = _lib.lib()
out = (C.c_ulonglong * 16)()
b.run(); L.g2g_v2_stamps(out, 1)
b.run(); L.g2g_v2_stamps(out, 1)
names = {1: "sources (staging, neighbours)", 2: "merges (phase A)", 3: "shuffles + decisions", 4: "list updates (phase B)", 5: "record scalars, trace byte", 6: "parking, boundary stores, column ring", 7: "barrier at the end of the step"}
for base, who in ((0, "first wave of each workgroup"), (8, "the other waves")):
    tot = sum(out[base + k] for k in range(8))
    print(who, "total ticks %.3g" % tot)
    for k in range(1, 8):
        print("   %-42s %5.1f %%" % (names[k], 100.0 * out[base + k] / max(1, tot)))
print("kernel ms", b.times_ms())
