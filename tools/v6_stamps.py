#!/usr/bin/env python3
"""Diagnostics: cycle shares of the phases of a v6 step.  Needs a library built with G2G_EXTRA_FLAGS=-DG2G_V6_STAMP
(python -c 'from prrn_aln_amd import build; build.build_lib(force=True)'); runs one bench sweep (pf divisions only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prrn_aln_amd import engine, operator as op, sweep, _lib
from prrn_aln_amd.synth import make_family
fam = make_family(256, 1024, 1)
sw = sweep.Sweep(fam, op.AlnParam(), weighted=True)
pf = [k for k in sw.order if sw.pwds[k].alnmode == 9]
ctx = engine.Context(0)
class H:
    def __init__(self, q): self.c = q
b = ctx.prepare([H(sw.pwds[k].problem) for k in pf])
L = _lib.lib()
out = (C.c_ulonglong * 16)()
b.run(); L.g2g_v6_stamps(out, 1)
b.run(); L.g2g_v6_stamps(out, 1)
names = ["step top/handover/loads", "heads", "Y merges", "X merges", "decide", "newdelta a + incdelta", "newdelta b + incdelta", "outputs", "bail", "loop tail (trace idx, sync)"]
tot = sum(out[k] for k in range(10))
steps = out[12]
print("steps %d, cycles/step %.0f (s_memtime ticks)" % (steps, tot / max(1, steps)))
for k, n in enumerate(names):
    print("%-28s %6.1f %%  %8.0f ticks/step" % (n, 100.0 * out[k] / tot, out[k] / max(1, steps)))
print("fwd ms", b.times_ms())
