#!/usr/bin/env python3
"""A mixed-mode workload for `rocprofv3 --kernel-trace --stats`: every record type of the path in ONE batch -- pairwise DPs of
gap-free single sequences (mode NGP: DPunit, kernel g2g_v7_*), divisions of tiny families (mode NTV: DPunit_nv, g2g_v8_*),
tree-branch divisions of a 96 x 600 family (HLF / RHF: g2g_v3r_hf*, GPF: g2g_v6_pf* / g2g_v2_pf*) -- then the guide-tree stage
(alnScoreD all pairs: g2g_dist_lds_kernel; alignB_ng on a sample of pairs: g2g_pairaln_lds*).  Prints cells per mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prrn_aln_amd import engine, guide, operator as op, sweep
from prrn_aln_amd.synth import make_family
alp = op.AlnParam()
pw, keep = [], []
for seed in range(64):                                     # NGP: pairs of single sequences
    fam = make_family(2, 800, 100 + seed, sub=0.3, indel=0.02)
    g = [op.mSeq(op.encode([r.replace("-", "")], alp.molc), alp, None) for r in fam.msa]
    keep.append(g); pw.append(op.PwdM(g, alp))
for seed in range(24):                                     # NTV: families of 3-6 sequences
    sw = sweep.Sweep(make_family(3 + seed % 4, 800, 300 + seed, indel=0.03, max_indel=20), alp)
    keep.append(sw); pw += sw.pwds
sw = sweep.Sweep(make_family(96, 600, 7), alp)             # HLF / RHF / GPF
keep.append(sw); pw += sw.pwds
modes = {}
ctx = engine.Context(0)
class H:
    def __init__(self, q): self.c = q
b = ctx.prepare([H(p.problem) for p in pw])
for p, c in zip(pw, [b.cells()] * 0 or [None] * len(pw)):
    modes[p.alnmode] = modes.get(p.alnmode, 0) + 1
for _ in range(3):
    t = time.perf_counter(); b.run(); dt = time.perf_counter() - t
res = b.fetch()
print("mixed batch: %d DPs (per mode: %s), %.3g cells, %.1f ms per run, failed %d" % (len(pw), sorted(modes.items()), b.cells(), 1e3 * dt, sum(1 for r in res if r[3] != 0)), flush=True)
fam = make_family(96, 600, 7)
seqs = [op.encode([r.replace("-", "")], alp.molc)[:, 0].copy() for r in fam.msa]
prm, _keep = alp.to_c()
ia, ib = guide.all_pairs(len(seqs))
for _ in range(3):
    t = time.perf_counter(); sc, st = guide.alnscored_batch(ctx, prm, seqs, ia, ib); dt = time.perf_counter() - t
print("alnScoreD: %d pairs, %.1f ms per call, failed %d" % (len(ia), 1e3 * dt, int((st != 0).sum())), flush=True)
for _ in range(2):
    t = time.perf_counter(); r = guide.alignb_ng_batch(ctx, prm, seqs, ia[:1024], ib[:1024]); dt = time.perf_counter() - t
print("alignB_ng: %d pairs, %.1f ms per call, failed %d" % (len(r), 1e3 * dt, sum(1 for x in r if x[2] != 0)), flush=True)
