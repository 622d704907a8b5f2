#!/usr/bin/env python3
"""Time g2g_alignb_ng_batch on pairs that go through the linear-space recursion (>= 16 Mi cells): the two full-size reference
fixtures (results checked) and a configs[4]-like batch of 4800-nt DNA sequences with -yl3 (all pairs of 16).
Usage (GPU box): python tools/lsp_probe.py"""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import distlib
from prrn_aln_amd import engine, guide, operator as op
from prrn_aln_amd.synth import make_family, DNA


def alp(d):
    p = distlib.params(d)
    p.k1, p.u1, p.molc = 7, 0.6, int(d["molc"][0])
    return p


ctx = engine.Context()
for name in ("lsp_dna3_4400_ls3", "lsp_prot2_6200"):
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "lsp", name + ".npz")))
    seqs = distlib.split(d)
    for rep in range(2):
        t = time.perf_counter(); res = guide.alignb_ng_batch(ctx, alp(d), seqs, d["ia"], d["ib"]); dt = time.perf_counter() - t
    ok = all(st == 0 and scr == d["alignb_scr"][k] for k, (scr, skl, st) in enumerate(res))
    print("%-20s %d pairs  %.1f ms  (%.1f ms per pair)  equal to the reference: %s" % (name, len(res), 1e3 * dt, 1e3 * dt / len(res), ok), flush=True)
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "lsp", "lsp_dna3_4400_ls3.npz")))
fam = make_family(16, 4800, 5, alphabet=DNA, indel=0.01, max_indel=20)
seqs = [op.encode([r.replace("-", "")], op.DNA)[:, 0].copy() for r in fam.msa]
ia, ib = guide.all_pairs(len(seqs))
for threads in (256, 128):
    ctx.set_option("CENTER_THREADS", threads)
    for rep in range(2):
        t = time.perf_counter(); res = guide.alignb_ng_batch(ctx, alp(d), seqs, ia, ib); dt = time.perf_counter() - t
    cells = sum(len(seqs[i]) * len(seqs[j]) for i, j in zip(ia, ib))
    print("16 x 4800 nt, -yl3: %d pairs (%.3g cells), %d threads per phase: %.1f ms = %.1f ms per pair, statuses %s" %
          (len(res), cells, threads, 1e3 * dt, 1e3 * dt / len(res), sorted(set(st for _, _, st in res))), flush=True)
ctx.close()
