#!/usr/bin/env python3
"""Generate tests/golden/dist/*.npz from the REAL reference (oracle/_ref): the guide-tree DPs on single sequences.

Per fixture: residue codes of N sequences (as the reference's reader encoded them), the parameters Fwd2d / Aln2b1 read
(alprm.u / v / scale / tgapf / sh, the default similarity matrix), and for every pair i < j what the reference computed:
alnScoreD (fwd2d1.cc:324), alnscore2dist as dpscore calls it (phyl.cc:222-252), the self scores, and alignB_ng's score and
skeleton (fwd2b1.cc:1347).  Data only.  The reference keeps parameters in process globals: one process per parameter set.

Usage: python tools/make_dist_golden.py
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden", "dist")


def run(name, molc, ls, sh, tgapf, seqs, uv=None, maxvmf=None, sub="dist", pairs=None):
    import refdump
    R = refdump.RefLib(molc=molc, ls=ls, sh=sh, tgapf=tgapf)
    L = R.lib
    if uv is not None:                   # non-dyadic penalties: partial sums of the boundary ramps round (Fwd2d ctor, fwd2d1.cc:58-98)
        L.ref_set_uv.argtypes = [C.c_double, C.c_double]
        L.ref_set_uv(*uv)
    if maxvmf is not None:               # the linear-space recursion on DPs of this many cells or more (default 16 Mi)
        L.ref_set_vmfspace.argtypes = [C.c_long]
        L.ref_set_vmfspace(maxvmf)
    L.ref_seq_read.restype = C.c_void_p
    L.ref_seq_read.argtypes = [C.c_char_p]
    L.ref_seq_len.argtypes = [C.c_void_p]
    L.ref_seq_range.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_seq_codes.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte)]
    L.ref_alnscored.restype = C.c_double
    L.ref_alnscored.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_selfalnscr.restype = C.c_double
    L.ref_selfalnscr.argtypes = [C.c_void_p]
    L.ref_alnscore2dist.restype = C.c_double
    L.ref_alnscore2dist.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.ref_alignb_ng.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double)]
    uvst = (C.c_double * 4)()
    shv, dim, rows = C.c_int(), C.c_int(), C.c_int()
    mtx = (C.c_double * 4096)()
    assert L.ref_dist_params(uvst, C.byref(shv), mtx, 4096, C.byref(dim), C.byref(rows)) == 0
    hs, codes, lens = [], [], []
    with tempfile.TemporaryDirectory() as td:
        for k, s in enumerate(seqs):
            fn = os.path.join(td, "s%d.fa" % k)
            with open(fn, "w") as fd:
                fd.write(">s%d\n%s\n" % (k, s))
            h = L.ref_seq_read(fn.encode())
            assert h
            n = L.ref_seq_len(h)
            l, r = C.c_int(), C.c_int()
            assert L.ref_seq_range(h, C.byref(l), C.byref(r)) == 1 and l.value == 0 and r.value == n
            buf = (C.c_ubyte * n)()
            L.ref_seq_codes(h, buf)
            hs.append(h); codes.append(np.frombuffer(buf, np.uint8).copy()); lens.append(n)
    N = len(hs)
    selfs = np.array([L.ref_selfalnscr(h) for h in hs])
    ia, ib, scd, dist, bscr, bskl, bnskl, pwdc = [], [], [], [], [], [], [], (C.c_double * 6)()
    cap = 4 * (max(lens) + 2)
    out = (C.c_int * (2 * cap))()
    for i in range(N):
        for j in range(i + 1, N):
            if pairs is not None and (i, j) not in pairs:
                continue
            ia.append(i); ib.append(j)
            scd.append(L.ref_alnscored(hs[i], hs[j]))
            dist.append(L.ref_alnscore2dist(hs[i], hs[j], float(np.sqrt(selfs[i] * selfs[j]))))
            s = C.c_double()
            n = L.ref_alignb_ng(hs[i], hs[j], C.byref(s), out, cap, pwdc)
            assert n > 0, n
            bscr.append(s.value); bnskl.append(n)
            bskl.append(np.array(out[:2 * n], np.int32).reshape(n, 2))
    gold = os.path.join(os.path.dirname(GOLD), sub)
    os.makedirs(gold, exist_ok=True)
    np.savez_compressed(os.path.join(gold, name + ".npz"), maxvmf=np.array([maxvmf if maxvmf is not None else 16 * 1024 * 1024], np.int64),
                        molc=np.array([molc]), ls=np.array([ls]), lens=np.array(lens, np.int32), codes=np.concatenate(codes),
                        u=np.array([uvst[0]]), v=np.array([uvst[1]]), scale=np.array([uvst[2]]), tgapf=np.array([uvst[3]]),
                        sh=np.array([shv.value]), simmtx=np.array(mtx[:rows.value * dim.value]).reshape(rows.value, dim.value),
                        ia=np.array(ia, np.int32), ib=np.array(ib, np.int32), alnscored=np.array(scd), dist=np.array(dist),
                        selfscr=selfs, alignb_scr=np.array(bscr), alignb_nskl=np.array(bnskl, np.int32),
                        alignb_skl=np.concatenate(bskl), pwdb=np.array(pwdc[:6]))
    print("%-22s %d seqs, %d pairs, lens %d..%d, sh %d, alnScoreD %.1f .. %.1f" % (name, N, len(ia), min(lens), max(lens), shv.value, min(scd), max(scd)))


def synth(n, length, seed, alphabet=None, **kw):
    from prrn_aln_amd.synth import make_family, DNA
    fam = make_family(n, length, seed, **({"alphabet": DNA} if alphabet == "dna" else {}), **kw)
    return [r.replace("-", "") for r in fam.msa]


def fasta_members(path):
    from prrn_aln_amd import seqio
    _, rows = seqio.read_msa(path)
    return [r.replace("-", "") for r in rows]


JOBS = {
    "prot12": lambda: run("prot12_sh60", 1, 0, 0, None, synth(12, 180, 11, indel=0.04, max_indel=15)),
    "prot_ragged": lambda: run("prot_ragged", 1, 0, 0, None, [s[: 40 + 23 * k] for k, s in enumerate(synth(8, 260, 12, indel=0.03))]),
    "prot_unbanded": lambda: run("prot8_sh100", 1, 0, -100, None, synth(8, 150, 13, indel=0.06, max_indel=25)),
    "prot_tgapf": lambda: run("prot8_tgapf1", 1, 0, 0, 1.0, synth(8, 140, 14, indel=0.05)),
    # terminal gaps at half price / free: Fwd2d::lastD's walks (fwd2d1.cc:100-134) and the scaled ramps; ragged lengths so that
    # the optimal path ends in long terminal gaps
    "prot_tgapf05": lambda: run("prot8_tgapf05", 1, 0, 0, 0.5, [s[: 50 + 17 * k] for k, s in enumerate(synth(8, 200, 17, indel=0.05))]),
    "prot_tgapf0": lambda: run("prot8_tgapf0", 1, 0, 0, 0.0, [s[(5 * k) % 30: 60 + 19 * k] for k, s in enumerate(synth(8, 220, 18, indel=0.05))]),
    "dna_tgapf05_ls3": lambda: run("dna8_tgapf05_ls3", 2, 3, 0, 0.5, [s[: 70 + 21 * k] for k, s in enumerate(synth(8, 240, 19, alphabet="dna", indel=0.04))]),
    # penalties that are not dyadic rationals: the ramps' partial sums round, the closed form no longer applies
    "prot_uv": lambda: run("prot8_u21_v93", 1, 0, 0, None, synth(8, 160, 20, indel=0.05), uv=(2.1, 9.3)),
    "prot_uv_tgapf": lambda: run("prot8_u21_v93_tgapf03", 1, 0, 0, 0.3, [s[: 45 + 16 * k] for k, s in enumerate(synth(8, 180, 21, indel=0.05))], uv=(2.1, 9.3)),
    "prot_ls3": lambda: run("prot8_ls3", 1, 3, 0, None, synth(8, 200, 15, indel=0.05, max_indel=40)),
    "dna": lambda: run("dna10_sh60", 2, 0, 0, None, synth(10, 240, 16, alphabet="dna", indel=0.03)),
    # ---- tests/golden/lsp: alignB_ng through the linear-space recursion (lspB_ng / centerB_ng, fwd2b1.cc:382-782, 1053-1095) ----
    # MaxVmfSpace lowered so that small DPs recurse several levels deep (the reference's own switch, vmf.cc:27) ...
    "lsp_prot": lambda: run("lsp_prot8_vmf4k", 1, 0, 0, None, synth(8, 220, 31, indel=0.05, max_indel=20), maxvmf=4096, sub="lsp"),
    "lsp_prot_unbanded": lambda: run("lsp_prot6_sh100_vmf2k", 1, 0, -100, None, [s[: 90 + 31 * k] for k, s in enumerate(synth(6, 260, 32, indel=0.06, max_indel=30))], maxvmf=2048, sub="lsp"),
    "lsp_prot_ls3": lambda: run("lsp_prot8_ls3_vmf4k", 1, 3, 0, None, synth(8, 240, 33, indel=0.06, max_indel=50), maxvmf=4096, sub="lsp"),
    "lsp_dna_ls3": lambda: run("lsp_dna8_ls3_vmf8k", 2, 3, 0, None, synth(8, 400, 34, alphabet="dna", indel=0.04, max_indel=30), maxvmf=8192, sub="lsp"),
    "lsp_dna_tgapf": lambda: run("lsp_dna6_tgapf05_vmf2k", 2, 0, 0, 0.5, [s[: 120 + 40 * k] for k, s in enumerate(synth(6, 360, 35, alphabet="dna", indel=0.04))], maxvmf=2048, sub="lsp"),
    "lsp_prot_uv": lambda: run("lsp_prot6_u21_v93_vmf1k", 1, 3, 0, 0.3, synth(6, 200, 36, indel=0.08, max_indel=40), uv=(2.1, 9.3), maxvmf=1024, sub="lsp"),
    # ... and the default 16 Mi on sequences long enough to cross it (configs[4]'s regime: 4096-nt DNA with -yl3; one protein pair)
    "lsp_dna_full": lambda: run("lsp_dna3_4400_ls3", 2, 3, 0, None, synth(3, 4400, 37, alphabet="dna", indel=0.01, max_indel=20), sub="lsp"),
    "lsp_prot_full": lambda: run("lsp_prot2_6200", 1, 0, 0, None, synth(2, 6200, 38, indel=0.01, max_indel=30), sub="lsp"),
    "pas": lambda: run("pas_native", 1, 0, 0, None, fasta_members("/root/reference/sample/pas/native_A") + fasta_members("/root/reference/sample/pas/native_B")),
}

if __name__ == "__main__":
    if len(sys.argv) > 1:
        JOBS[sys.argv[1]]()
    else:
        for j in JOBS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), j])
