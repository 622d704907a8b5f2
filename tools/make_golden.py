#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built by oracle/Makefile.ref).

Each fixture is data only: the flattened inputs the reference's group-to-group DP consumed for one
align2() call (dumped out of the reference's own objects by oracle/ref_shim.cc) and what it produced
(score, raw VMF traceback, stdskl skeleton, HomScore, fstat).  Run in THIS container (needs
/root/reference); the GPU box only sees the committed .npz files.

Usage: python tools/make_golden.py            (regenerates everything)
The reference keeps parameters in process globals, so every parameter set runs in its own process.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
SAMPLE = "/root/reference/sample"


def save(name, d):
    # drop bulky arrays nobody checks (nres is not read by the DP)
    d = {k: v for k, v in d.items() if not k.endswith("_nres")}
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **d)
    print("%-28s mode %2d sim %3d Noll %d a %dx%d b %dx%d scr %.6f ntrace %d" % (
        name, d["alnmode"][0], d["sim2_kind"][0], d["Noll"][0], d["a_many"][0], d["a_len"][0],
        d["b_many"][0], d["b_len"][0], d["scr"][0] if "scr" in d else float("nan"),
        len(d.get("vmf_trace", []))))


def split_case(R, fam, side, weights, name):
    from prrn_aln_amd.synth import drop_common_gaps
    n = len(fam.msa)
    other = [i for i in range(n) if i not in side]
    ra = drop_common_gaps([fam.msa[i] for i in side])
    rb = drop_common_gaps([fam.msa[i] for i in other])
    wa = [weights[i] for i in side] if weights else None
    wb = [weights[i] for i in other] if weights else None
    ga = R.group([fam.names[i] for i in side], ra, wa)
    gb = R.group([fam.names[i] for i in other], rb, wb)
    save(name, R.align_dump(ga, gb))


def job_protein():
    import refdump
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    R = refdump.RefLib(molc=refdump.PROTEIN)
    # the reference's own sample inputs (sample/test.sh:3-7)
    for nm, fa, fb in (("pas_multiA_multiB", "pas/Multi_A", "pas/Multi_B"),
                       ("pas_nativeA_nativeB", "pas/native_A", "pas/native_B"),
                       ):
        # (pas/ce13a* carry ";b" intron annotations -> SigII intron-position bonus, fwd2c.h:446-452:
        #  gene-structure path, out of scope; NGP goldens come from unannotated seeded pairs below)
        ga = R.group_file(os.path.join(SAMPLE, fa))
        gb = R.group_file(os.path.join(SAMPLE, fb))
        save(nm, R.align_dump(ga, gb))
    # seeded synthetic families: every engine (DPunit / _hf / _pf / _nv) and scorer on this path
    fam = make_family(24, 120, 3)
    w = tree_weights(fam.tree, 24)
    br = tree_branches(fam.tree)
    done = set()
    for b in br:
        k = min(len(b), 24 - len(b))
        if k in done or k > 12:
            continue
        done.add(k)
        split_case(R, fam, b, w, "syn24x120_k%d" % k)
    # unweighted variant (integer scorers sim32i ...)
    split_case(R, fam, br[2], None, "syn24x120_unweighted")
    fam = make_family(64, 96, 5)
    w = tree_weights(fam.tree, 64)
    br = sorted(tree_branches(fam.tree), key=lambda b: -min(len(b), 64 - len(b)))
    split_case(R, fam, br[0], w, "syn64x96_balanced")      # both sides profiles: sim33
    split_case(R, fam, br[1], w, "syn64x96_balanced2")
    # tiny groups: naive (NTV) engine and no-gap (NGP) engine
    fam = make_family(5, 90, 7, indel=0.03)
    w = tree_weights(fam.tree, 5)
    for i, b in enumerate(tree_branches(fam.tree)):
        split_case(R, fam, b, w, "syn5x90_b%d" % i)
    fam = make_family(4, 80, 11, indel=0.03)
    for i, b in enumerate(tree_branches(fam.tree)):
        split_case(R, fam, b, None, "syn4x80_unw_b%d" % i)
    # no internal gaps at all -> NGP engine (DPunit): single pair, and gap-free groups
    fam = make_family(2, 300, 17, sub=0.3, indel=0.02)
    split_case(R, fam, [0], None, "syn2x300_pair")
    fam = make_family(2, 120, 19, sub=0.2, indel=0.05)
    split_case(R, fam, [0], None, "syn2x120_pair")
    fam = make_family(7, 100, 23, sub=0.2, indel=0.0)
    w = tree_weights(fam.tree, 7)
    for i, b in enumerate(tree_branches(fam.tree)[:4]):
        split_case(R, fam, b, w, "syn7x100_nogap_b%d" % i)
    fam = make_family(30, 60, 29, sub=0.2, indel=0.0)
    w = tree_weights(fam.tree, 30)
    br = sorted(tree_branches(fam.tree), key=lambda b: -min(len(b), 30 - len(b)))
    split_case(R, fam, br[0], w, "syn30x60_nogap_balanced")
    split_case(R, fam, br[-1], w, "syn30x60_nogap_leaf")


def job_dna_ls3():
    import refdump
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights, DNA
    R = refdump.RefLib(molc=refdump.DNA, ls=3)
    fam = make_family(16, 100, 2, alphabet=DNA, indel=0.02, max_indel=12)
    w = tree_weights(fam.tree, 16)
    br = tree_branches(fam.tree)
    done = set()
    for b in br:
        k = min(len(b), 16 - len(b))
        if k in done:
            continue
        done.add(k)
        split_case(R, fam, b, w, "dna16x100_ls3_k%d" % k)
    fam = make_family(3, 100, 4, alphabet=DNA, indel=0.05, max_indel=30)
    for i, b in enumerate(tree_branches(fam.tree)):
        split_case(R, fam, b, None, "dna3x100_ls3_b%d" % i)


def job_protein_ls3():
    import refdump
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    R = refdump.RefLib(molc=refdump.PROTEIN, ls=3)
    fam = make_family(16, 100, 9, indel=0.03, max_indel=20)
    w = tree_weights(fam.tree, 16)
    br = tree_branches(fam.tree)
    done = set()
    for b in br:
        k = min(len(b), 16 - len(b))
        if k in done:
            continue
        done.add(k)
        split_case(R, fam, b, w, "prot16x100_ls3_k%d" % k)


def job_protein_tgapf():
    import refdump
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    R = refdump.RefLib(molc=refdump.PROTEIN, tgapf=0.5)
    fam = make_family(12, 80, 13, indel=0.03)
    w = tree_weights(fam.tree, 12)
    br = tree_branches(fam.tree)
    done = set()
    for b in br:
        k = min(len(b), 12 - len(b))
        if k in done:
            continue
        done.add(k)
        split_case(R, fam, b, w, "prot12x80_tgapf05_k%d" % k)


def _w21_cases(R, tag, **kw):
    """weighted tiny families whose division is NTV with a several and b a single sequence: the SPunit_w21 unit of
    PreSpScore::calcSpScore (reference src/fspscore.cc:598-610, calcstat :192-250)"""
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    for seed, n in ((1, 3), (1, 4), (3, 3)):
        fam = make_family(n, 70, seed, indel=0.04, **kw)
        w = tree_weights(fam.tree, n)
        split_case(R, fam, tree_branches(fam.tree)[0], w, "%s%dx70_w21_s%d" % (tag, n, seed))


def job_w21_protein():
    import refdump
    _w21_cases(refdump.RefLib(molc=refdump.PROTEIN), "prot")


def job_w21_dna_ls3():
    import refdump
    from prrn_aln_amd.synth import DNA
    _w21_cases(refdump.RefLib(molc=refdump.DNA, ls=3), "dna_ls3_", alphabet=DNA)


def job_sim23():
    """gap-free groups with the SMALL group first: a = 2-3 raw members, b = a profile -> PwdM::sim23i / sim23w
    (reference src/maln2.cc:347-399 selection table, scorers :570-600,1273-1285); no swap in the NGP modes"""
    import refdump
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    R = refdump.RefLib(molc=refdump.PROTEIN)
    fam = make_family(30, 60, 29, sub=0.2, indel=0.0)
    w = tree_weights(fam.tree, 30)
    n = 30
    done = set()
    for b in tree_branches(fam.tree):
        small = list(b) if len(b) <= n - len(b) else [i for i in range(n) if i not in set(b)]
        k = len(small)
        if k not in (2, 3, 4) or k in done:
            continue
        done.add(k)
        split_case(R, fam, small, w, "syn30x60_nogap_small%d_first_w" % k)
        split_case(R, fam, small, None, "syn30x60_nogap_small%d_first_i" % k)


def job_intron():
    """inputs that carry exon-boundary annotations (';C join(...)' lines -> SigII): the intron-position bonus of forwardB
    (PfqItr::match_score, reference src/fwd2c.h:378-379,446-452,472).  BASELINE configs[0]'s own pair and groups of the
    annotated family sample/pas/ce13a*."""
    import refdump
    R = refdump.RefLib(molc=refdump.PROTEIN)
    pas = os.path.join(SAMPLE, "pas")
    pairs = [("ce13a1", "ce13a2"), ("ce13a3", "ce13a5"), ("ce13a4", "ce13a6"), ("ce13a7", "ce13a2"), ("ce13a5", "ce13a1"),
             ("ce13a.msa", "ce13a1"), ("ce13a.msa", "ce13a17.fa"), ("ce13a10", "ce13a4")]
    n = 0
    for fa, fb in pairs:
        ga = R.group_file(os.path.join(pas, fa))
        gb = R.group_file(os.path.join(pas, fb))
        d = R.align_dump(ga, gb)
        if not ("a_pfq_pos" in d and "b_pfq_pos" in d and d["spb_fact"][0] != 0):
            print("(%s x %s: not annotated on both sides, skipped)" % (fa, fb))
            continue
        save("intron_%s_%s" % (fa.replace(".", "_"), fb.replace(".", "_")), d)
        n += 1
    assert n >= 4


def _rect_cases(R, tag, fam_kw, picks, weighted=True):
    from prrn_aln_amd.synth import make_family, tree_branches, tree_weights
    fam = make_family(**fam_kw)
    n = len(fam.msa)
    w = tree_weights(fam.tree, n) if weighted else None
    br = sorted(tree_branches(fam.tree), key=lambda b: min(len(b), n - len(b)))
    for k in picks:
        split_case(R, fam, br[k], w, "rect_%s_b%d" % (tag, k))


def job_rect_protein():
    """the rectangular engine: `-A` clears algmode.bnd, PwdM picks the _ALN modes and align2 runs alignC<recd_t>(..., rectangle = true)
    = Fwd2c::forwardA (reference src/fwd2c.h:232-356, initA :111-135; dispatch src/maln2.cc:1906-1910): every record type"""
    import refdump
    R = refdump.RefLib(molc=refdump.PROTEIN, band=False)
    _rect_cases(R, "syn20x90", dict(n_seq=20, length=90, seed=41, indel=0.03, max_indel=6), [0, 5, 12, -1, -3])       # HLF / RHF / GPF
    _rect_cases(R, "syn5x70", dict(n_seq=5, length=70, seed=42, indel=0.04), [0, 2, 4])                              # NTV
    _rect_cases(R, "syn6x80_nogap", dict(n_seq=6, length=80, seed=43, sub=0.25, indel=0.0), [0, 3])                   # NGP
    _rect_cases(R, "syn2x150", dict(n_seq=2, length=150, seed=44, sub=0.3, indel=0.04), [0], weighted=False)            # NGP, a pair
    _rect_cases(R, "syn40x60", dict(n_seq=40, length=60, seed=45, indel=0.03), [-1, -2])                              # balanced: sim33


def job_rect_ls3():
    """... with the double-affine penalty: forwardA's Vertical2 opens at v2divv1 + gop (fwd2c.h:276, a sum where forwardB multiplies)"""
    import refdump
    from prrn_aln_amd.synth import DNA
    R = refdump.RefLib(molc=refdump.PROTEIN, ls=3, band=False)
    _rect_cases(R, "prot16x100_ls3", dict(n_seq=16, length=100, seed=46, indel=0.04, max_indel=20), [0, 6, -1])
    _rect_cases(R, "prot4x90_ls3", dict(n_seq=4, length=90, seed=47, indel=0.05, max_indel=15), [0, 3])
    _rect_cases(R, "prot2x200_ls3", dict(n_seq=2, length=200, seed=49, sub=0.3, indel=0.05, max_indel=30), [0], weighted=False)   # NGP + Noll 3
    _rect_cases(R, "prot6x90_nogap_ls3", dict(n_seq=6, length=90, seed=50, sub=0.25, indel=0.0), [0, 3])


def job_rect_dna_ls3():
    import refdump
    from prrn_aln_amd.synth import DNA
    R = refdump.RefLib(molc=refdump.DNA, ls=3, band=False)
    _rect_cases(R, "dna12x150_ls3", dict(n_seq=12, length=150, seed=48, alphabet=DNA, indel=0.03, max_indel=40), [0, 4, -1])


JOBS = {"rect_protein": job_rect_protein, "rect_ls3": job_rect_ls3, "rect_dna_ls3": job_rect_dna_ls3, "intron": job_intron, "sim23": job_sim23, "protein": job_protein, "dna_ls3": job_dna_ls3, "protein_ls3": job_protein_ls3,
        "protein_tgapf": job_protein_tgapf, "w21_protein": job_w21_protein, "w21_dna_ls3": job_w21_dna_ls3}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1:
        JOBS[sys.argv[1]]()
    else:
        for j in JOBS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), j])
