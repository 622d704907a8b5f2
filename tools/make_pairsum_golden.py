#!/usr/bin/env python3
"""Generate tests/golden/pairsum/*.json from the REAL reference: values of Ssrel::pairsum_ss (reference src/fspscore.cc:896-922), the
whole-MSA sum-of-pairs score prrn reports, traced by oracle/_ref/prrn5_trace (ld --wrap, oracle/ref_trace.cc) together with the
weighting tree and the MSA each value was computed on.  Data only.  Usage: python tools/make_pairsum_golden.py"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden", "pairsum")
REF = os.path.join(ROOT, "oracle", "_ref")

CASES = [
    ("prot12x80", dict(n_seq=12, length=80, seed=3), False, []),
    ("prot20x100", dict(n_seq=20, length=100, seed=11), False, []),
    ("dna16x120_ls3", dict(n_seq=16, length=120, seed=8), True, ["-yl3"]),
    ("prot40x60_ls3", dict(n_seq=40, length=60, seed=21, indel=0.04, max_indel=9), False, ["-yl3"]),
    ("prot72x50", dict(n_seq=72, length=50, seed=31), False, []),                 # more members than ndesc_thr = 60: the tree recursion
    ("prot66x40_ls3", dict(n_seq=66, length=40, seed=33), False, ["-yl3"]),
    # (DNA families of more than 60 members make the reference itself abort in this stage: not a fixture)
]


def main():
    import refdump
    from prrn_aln_amd.synth import DNA, make_family
    os.makedirs(GOLD, exist_ok=True)
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"))
    for name, kw, dna, extra in CASES:
        if dna:
            kw = dict(kw, alphabet=DNA)
        fam = make_family(**kw)
        names = ["s%02d" % i for i in range(len(fam.msa))]
        with tempfile.TemporaryDirectory() as tmp:
            refdump.write_multi(os.path.join(tmp, "fam.msa"), names, list(fam.msa), "fam")
            tr = os.path.join(tmp, "trace.txt")
            subprocess.run([os.path.join(REF, "prrn5_trace"), "-YH0", "-R1", "-I1", "-O4"] + extra + ["fam.msa"], cwd=tmp,
                           env=dict(env, G2G_TRACE=tr), check=True, capture_output=True)
            L = [l.rstrip("\n") for l in open(tr)]
        cases, Q = [], []
        for l in L:
            if l.startswith("Q "):
                Q.append(l.split())
                continue
            if not l.startswith("P "):
                continue
            head, rows = l.split("|")
            h = head.split()
            rows = rows.split()
            assert len(rows) == int(h[3]) and all(len(r) == int(h[4]) for r in rows)
            assert [int(t[1]) for t in Q] == list(range(2 * len(rows) - 1))
            tree = {"left": [int(t[2]) for t in Q], "right": [int(t[3]) for t in Q], "parent": [int(t[4]) for t in Q],
                    "vol": [float(t[5]) for t in Q], "cur": [float(t[6]) for t in Q]}
            cases.append({"use_pw": int(h[1]), "value": float(h[2]), "rows": rows, "tree": tree})
            Q = []
        assert cases
        fix = {"name": name, "molc": 2 if dna else 1, "ls": 3 if "-yl3" in extra else 1, "cases": cases}
        json.dump(fix, open(os.path.join(GOLD, name + ".json"), "w"))
        print("%-18s %d members, %d pairsum_ss values: %s" % (name, len(names), len(cases), ", ".join("%.3f" % c["value"] for c in cases[:4])))


if __name__ == "__main__":
    main()
