#!/usr/bin/env python3
"""Generate tests/golden/refine_*.json: traces of the REFERENCE's own refinement loop (Prrn::rir, reference
src/prrn5.cc:633-666) on seeded synthetic families, taken by oracle/_ref/prrn5_trace (the reference's unmodified prrn5.cc
linked with oracle/ref_trace.cc: `ld --wrap` on Randiv's constructor / nextrandiv, align2 and synthgap).  Each fixture is
data only: the start MSA, the weighting tree the reference built (topology, Kirchhoff vol / cur per node), the branch
sequence, (DP score, fstat.val) of every align2() call, every accepted move (member lists + skeleton) and the final MSA.
Run in THIS container (needs /root/reference); the GPU box only sees the committed JSON files."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

# the headline workload itself (BASELINE.json configs[2]): 256 proteins x 1024 aa, refinement-only entry
# from the reference's own serial progressive MSA (tests/golden/msa/prog256x1024.npz, tools/make_progressive_fixture.py).  Takes
# the reference the better part of an hour on one core; the fixture is stored gzip-compressed.
BIG = ("prot256x1024_prog", os.path.join("msa", "prog256x1024.npz"))

CASES = [
    ("prot12x80_s3", dict(n_seq=12, length=80, seed=3, indel=0.03, max_indel=6), False, []),
    ("prot20x100_s11", dict(n_seq=20, length=100, seed=11, indel=0.03, max_indel=8), False, []),
    ("dna16x120_s8_ls3", dict(n_seq=16, length=120, seed=8, indel=0.03, max_indel=8), True, ["-yl3"]),
    ("prot48x300_s5", dict(n_seq=48, length=300, seed=5, indel=0.03, max_indel=8), False, []),      # the size INTEGRATION.md times
]


def parse_msa(text):
    rows = {}
    order = []
    for line in text.splitlines():
        m = re.match(r"^\s*\d+ (.{60})\| (\S+)\s*$", line)
        if m:
            if m.group(2) not in rows:
                rows[m.group(2)] = ""
                order.append(m.group(2))
            rows[m.group(2)] += m.group(1)
    return [rows[k].rstrip() for k in order], order


def start_rows(name, kw, dna):
    """rows of the start MSA: a seeded synthetic family's true alignment, or (BIG) the reference's progressive MSA of the bench family"""
    from prrn_aln_amd.synth import DNA, make_family
    if name == BIG[0]:
        import numpy as np
        from prrn_aln_amd import operator as op
        codes = np.load(os.path.join(GOLD, BIG[1]))["codes"]
        inv = {v: k for k, v in op._AA.items() if k.isupper() or k == "-"}
        return ["".join(inv.get(int(c), "X") for c in codes[:, j]) for j in range(codes.shape[1])]
    if dna:
        kw = dict(kw, alphabet=DNA)
    return list(make_family(**kw).msa)


def build_fixture(name, rows, dna, extra, opts, L, out, ref_seconds):
    """the fixture dict from the trace lines L, the reference's printed final alignment `out` and its wall time"""
    names = [("s%03d" if len(rows) > 99 else "s%02d") % i for i in range(len(rows))]
    final, order = parse_msa(out)
    assert order == names and len({len(r) for r in final}) <= 2, (order[:3], [len(r) for r in final][:5])
    width = max(len(r) for r in final)
    final = [r.ljust(width, "-").replace(" ", "-") for r in final]
    T = [l.split() for l in L if l.startswith("T ")]
    fix = {
        "name": name, "molc": 2 if dna else 1, "ls": 3 if "-yl3" in extra else 1, "options": opts,
        "rows": list(rows),
        "tree": {"left": [int(t[2]) for t in T], "right": [int(t[3]) for t in T], "parent": [int(t[4]) for t in T],
                 "vol": [float(t[5]) for t in T], "cur": [float(t[6]) for t in T]},
        "cycle": int([l for l in L if l.startswith("C ")][0].split()[1]),
        "branches": [int(l.split()[1]) for l in L if l.startswith("D ")],
        "align2": [], "accepted": [], "final_rows": final,
        "reference_seconds": round(ref_seconds, 1),
    }
    assert [int(t[1]) for t in T] == list(range(len(T)))
    for l in L:
        if l.startswith("A "):
            h = l[:400].split("|")[0].split()
            fix["align2"].append({"na": int(h[1]), "nb": int(h[2]), "swp": int(h[3]), "scr": float(h[4]), "val": float(h[5])})
        elif l.startswith("S "):
            p = l.split("|")
            sk = [int(x) for x in p[3].split()]
            mv = {"lst0": [int(x) for x in p[1].split()], "lst1": [int(x) for x in p[2].split()],
                  "skl": [sk[i:i + 2] for i in range(0, len(sk), 2)]}
            if name == BIG[0]:
                # 834 moves of a 256-member family: the member list of the smaller group (the larger is its complement) and a
                # checksum of the skeleton (count of corners + CRC-32 of the int32 corner array) instead of ~4000 numbers per move
                import zlib
                import numpy as np
                mv = {"lst1": mv["lst1"], "ncorners": len(mv["skl"]), "skl_crc32": zlib.crc32(np.asarray(mv["skl"], np.int32).tobytes())}
            fix["accepted"].append(mv)
    return fix, width


def write_fixture(name, fix):
    import gzip
    if name == BIG[0]:
        with gzip.open(os.path.join(GOLD, "refine_%s.json.gz" % name), "wt", compresslevel=9) as fd:
            json.dump(fix, fd)
    else:
        json.dump(fix, open(os.path.join(GOLD, "refine_%s.json" % name), "w"))


def main(only=None, reuse=None):
    """reuse = (trace file, file with the reference's printed output, seconds): assemble the fixture from a run made earlier (the
    256 x 1024 case takes the reference 45 minutes traced and as long again plain)"""
    import time
    import refdump
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"))
    cases = CASES + ([(BIG[0], {}, False, [])] if only == BIG[0] else [])
    for name, kw, dna, extra in cases:
        if only and name != only:
            continue
        rows = start_rows(name, kw, dna)
        names = [("s%03d" if len(rows) > 99 else "s%02d") % i for i in range(len(rows))]
        opts = ["-YH0", "-R1"] + extra
        if reuse:
            L = [l.rstrip("\n") for l in open(reuse[0])]
            out = open(reuse[1]).read()
            ref_seconds = float(reuse[2])
        else:
            with tempfile.TemporaryDirectory() as tmp:
                refdump.write_multi(os.path.join(tmp, "fam.msa"), names, list(rows), "fam")
                tr = os.path.join(tmp, "trace.txt")
                subprocess.run([os.path.join(REF, "prrn5_trace")] + opts + ["-O4", "fam.msa"], cwd=tmp, env=dict(env, G2G_TRACE=tr),
                               check=True, capture_output=True)
                t0 = time.time()
                out = subprocess.run([os.path.join(REF, "prrn5")] + opts + ["fam.msa"], cwd=tmp, env=env, check=True,
                                     capture_output=True, text=True).stdout
                ref_seconds = time.time() - t0          # the reference's own serial refinement, one core of the build container
                L = [l.rstrip("\n") for l in open(tr)]
        fix, width = build_fixture(name, rows, dna, extra, opts, L, out, ref_seconds)
        write_fixture(name, fix)
        print("%-22s members %d, cycle %d, %d divisions drawn, %d align2 calls, %d accepted, %d -> %d columns" % (
            name, len(names), fix["cycle"], len(fix["branches"]), len(fix["align2"]), len(fix["accepted"]), len(rows[0]), width))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--reuse":
        main(sys.argv[1], sys.argv[3:6])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else None)
