#!/usr/bin/env python3
"""Add the REFERENCE's own Ssrel::pairsum_ss read-outs (reference src/fspscore.cc:896-922; the -O4 line of src/prrn5.cc:1776-1793) to the
256 x 1024 refinement fixture tests/golden/refine_prot256x1024_prog.json.gz: every value the reference computed during
`prrn5 -YH0 -R1 -O4 start.msa` -- the start MSA (Prrn's initsp), the refined MSA on the tree the refinement ran on, and the two -O4
read-outs (unweighted / weighted) on the tree the reference rebuilt from the refined MSA -- each with the tree THAT call used,
traced by oracle/_ref/prrn5_trace (ld --wrap of pairsum_ss, oracle/ref_trace.cc).  Data only.

    python tools/make_refine_pairsum.py [trace.txt]       # without an argument: runs the reference (the better part of an hour)
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FIX = os.path.join(ROOT, "tests", "golden", "refine_prot256x1024_prog.json.gz")


def run_reference():
    import make_refine_golden as m
    import refdump
    ref = os.path.join(ROOT, "oracle", "_ref")
    rows = m.start_rows(m.BIG[0], {}, False)
    tmp = tempfile.mkdtemp()
    refdump.write_multi(os.path.join(tmp, "fam.msa"), ["s%03d" % i for i in range(len(rows))], list(rows), "fam")
    tr = os.path.join(tmp, "trace.txt")
    subprocess.run([os.path.join(ref, "prrn5_trace"), "-YH0", "-R1", "-O4", "fam.msa"], cwd=tmp,
                   env=dict(os.environ, ALN_TAB=os.path.join(ref, "table"), G2G_TRACE=tr), check=True, capture_output=True)
    return tr


def main():
    from prrn_aln_amd import operator as op
    tr = sys.argv[1] if len(sys.argv) > 1 else run_reference()
    f = json.load(gzip.open(FIX, "rt"))
    inv = {v: k for k, v in op._AA.items() if k.isupper() or k == "-"}
    start, final = list(f["rows"]), list(f["final_rows"])
    out, Q = [], []
    with open(tr) as fd:
        for l in fd:
            if l.startswith("Q "):
                Q.append(l.split())
            elif l.startswith("P "):
                head, rows = l.split("|")
                h = head.split()
                rows = ["".join(inv.get(ord(c) - 65, "X") for c in r) for r in rows.split()]
                assert len(rows) == int(h[3]) and all(len(r) == int(h[4]) for r in rows)
                which = "start" if rows == start else "final" if rows == final else None
                assert which, "a pairsum_ss call on an MSA that is neither the fixture's start nor its final MSA"
                assert [int(t[1]) for t in Q] == list(range(2 * len(rows) - 1))
                tree = {"left": [int(t[2]) for t in Q], "right": [int(t[3]) for t in Q], "parent": [int(t[4]) for t in Q],
                        "vol": [float(t[5]) for t in Q], "cur": [float(t[6]) for t in Q]}
                same_tree = tree == f["tree"]
                out.append({"msa": which, "use_pw": int(h[1]), "value": float(h[2]), "tree": None if same_tree else tree,
                            "tree_is": "the tree the refinement ran on" if same_tree else "rebuilt by the reference from this MSA (-O4 read-out)"})
                Q = []
    assert len(out) >= 2 and out[0]["msa"] == "start" and any(o["msa"] == "final" for o in out)
    f["pairsum_ss"] = out
    with gzip.open(FIX, "wt", compresslevel=9) as fd:
        json.dump(f, fd)
    for o in out:
        print("%-5s use_pw %d  %-60s %.17g" % (o["msa"], o["use_pw"], o["tree_is"], o["value"]))


if __name__ == "__main__":
    main()
