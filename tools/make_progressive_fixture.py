#!/usr/bin/env python3
"""tests/golden/msa/prog256x1024.npz: the REFERENCE's own progressive alignment of the bench family (256 proteins x 1024 aa,
prrn_aln_amd/synth.py seed 1) -- `prrn5 -YH0 -S0 seqs.fa`, i.e. guide forest + progressive alignment, no refinement -- as
the start MSA of bench.py's second workload (refinement starts from the serial progressive MSA,
which is shorter and rougher than the synthetic true alignment).  Data only: residue codes, members in family order.
Run in THIS container (needs /root/reference and oracle/_ref); ~40 s."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")


def main():
    from prrn_aln_amd import operator as op
    from prrn_aln_amd.synth import make_family
    fam = make_family(256, 1024, 1)
    with tempfile.TemporaryDirectory() as tmp:
        tab = os.path.join(tmp, "table")
        os.makedirs(tab)
        for d in ("/root/reference/table", os.path.join(REF, "table")):     # the reference's tables + the generated mdm_mtx
            for f in os.listdir(d):
                dst = os.path.join(tab, f)
                if not os.path.lexists(dst):
                    os.symlink(os.path.join(d, f), dst)
        with open(os.path.join(tmp, "seqs.fa"), "w") as fd:
            for i, r in enumerate(fam.msa):
                s = r.replace("-", "")
                fd.write(">s%03d\n" % i)
                for k in range(0, len(s), 60):
                    fd.write(s[k:k + 60] + "\n")
        out = subprocess.run([os.path.join(REF, "prrn5"), "-YH0", "-S0", "seqs.fa"], cwd=tmp, env=dict(os.environ, ALN_TAB=tab),
                             check=True, capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"^\s*\d+ (.{60})\| (\S+)\s*$", line)
        if m:
            rows[m.group(2)] = rows.get(m.group(2), "") + m.group(1)
    names = ["s%03d" % i for i in range(256)]
    assert sorted(rows) == names
    width = max(len(r.rstrip()) for r in rows.values())
    msa = [rows[n].rstrip().ljust(width, "-").replace(" ", "-") for n in names]
    for r, t in zip(msa, fam.msa):
        assert r.replace("-", "") == t.replace("-", "")
    codes = op.encode(msa, op.PROTEIN)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "msa", "prog256x1024.npz"), codes=codes)
    print("progressive MSA: %d members x %d columns (true alignment: %d columns)" % (codes.shape[1], codes.shape[0], len(fam.msa[0])))


if __name__ == "__main__":
    main()
