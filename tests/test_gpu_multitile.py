"""DPs that span several strips AND several column blocks of the tile kernels (rows > 32, columns > 512),
for every kernel variant (hf2, hf3, pf2, pf3) -- bit-exact against the oracle on the same flattened problem."""
import numpy as np
import pytest

import oraclelib
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family, DNA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


CASES = [
    ("prot_ls1", dict(n_seq=24, length=700, seed=31), op.AlnParam()),
    ("prot_ls3", dict(n_seq=20, length=650, seed=32, max_indel=25), op.AlnParam(ls=3)),
    ("dna_ls3", dict(n_seq=16, length=900, seed=33, alphabet=DNA, indel=0.015, max_indel=30), op.AlnParam(ls=3, molc=op.DNA, max_code=17)),
    ("prot_tgapf", dict(n_seq=12, length=600, seed=34), op.AlnParam(tgapf=0.5)),
]


@pytest.mark.parametrize("name,fam_kw,alp", CASES, ids=[c[0] for c in CASES])
def test_multitile_vs_oracle(ctx, name, fam_kw, alp):
    fam = make_family(**fam_kw)
    sw = sweep.Sweep(fam, alp)
    assert max(p.problem.b.len for p in sw.pwds) > 520 and max(p.problem.a.len for p in sw.pwds) > 64
    res = op.align2_batch(ctx, sw.pwds)
    L = oraclelib.load()
    modes = set()
    for pw, (scr, skl, st) in zip(sw.pwds, res):
        assert st == 0
        modes.add((pw.alnmode, pw.problem.noll))

        class H:
            c = pw.problem
        oscr, ocells, otr = oraclelib.forward(L, H)
        assert scr == oscr, (name, pw.alnmode, scr, oscr)
        assert np.array_equal(skl, oraclelib.stdskl(L, otr))
    assert len(modes) >= 2


def test_baseline_config2_shape(ctx):
    """BASELINE configs[1]: two groups of ~32 proteins x 512 aa (one 64-sequence family split at its most balanced
    tree branch), aligned once by align2 -- mode GPF_ALB (9), profile-profile DP; bit-exact vs the oracle."""
    fam = make_family(n_seq=64, length=512, seed=1)
    sw = sweep.Sweep(fam, op.AlnParam())
    k = min(range(len(sw)), key=lambda i: abs(len(sw.branches[i]) - 32))
    pw = sw.pwds[k]
    assert pw.alnmode == 9 and min(pw.problem.a.many, pw.problem.b.many) >= 16
    (scr, skl, st), = op.align2_batch(ctx, [pw])
    assert st == 0
    L = oraclelib.load()

    class H:
        c = pw.problem
    oscr, ocells, otr = oraclelib.forward(L, H)
    assert scr == oscr and ocells > 100000
    assert np.array_equal(skl, oraclelib.stdskl(L, otr))
