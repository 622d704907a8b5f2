"""f1 on the GPU: g2g_pairsum (Ssrel::pairsum_ss) against the values traced out of the reference (tests/golden/pairsum): families of
up to 60 members (the naive branch: g2g_pairsum_kernel) and of more (the tree recursion: naive sub-trees + calcscore_grp joins through
the level-1 builders and g2g_batch_spscore), weighted and unweighted, Noll 2 and 3."""
import glob
import json
import os

import pytest

import pairsumlib
from prrn_aln_amd import engine
from prrn_aln_amd.refine import KTree, pairsum

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pairsum", "*.json")))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-5] for p in FIX])
def test_pairsum_matches_reference(ctx, path):
    f = json.load(open(path))
    alp = pairsumlib.alp_of(f)
    for c in f["cases"]:
        t = c["tree"]
        tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
        got = pairsum(ctx, pairsumlib.case_codes(c), tree, alp, use_pw=bool(c["use_pw"]))
        assert got == c["value"], (f["name"], c["use_pw"], got, c["value"])


def test_pairsum_of_the_256_member_bench_family_matches_the_reference(ctx):
    """the headline fixture itself: every Ssrel::pairsum_ss value the reference computed in `prrn5 -YH0 -R1 -O4` on the 256 x 1024 aa
    family (tools/make_refine_pairsum.py): the start MSA and the refined MSA on the refinement's tree, and the two -O4 read-outs
    (unweighted, weighted) on the tree the reference rebuilt from the refined MSA -- 256 members, four levels of the tree recursion"""
    import gzip
    from prrn_aln_amd import operator as op
    f = json.load(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "refine_prot256x1024_prog.json.gz"), "rt"))
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25)
    msa = {"start": op.encode(f["rows"], f["molc"]), "final": op.encode(f["final_rows"], f["molc"])}
    assert len(f["pairsum_ss"]) == 4 and len(f["rows"]) == 256
    for c in f["pairsum_ss"]:
        t = c["tree"] or f["tree"]
        tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
        got = pairsum(ctx, msa[c["msa"]], tree, alp, use_pw=bool(c["use_pw"]))
        assert got == c["value"], (c["msa"], c["use_pw"], c["tree_is"], got, c["value"])
