"""f4: the reference's MSA file formats (prrn_aln_amd/seqio.py) against the reference's own sample files (copied as data under
tests/golden/pas/) and, where oracle/_ref is built, against the reference's own reader."""
import os

import numpy as np
import pytest

import refdump
from prrn_aln_amd import operator as op, seqio

PAS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pas")


def test_native_and_sequential_sample_files_hold_the_same_alignment():
    """sample/pas/native_A (interleaved, as prrn5 prints it) and sample/pas/Multi_A (sequential) are one alignment"""
    n1, r1 = seqio.read_msa(os.path.join(PAS, "native_A"))
    n2, r2 = seqio.read_msa(os.path.join(PAS, "Multi_A"))
    assert seqio.sniff(open(os.path.join(PAS, "native_A")).read()) == "native"
    assert seqio.sniff(open(os.path.join(PAS, "Multi_A")).read()) == "sequential"
    assert n1 == n2 and len(n1) == 3
    assert [r.rstrip("-") for r in r1] == [r.rstrip("-") for r in r2]
    assert len({len(r) for r in r2}) == 1 and len(r2[0]) == 162


@pytest.mark.parametrize("writer", ["sequential", "native", "fasta"])
def test_round_trip(tmp_path, writer):
    names, rows = seqio.read_msa(os.path.join(PAS, "Multi_B"))
    path = str(tmp_path / "x.msa")
    {"sequential": seqio.write_sequential, "native": seqio.write_native, "fasta": seqio.write_fasta}[writer](path, names, rows)
    n2, r2 = seqio.read_msa(path)
    assert n2 == names and [r.rstrip("-") for r in r2] == [r.rstrip("-") for r in rows]


@pytest.mark.skipif(not refdump.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("writer", ["sequential", "native"])
def test_reference_reads_what_we_write(tmp_path, writer):
    """a file written here, read by the REFERENCE's reader, gives the residue matrix the reference reads from its own file"""
    names, rows = seqio.read_msa(os.path.join(PAS, "Multi_A"))
    nb, rb = seqio.read_msa(os.path.join(PAS, "Multi_B"))
    path = str(tmp_path / "A.msa")
    {"sequential": seqio.write_sequential, "native": seqio.write_native}[writer](path, names, rows, "A")
    R = refdump.RefLib(molc=refdump.PROTEIN)
    gb = R.group_file(os.path.join(PAS, "Multi_B"))
    d_ref = R.align_dump(R.group_file(os.path.join(PAS, "Multi_A")), gb)
    gb2 = R.group_file(os.path.join(PAS, "Multi_B"))
    d_mine = R.align_dump(R.group_file(path), gb2)
    for k in ("a_seq", "b_seq", "scr", "align2_skl"):
        assert np.array_equal(d_ref[k], d_mine[k]), k
    # and our own encoding of the rows is the reference's residue matrix (positions 0..len-1 of [pos][member])
    many, length = int(d_ref["a_many"][0]), int(d_ref["a_len"][0])
    ref_codes = np.asarray(d_ref["a_seq"]).reshape(-1, many)[1:1 + length]
    mine = op.encode(rows, op.PROTEIN)
    keep = ref_codes > 1                                      # (the reference recodes terminal gaps; residues must agree)
    assert np.array_equal(ref_codes[keep], mine[keep])
