"""Level-1 host builders of libg2g.so (mode selection, exg_seq, mkthick, convseq, Gfq, stripe: SURVEY §8
rows a7-a9) against what the reference itself built for the same groups (arrays inside the goldens).
Exact equality: these arrays are doubles computed in the reference's operation order.  CPU only."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from prrn_aln_amd import _abi, operator as op

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def params_from_golden(d):
    molc = int(d["a_molc"][0])
    return op.AlnParam(u=float(d["alnprm_u"][0]), v=float(d["alnprm_v"][0]), u0=float(d["alnprm_u0"][0]),
                       u1=float(d["alnprm_u1"][0]), tgapf=float(d["alnprm_tgapf"][0]),
                       scale=float(d["alnprm_scale"][0]), gamma=float(d["alnprm_gamma"][0]),
                       k1=int(d["alnprm_k1"][0]), ls=int(d["alnprm_ls"][0]), sh=int(d["alnprm_sh"][0]),
                       banded=int(d["algmode_bnd"][0]), molc=molc, simmtx=d["simmtx"],
                       max_code=int(d["a_max_code"][0]))


def groups_from_golden(d, alp):
    """Undo the PwdM swap and the nil re-coding: what the caller handed to PwdM."""
    def raw(pfx):
        s = d[pfx + "seq"][1:-1].copy()
        s[s == 0] = 1                       # nil_code -> gap_code (exg_seq will redo it)
        w = d[pfx + "weight"] if (pfx + "weight") in d else None
        return s, w
    (sa, wa), (sb, wb) = raw("a_"), raw("b_")
    if int(d["swp"][0]):
        (sa, wa), (sb, wb) = (sb, wb), (sa, wa)
    return op.mSeq(sa, alp, wa), op.mSeq(sb, alp, wb)


def arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_builders_match_reference(path):
    d = dict(np.load(path))
    alp = params_from_golden(d)
    ga, gb = groups_from_golden(d, alp)
    pw = op.PwdM([ga, gb], alp)
    q = pw.problem
    assert pw.swp == bool(d["swp"][0])
    assert q.alnmode == d["alnmode"][0]
    assert q.sim2_kind == d["sim2_kind"][0]
    assert q.crg2_kind == d["crg2_kind"][0]
    assert q.noll == d["Noll"][0] and q.codonk1 == d["codonk1"][0]
    assert (q.lw, q.up) == (d["wdw_lw"][0], d["wdw_up"][0])
    assert q.basic_gop == d["Basic_GOP"][0] and q.weighted_gop == d["Weighted_GOP"][0]
    sp = op.spparams(pw)                               # scalars of PreSpScore::calcSpScore (f1)
    assert sp.vab == d["Vab"][0] and sp.basic_gep == d["BasicGEP"][0]
    assert sp.diffu == d["LongGEP"][0] - d["BasicGEP"][0]
    assert sp.diff_u == d["diff_u"][0]
    ref = _abi.problem_from_arrays(d).c
    assert q.u2divu1 == ref.u2divu1 and q.v2divv1 == ref.v2divv1 and q.u == ref.u
    for pfx, s in (("a_", q.a), ("b_", q.b)):
        g = lambda k: d[pfx + k]
        n = int(g("len")[0]); many = int(g("many")[0])
        assert (s.many, s.len, s.left, s.right) == (many, n, g("left")[0], g("right")[0])
        assert s.nils == g("nils")[0] and s.dels == g("dels")[0]
        assert np.array_equal(arr(s.seq, (n + 2) * many, np.uint8), g("seq").reshape(-1))
        assert np.array_equal(arr(s.thk, (n + 2) * 3, np.float64), g("thk").reshape(-1)), pfx + "thk"
        if (pfx + "pseq") in d and g("vect")[0]:
            assert s.nelm == g("nelm")[0] and s.felm == g("felm")[0]
            mine = arr(s.pseq, (n + 2) * s.nelm, np.float64).reshape(n + 2, s.nelm)
            want = g("pseq")
            keep = np.ones(s.nelm, bool)
            if int(g("molc")[0]) != 1 and s.nelm > s.felm + 1:
                # profile_n (mseq.cc:392-411) leaves the gap slot and the slots >= max_code of the
                # freshly new'ed vector unwritten (garbage in the reference, never read by a scorer)
                keep[s.felm + 1] = False
                keep[s.felm + int(g("max_code")[0]):s.nelm - 1] = False
            assert np.array_equal(mine[:, keep], want[:, keep]), pfx + "pseq"
        else:
            assert not s.pseq
        if (pfx + "hetero") in d:
            assert s.has_gfq and s.gfq.hetero == g("hetero")[0]
            for v, nm in enumerate(("sfq", "tfq", "rfq")):
                off = arr(s.gfq.off[v], n + 2, np.int32)
                assert np.array_equal(off, g(nm + "_off")), pfx + nm
                assert np.array_equal(arr(s.gfq.glen[v], off[-1], np.int32), g(nm + "_glen")), pfx + nm
                assert np.array_equal(arr(s.gfq.freq[v], off[-1], np.float64), g(nm + "_freq")), pfx + nm
        if q.crg2_kind:
            assert np.array_equal(arr(s.gapdens, (n + 2) * many, np.float64), g("gapdens").reshape(-1))
            assert np.array_equal(arr(s.postgapdens, (n + 2) * many, np.float64), g("postgapdens").reshape(-1))
    for key, ptr, m in (("wta", q.a.weight, q.a.many), ("wtb", q.b.weight, q.b.many)):
        if key in d:
            assert np.array_equal(arr(ptr, m, np.float64), d[key])
        else:
            assert not ptr
