"""Python mirror of the reference's operator surface for this path -- mSeq / PwdM / align2() -- on top of
level 1 of the C ABI (include/g2g.h).  Names and argument meaning follow the reference
(src/maln.h:276,342-344); the work is done by libg2g.so (host builders in C++, DP on the GPU)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._lib import G2GError, last_error, lib

PROTEIN, DNA = 1, 2

# residue codes, reference src/cmn.h:111-114: NIL,UNP,AMB,ALA,ARG,ASN,ASP,CYS,GLN,GLU,GLY,HIS,ILE,LEU,
# LYS,MET,PHE,PRO,SER,THR,TRP,TYR,VAL,ASX,GLX ; nucleotides ___,_,A,C,M,G,R,S,V,T,...
_AA = {c: i + 3 for i, c in enumerate("ARNDCQEGHILKMFPSTWYV")}
_AA.update({"B": 23, "Z": 24, "X": 2, "-": 1, ".": 1})
_NT = {"A": 2, "C": 3, "M": 4, "G": 5, "R": 6, "S": 7, "V": 8, "T": 9, "U": 9, "W": 10, "Y": 11, "H": 12,
       "K": 13, "D": 14, "B": 15, "N": 16, "-": 1, ".": 1}


def encode(rows: Sequence[str], molc: int) -> np.ndarray:
    """Aligned rows -> (len, many) uint8 residue codes ([pos][member], the layout of Seq::seq_)."""
    tab = np.full(256, 2 if molc == PROTEIN else 16, np.uint8)
    for k, v in (_AA if molc == PROTEIN else _NT).items():
        tab[ord(k)] = v
        tab[ord(k.lower())] = v
    arr = np.frombuffer("".join(rows).encode(), np.uint8).reshape(len(rows), -1)
    return np.ascontiguousarray(tab[arr].T)


@dataclass
class AlnParam:
    """Subset of the reference's ALPRM (src/seq.h:27-28) with prrn's effective defaults
    (PAM150 matrix, u=2, v=9, u1=0.6, k1=7, sh=-60, tgapf=1)."""
    u: float = 2.0
    v: float = 9.0
    u0: float = 0.0
    u1: float = 0.6
    tgapf: float = 1.0
    scale: float = 1.0
    gamma: float = 0.5
    k1: int = 7
    ls: int = 1
    sh: int = -60
    banded: int = 1
    molc: int = PROTEIN
    simmtx: Optional[np.ndarray] = None
    max_code: int = 25

    def to_c(self) -> Tuple[_abi.Params, np.ndarray]:
        sm = default_simmtx(self.molc) if self.simmtx is None else np.ascontiguousarray(self.simmtx, np.float64)
        f32 = lambda x: float(np.float32(x))      # ALPRM fields are floats in the reference
        p = _abi.Params()
        p.u, p.v, p.u0, p.u1 = f32(self.u), f32(self.v), f32(self.u0), f32(self.u1)
        p.tgapf, p.scale, p.gamma = f32(self.tgapf), f32(self.scale), f32(self.gamma)
        p.k1, p.ls, p.sh, p.banded, p.molc = self.k1, self.ls, self.sh, self.banded, self.molc
        p.simmtx = sm.ctypes.data_as(_abi.c_f64p)
        p.simrows, p.simdim = sm.shape
        p.max_code = self.max_code
        return p, sm


_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def default_simmtx(molc: int) -> np.ndarray:
    """The substitution matrix prrn effectively uses (data dumped from the reference's Simmtx::mtx:
    PAM150 for proteins, the +2/-2 style nucleotide matrix for DNA)."""
    return np.load(os.path.join(_DATA, "simmtx_protein.npy" if molc == PROTEIN else "simmtx_dna.npy"))


class mSeq:
    """<-> reference class mSeq (src/mseq.h:86): one group = aligned members + optional weights."""

    def __init__(self, rows_or_codes, alp: AlnParam, weight: Optional[Sequence[float]] = None):
        L = lib()
        codes = rows_or_codes if isinstance(rows_or_codes, np.ndarray) else encode(rows_or_codes, alp.molc)
        codes = np.ascontiguousarray(codes, np.uint8)
        self.len, self.many = codes.shape
        self._prm, self._sm = alp.to_c()
        w = None if weight is None else np.ascontiguousarray(weight, np.float64)
        self._h = L.g2g_group_create(None, C.byref(self._prm), self.many, self.len,
                                     codes.ctypes.data_as(_abi.c_u8p),
                                     None if w is None else w.ctypes.data_as(_abi.c_f64p))
        if not self._h:
            raise G2GError(last_error())

    def __del__(self):
        try:
            if self._h:
                lib().g2g_group_free(self._h)
                self._h = None
        except Exception:
            pass


class PwdM:
    """<-> reference class PwdM (src/maln.h:144, ctor src/maln2.cc:254): picks alnmode and scorer, swaps
    the pair so that the profile side is `a` (self.swp), builds thickness / vectors / gap profiles."""

    def __init__(self, seqs: Sequence[mSeq], alp: AlnParam):
        L = lib()
        self.seqs = list(seqs)
        self._prm, self._sm = alp.to_c()
        swp = C.c_int(0)
        self._h = L.g2g_pwdm_create(None, C.byref(self._prm), seqs[0]._h, seqs[1]._h, C.byref(swp))
        if not self._h:
            raise G2GError(last_error())
        self.swp = bool(swp.value)
        self.problem = L.g2g_pwdm_problem(self._h).contents
        self.alnmode = self.problem.alnmode

    @classmethod
    def batch(cls, ctx, pairs: Sequence[Sequence[mSeq]], alp: AlnParam) -> List["PwdM"]:
        """g2g_pwdm_create_batch: the PwdMs of many pairs at once, thickness / vectors / gap profiles of all their groups built
        on the DEVICE (csrc/g2g_build.hip); the same objects as [PwdM(p, alp) for p in pairs]."""
        L = lib()
        n = len(pairs)
        prm, sm = alp.to_c()
        ha = (C.c_void_p * max(1, n))(*[p[0]._h for p in pairs])
        hb = (C.c_void_p * max(1, n))(*[p[1]._h for p in pairs])
        swp = (C.c_int * max(1, n))()
        out = (C.c_void_p * max(1, n))()
        rc = L.g2g_pwdm_create_batch(ctx._h, C.byref(prm), n, ha, hb, swp, out)
        if rc:
            raise G2GError("g2g_pwdm_create_batch rc=%d: %s" % (rc, last_error()))
        res = []
        for k in range(n):
            o = cls.__new__(cls)
            o.seqs = list(pairs[k])
            o._prm, o._sm = prm, sm
            o._h = out[k]
            o.swp = bool(swp[k])
            o.problem = L.g2g_pwdm_problem(o._h).contents
            o.alnmode = o.problem.alnmode
            res.append(o)
        return res

    def __del__(self):
        try:
            if self._h:
                lib().g2g_pwdm_free(self._h)
                self._h = None
        except Exception:
            pass


def align2_batch(ctx, pwds: Sequence[PwdM]):
    """<-> align2(seqs, pwdm, &scr, GsI) for every PwdM of a sweep: [(scr, skl (n,2) int32, status)]."""
    L = lib()
    n = len(pwds)
    hs = (C.c_void_p * n)(*[p._h for p in pwds])
    scr = (C.c_double * n)()
    skl = (C.POINTER(_abi.Skl) * n)()
    nskl = (C.c_int * n)()
    st = (C.c_int * n)()
    rc = L.g2g_align2_batch(ctx._h, n, hs, scr, skl, nskl, st)
    if rc:
        raise G2GError("g2g_align2_batch rc=%d: %s" % (rc, last_error()))
    out = []
    for i in range(n):
        s = np.zeros((nskl[i], 2), np.int32)
        if nskl[i]:
            s[:] = np.ctypeslib.as_array(C.cast(skl[i], C.POINTER(C.c_int32)), shape=(nskl[i] * 2,)).reshape(-1, 2)
            L.g2g_free(skl[i])
        out.append((scr[i], s, st[i]))
    return out


def align2(ctx, pwd: PwdM):
    return align2_batch(ctx, [pwd])[0]


def align2_score_batch(ctx, pwds: Sequence[PwdM], cur_skls, flags: int = 0):
    """g2g_align2_score_batch: align2() of every PwdM plus calcSpScore of its current alignment (cur_skls) and of the new one --
    what a window of the refinement needs, on one resident batch.  [(scr, skl, status, (val, gap, status, raw) of the current
    alignment, the same of the new one)]."""
    L = lib()
    n = len(pwds)
    hs = (C.c_void_p * n)(*[p._h for p in pwds])
    _bufs, cptr, ccnt = _skl_arrays(cur_skls)
    scr = (C.c_double * n)()
    skl = (C.POINTER(_abi.Skl) * n)()
    nskl = (C.c_int * n)()
    st = (C.c_int * n)()
    fc = (_abi.Fstat * max(1, n))()
    fn = (_abi.Fstat * max(1, n))()
    rc = L.g2g_align2_score_batch(ctx._h, n, hs, cptr, ccnt, flags, scr, skl, nskl, st, fc, fn)
    if rc:
        raise G2GError("g2g_align2_score_batch rc=%d: %s" % (rc, last_error()))
    out = []
    for i in range(n):
        s = np.zeros((nskl[i], 2), np.int32)
        if nskl[i]:
            s[:] = np.ctypeslib.as_array(C.cast(skl[i], C.POINTER(C.c_int32)), shape=(nskl[i] * 2,)).reshape(-1, 2)
            L.g2g_free(skl[i])
        out.append((scr[i], s, st[i], (fc[i].val, fc[i].gap, fc[i].status, fc[i].raw), (fn[i].val, fn[i].gap, fn[i].status, fn[i].raw)))
    return out


def HomScore(ctx, pwd: PwdM):
    """<-> VTYPE HomScore(seqs, pwdm, rr) (reference src/maln2.cc:1837): (score, (rr0, rr1))."""
    L = lib()
    scr = C.c_double(0)
    rr = (C.c_int64 * 2)()
    rc = L.g2g_homscore(ctx._h, pwd._h, C.byref(scr), rr)
    if rc:
        raise G2GError("g2g_homscore rc=%d: %s" % (rc, last_error()))
    return scr.value, (rr[0], rr[1])


def _skl_arrays(skls):
    n = len(skls)
    bufs = []
    ptrs = (C.POINTER(_abi.Skl) * n)()
    cnt = (C.c_int * n)()
    for i, s in enumerate(skls):
        s = np.ascontiguousarray(s, np.int32).reshape(-1, 2)
        buf = (_abi.Skl * max(1, len(s)))()
        C.memmove(buf, s.ctypes.data, s.nbytes)
        bufs.append(buf)
        ptrs[i] = C.cast(buf, C.POINTER(_abi.Skl))
        cnt[i] = len(s)
    return bufs, ptrs, cnt


def spparams(pwd: PwdM) -> "_abi.SpParams":
    sp = _abi.SpParams()
    rc = lib().g2g_pwdm_spparams(pwd._h, C.byref(sp))
    if rc:
        raise G2GError("g2g_pwdm_spparams rc=%d" % rc)
    return sp


def calcSpScore_batch(ctx, pwds: Sequence[PwdM], skls, stats: bool = False):
    """<-> PreSpScore::calcSpScore(GsI) (reference src/fspscore.cc:584): [(fstat.val, fstat.gap, status, raw score)] of the
    alignments the standardised skeletons describe; with stats=True each tuple continues with (fstat.mch, fstat.mmc, fstat.unp)."""
    L = lib()
    n = len(pwds)
    hs = (C.c_void_p * n)(*[p._h for p in pwds])
    bufs, ptrs, cnt = _skl_arrays(skls)
    out = (_abi.Fstat * max(1, n))()
    rc = L.g2g_spscore_batch(ctx._h, n, hs, ptrs, cnt, out)
    if rc:
        raise G2GError("g2g_spscore_batch rc=%d: %s" % (rc, last_error()))
    if stats:
        return [(out[i].val, out[i].gap, out[i].status, out[i].raw, out[i].mch, out[i].mmc, out[i].unp) for i in range(n)]
    return [(out[i].val, out[i].gap, out[i].status, out[i].raw) for i in range(n)]
