"""The drop-in claim, end to end: the reference's own programs (`aln`, `prrn5`) linked with integration/g2g_bind.cc
-- align2() (reference src/maln2.cc:1875) routed to libg2g.so by `ld --wrap`, nothing of the reference edited --
must print exactly what the unmodified programs print, with the DPs actually running on the GPU.

The binaries are built by `make -f oracle/Makefile.ref` (into oracle/_ref/, which travels to the GPU box but is
not in git); the tests skip when they are absent.  Inputs are synthetic families written in the reference's
sequential multi-sequence format."""
import os
import re
import subprocess

import pytest

import refdump
from prrn_aln_amd.synth import DNA, drop_common_gaps, make_family, tree_branches

REF = refdump.REF_DIR
BIN = {k: os.path.join(REF, k) for k in ("aln", "aln_g2g", "prrn5", "prrn5_g2g")}
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not all(os.path.exists(p) for p in BIN.values()) or not refdump.available(),
                                 reason="oracle/_ref drop-in binaries not built (make -f oracle/Makefile.ref)")]


def _run(exe, args, cwd, mode=None, **extra):
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"), G2G_BIND_STATS="1", **extra)
    if mode:
        env["G2G_BIND"] = mode
    p = subprocess.run([BIN[exe]] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout, p.stderr


def _stats(err):
    m = re.search(r"g2g_bind: (\d+) align2 calls, (\d+) on the GPU, (\d+) by the reference, (\d+) mismatches", err)
    assert m, err[-2000:]
    return tuple(int(x) for x in m.groups())


def _family(tmp, seed, n_seq, length, dna=False, **kw):
    a = dict(n_seq=n_seq, length=length, seed=seed, indel=0.03, max_indel=8)
    a.update(kw)
    if dna:
        a["alphabet"] = DNA
    fam = make_family(**a)
    names = ["s%03d" % i for i in range(len(fam.msa))]
    refdump.write_multi(os.path.join(tmp, "fam.msa"), names, list(fam.msa), "fam")
    return fam, names


@pytest.mark.parametrize("seed,n_seq,length", [(5, 24, 150), (6, 40, 90), (7, 12, 200)])
def test_aln_group_pair_identical_output(tmp_path, seed, n_seq, length):
    tmp = str(tmp_path)
    fam, names = _family(tmp, seed, n_seq, length)
    n = len(fam.msa)
    done = 0
    for side in sorted(tree_branches(fam.tree), key=lambda s: -min(len(s), n - len(s)))[:4]:
        sa = set(side)
        A = [i for i in range(n) if i in sa]
        B = [i for i in range(n) if i not in sa]
        refdump.write_multi(os.path.join(tmp, "GA"), [names[i] for i in A], drop_common_gaps([fam.msa[i] for i in A]), "GA")
        refdump.write_multi(os.path.join(tmp, "GB"), [names[i] for i in B], drop_common_gaps([fam.msa[i] for i in B]), "GB")
        ref_out, _ = _run("aln", ["-s", tmp, "GA", "GB"], tmp)
        out, err = _run("aln_g2g", ["-s", tmp, "GA", "GB"], tmp)
        calls, gpu, cpu, bad = _stats(err)
        assert gpu >= 1 and cpu == 0 and bad == 0, err[-500:]
        assert out == ref_out
        done += 1
    assert done


@pytest.mark.parametrize("seed,n_seq,length,dna,extra", [(5, 24, 150, False, []), (8, 16, 120, True, []),
                                                         (8, 16, 120, True, ["-yl3"]), (12, 20, 140, False, ["-yl3"])])
def test_prrn5_refinement_identical_output(tmp_path, seed, n_seq, length, dna, extra):
    """Refinement of a pre-aligned family: every align2() of Prrn::onecycle runs on the GPU; the refined MSA and
    the sum-of-pairs line are those of the unmodified prrn5.  `-yl3`: double-affine gap penalty (Noll 3 kernels)."""
    tmp = str(tmp_path)
    _family(tmp, seed, n_seq, length, dna=dna)
    for opts in (["-YH0", "-R1"] + extra, ["-YH0", "-R1", "-O4"] + extra):
        ref_out, _ = _run("prrn5", opts + ["fam.msa"], tmp)
        out, err = _run("prrn5_g2g", opts + ["fam.msa"], tmp)
        calls, gpu, cpu, bad = _stats(err)
        assert calls > 50 and gpu > 0 and bad == 0, err[-500:]
        assert gpu >= 0.9 * calls, (calls, gpu, cpu)       # (what stays on the CPU: empty ranges, modes off the path)
        assert out == ref_out


def test_prrn5_verify_mode_counts_no_mismatch(tmp_path):
    """G2G_BIND=verify: both implementations run on every call; scores, skeletons and fstat.val must agree."""
    tmp = str(tmp_path)
    _family(tmp, 11, 20, 100)
    out, err = _run("prrn5_g2g", ["-YH0", "-R1", "-O4", "fam.msa"], tmp, mode="verify")
    calls, gpu, cpu, bad = _stats(err)
    assert gpu > 0 and bad == 0, err[-800:]


def test_prrn5_threaded_calls_are_batched(tmp_path):
    """prrn5 -t8: thread_onecycle (reference src/prrn5.cc:565-592) calls align2 from 8 pthreads; the binding runs the calls
    that arrive together as one g2g_forward_batch.  Output must equal that of the unmodified threaded program."""
    tmp = str(tmp_path)
    _family(tmp, 9, 48, 200)
    opts = ["-YH0", "-R1", "-O4", "-t8", "fam.msa"]
    ref_out, _ = _run("prrn5", opts, tmp)
    out, err = _run("prrn5_g2g", opts, tmp, G2G_BIND_QUIET_US="5000")     # (a generous gather window: the assertion on
    calls, gpu, cpu, bad = _stats(err)                                        #  batch sizes must not depend on host load)
    m = re.search(r"(\d+) GPU batches, largest (\d+)", err)
    assert m, err[-500:]
    batches, largest = int(m.group(1)), int(m.group(2))
    assert gpu >= 0.9 * calls and bad == 0
    assert largest > 1 and batches < gpu, (batches, largest, gpu)
    assert out == ref_out


def test_aln_intron_annotated_pair_on_the_gpu():
    """BASELINE configs[0]: `aln -s sample/pas ce13a1 ce13a2` (tests/golden/pas/ holds the two data files of the
    reference's sample/).  Both inputs carry `;C join(...)` exon annotations, so the intron-position bonus
    (PfqItr::match_score, reference src/fwd2c.h:446-452) is live.  Since ABI 3 the flattened problem carries the exon-boundary
    lists and the DP runs on the GPU: Score = 2325.0, byte-identical output, the DP counted on the GPU; `verify` finds no
    mismatch; G2G_BIND_NO_INTRON=1 still routes the pair to the reference's own forwardB."""
    pas = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pas")
    ref_out, _ = _run("aln", ["-s", pas, "ce13a1", "ce13a2"], pas)
    out, err = _run("aln_g2g", ["-s", pas, "ce13a1", "ce13a2"], pas)
    calls, gpu, cpu, bad = _stats(err)
    assert calls == 1 and gpu == 1 and cpu == 0 and bad == 0, err[-500:]
    assert "Score = 2325.0" in out
    assert out == ref_out
    out, err = _run("aln_g2g", ["-s", pas, "ce13a1", "ce13a2"], pas, mode="verify")
    assert _stats(err)[1] == 1 and _stats(err)[3] == 0, err[-500:]
    out, err = _run("aln_g2g", ["-s", pas, "ce13a1", "ce13a2"], pas, G2G_BIND_NO_INTRON="1")
    calls, gpu, cpu, bad = _stats(err)
    assert gpu == 0 and cpu == 1 and out == ref_out


def test_aln_rectangular_engine_pair_on_the_gpu(tmp_path):
    """`aln -A` (algmode.bnd = 0: PwdM picks the _ALN modes, align2 runs Fwd2c::forwardA, reference src/fwd2c.h:232-356): two ungapped
    single sequences make an NGP_ALN pair, which the binding sends to g2g_forward_kernel's rectangle mode -- byte-identical output;
    a pair of gapped groups (HLF / GPF_ALN: gap state) stays with the reference's own code."""
    tmp = str(tmp_path)
    fam = make_family(n_seq=2, length=180, seed=91, sub=0.3, indel=0.05, max_indel=12)
    rows = [r.replace("-", "") for r in fam.msa]
    refdump.write_multi(os.path.join(tmp, "SA"), ["sa"], [rows[0]], "SA")
    refdump.write_multi(os.path.join(tmp, "SB"), ["sb"], [rows[1]], "SB")
    ref_out, _ = _run("aln", ["-A", "-s", tmp, "SA", "SB"], tmp)
    out, err = _run("aln_g2g", ["-A", "-s", tmp, "SA", "SB"], tmp)
    calls, gpu, cpu, bad = _stats(err)
    assert calls == 1 and gpu == 1 and cpu == 0 and bad == 0, err[-500:]
    assert out == ref_out
    out, err = _run("aln_g2g", ["-A", "-s", tmp, "SA", "SB"], tmp, mode="verify")
    assert _stats(err)[1] == 1 and _stats(err)[3] == 0, err[-500:]
    fam, names = _family(tmp, 92, 10, 80)
    A, B = [0, 1, 2, 3], [4, 5, 6, 7, 8, 9]
    refdump.write_multi(os.path.join(tmp, "GA"), [names[i] for i in A], drop_common_gaps([fam.msa[i] for i in A]), "GA")
    refdump.write_multi(os.path.join(tmp, "GB"), [names[i] for i in B], drop_common_gaps([fam.msa[i] for i in B]), "GB")
    ref_out, _ = _run("aln", ["-A", "-s", tmp, "GA", "GB"], tmp)
    out, err = _run("aln_g2g", ["-A", "-s", tmp, "GA", "GB"], tmp)
    calls, gpu, cpu, bad = _stats(err)
    assert gpu == 0 and cpu == calls and out == ref_out, err[-500:]


def _dstats(err):
    m = re.search(r"g2g_bind: (\d+) alnScoreD calls, (\d+) on the GPU, (\d+) by the reference, (\d+) mismatches; (\d+) GPU batches, largest (\d+)", err)
    assert m, err[-2000:]
    return tuple(int(x) for x in m.groups())


@pytest.mark.parametrize("threads", [[], ["-t8"]], ids=["serial", "t8"])
def test_prrn5_from_unaligned_sequences_guide_tree_on_gpu(tmp_path, threads):
    """The whole program from unaligned sequences: the distance matrix (dpscore -> alnScoreD, one call per pair of members),
    the progressive alignment and the refinement.  With the binding every alnScoreD and align2 runs on the GPU (f3 + a1..a9);
    the output must be the unmodified program's, byte for byte."""
    tmp = str(tmp_path)
    fam = make_family(n_seq=14, length=120, seed=21, indel=0.03, max_indel=8)
    with open(os.path.join(tmp, "fam.fa"), "w") as fd:
        for i, r in enumerate(fam.msa):
            fd.write(">s%03d\n%s\n" % (i, r.replace("-", "")))
    opts = ["-R1", "-O4"] + threads + ["fam.fa"]
    ref_out, _ = _run("prrn5", opts, tmp)
    out, err = _run("prrn5_g2g", opts, tmp, G2G_BIND_QUIET_US="3000")
    dcalls, dgpu, dcpu, dbad, dbatches, dlargest = _dstats(err)
    assert dcalls >= 14 * 13 // 2 and dgpu == dcalls and dbad == 0, err[-600:]
    calls, gpu, cpu, bad = _stats(err)
    assert gpu > 0 and bad == 0
    assert out == ref_out
    if threads:
        assert dlargest > 1 and dbatches < dgpu, (dbatches, dlargest, dgpu)
    out, err = _run("prrn5_g2g", opts, tmp, mode="verify")
    assert _dstats(err)[3] == 0 and _stats(err)[3] == 0, err[-600:]
