"""f2 end to end on the GPU: g2g_refine (C++ behind the C ABI: windows of speculative divisions, DPs and calcSpScore on the
GPU, in-order acceptance) must walk the reference's own trajectory -- tests/golden/refine_*.json(.gz), traces of Prrn::rir
(reference src/prrn5.cc:633-666) taken by oracle/_ref/prrn5_trace: the same branch sequence, bit-identical DP score and
fstat.val at every align2(), the same accepted moves (member lists and skeletons) and the same final MSA.  The largest trace
is BASELINE.json's headline workload itself: 256 proteins x 1024 aa from the reference's progressive MSA (modes 6, 7, 9 and 10
all occur in it)."""
import os

import numpy as np
import pytest

import refinelib
from prrn_aln_amd import engine, operator as op
from prrn_aln_amd.refine import refine_native

pytestmark = pytest.mark.gpu
FIX = refinelib.fixtures()


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.mark.parametrize("path", FIX, ids=[refinelib.fixture_id(p) for p in FIX])
def test_native_refinement_walks_the_reference_trajectory(ctx, path):
    f, tree, alp, start = refinelib.load(path)
    final, steps, stats = refine_native(ctx, start, tree, alp, seed=1, maxitr=10, window=16, want_moves=True)
    refinelib.check_against_trace(f, final, steps, stats)
    assert stats["batches"] < len(f["align2"])                          # the DPs really ran in batches
    # every run is evidence about the scheduler's waits: an ordinary run has none that gave up
    print("wait_timeouts %d, recovered_dps %d %s" % (stats["wait_timeouts"], stats["recovered_dps"], ctx.last_timeout() if stats["recovered_dps"] else ""))


def test_current_alignment_walks_beside_the_dps(ctx):
    """Option SP_OVERLAP: calcSpScore of the windows' CURRENT alignments runs on a stream of its own beside the DP kernels
    (g2g_batch_spscore_begin / _end; off by default -- measured slower).  Same trajectory, bit for bit."""
    path = [p for p in FIX if "256x1024" not in p][0]
    f, tree, alp, start = refinelib.load(path)
    ctx.set_option("SP_OVERLAP", 1)
    try:
        final, steps, stats = refine_native(ctx, start, tree, alp, seed=1, maxitr=10, window=16, want_moves=True)
    finally:
        ctx.reset_options()
    refinelib.check_against_trace(f, final, steps, stats)


def _native_rank(rank, world, port, path, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prrn_aln_amd import engine as eng
    from prrn_aln_amd.refine import refine_native, torch_exchange
    f, tree, alp, start = refinelib.load(path)
    c = eng.Context(0)
    ex = torch_exchange()
    final, steps, stats = refine_native(c, start, tree, alp, seed=1, maxitr=10, window=16, exchange=ex)
    c.close()
    q.put((rank, final.tobytes(), final.shape, [s["branch"] for s in steps], stats))
    dist.destroy_process_group()


def test_native_refinement_sharded_over_two_ranks():
    """g2g_refine with an exchange callback (torch.distributed all_gather, gloo here; RCCL with GPU tensors): two ranks -- two
    processes sharing this box's GPU, each scoring half of every window -- end with the reference's final MSA, and neither
    scored every division."""
    import multiprocessing as mp
    path = [p for p in FIX if "prot12x80" in p][0]
    f = refinelib.load(path)[0]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [mpc.Process(target=_native_rank, args=(r, 2, port, path, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = op.encode(f["final_rows"], f["molc"])
    total = 0
    for rank, raw, shape, branches, stats in got:
        assert np.array_equal(np.frombuffer(raw, np.uint8).reshape(shape), want)
        assert branches == f["branches"]
        total += stats["divisions_scored_here"]
        assert 0 < stats["divisions_scored_here"]
    assert got[0][4]["divisions_scored_here"] != total                 # the work really was split
