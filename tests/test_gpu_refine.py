"""f2 end to end on the GPU: prrn_aln_amd.refine.Refiner (divisions batched, DPs and calcSpScore on the GPU through the C
ABI) must walk the reference's own trajectory -- tests/golden/refine_*.json, traces of Prrn::rir taken by
oracle/_ref/prrn5_trace: the same branch sequence, bit-identical DP score and fstat.val at every align2(), the same accepted
moves (member lists and skeletons) and the same final MSA."""
import glob
import json
import os

import numpy as np
import pytest

from prrn_aln_amd import engine, operator as op
from prrn_aln_amd.refine import KTree, Refiner

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refine_*.json")))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context()
    yield c
    c.close()


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_refinement_walks_the_reference_trajectory(ctx, path):
    f = json.load(open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    r = Refiner(ctx, op.encode(f["rows"], f["molc"]), tree, alp, seed=1, maxitr=10, window=16)
    final = r.run()
    steps = r.steps
    assert [s.branch for s in steps] == f["branches"][:len(steps)] and len(steps) == len(f["branches"])
    dps = [s for s in steps if s.delta != float("-inf")]            # (divisions with nothing to re-align make no align2 call)
    assert len(dps) == len(f["align2"])
    for s, ref in zip(dps, f["align2"]):
        assert (max(s.na, s.nb), min(s.na, s.nb), int(s.swp)) == (max(ref["na"], ref["nb"]), min(ref["na"], ref["nb"]), ref["swp"])
        assert s.scr == ref["scr"] and s.val_new == ref["val"], (s.branch, s.scr, ref["scr"], s.val_new, ref["val"])
    acc = [s for s in steps if s.accepted]
    assert len(acc) == len(f["accepted"])
    for s, ref in zip(acc, f["accepted"]):
        assert list(s.lst[0]) == ref["lst0"] and list(s.lst[1]) == ref["lst1"]
        assert np.array_equal(s.skl, np.asarray(ref["skl"], np.int32))
    assert np.array_equal(final, op.encode(f["final_rows"], f["molc"]))
    assert r.batches < len(dps)                                       # the DPs really ran in batches


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-5] for p in FIX])
def test_native_refinement_walks_the_reference_trajectory(ctx, path):
    """The same loop in C++ behind the C ABI (g2g_refine): branch sequence, DP score and fstat.val at every align2(), the accepted
    moves and the final MSA of the reference's trace."""
    from prrn_aln_amd.refine import refine_native
    f = json.load(open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    final, steps, stats = refine_native(ctx, op.encode(f["rows"], f["molc"]), tree, alp, seed=1, maxitr=10, window=16)
    assert [s["branch"] for s in steps] == f["branches"]
    dps = [s for s in steps if not s["skipped"]]
    assert len(dps) == len(f["align2"])
    for s, ref in zip(dps, f["align2"]):
        assert (max(s["na"], s["nb"]), min(s["na"], s["nb"]), int(s["swp"])) == (max(ref["na"], ref["nb"]), min(ref["na"], ref["nb"]), ref["swp"])
        assert s["scr"] == ref["scr"] and s["val_new"] == ref["val"], (s["branch"], s["scr"], ref["scr"])
    assert sum(1 for s in steps if s["accepted"]) == len(f["accepted"]) == stats["accepted"]
    assert np.array_equal(final, op.encode(f["final_rows"], f["molc"]))
    assert stats["batches"] < len(dps)


def _native_rank(rank, world, port, path, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prrn_aln_amd import engine as eng
    from prrn_aln_amd.refine import refine_native, torch_exchange
    f = json.load(open(path))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    alp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    c = eng.Context(0)
    ex = torch_exchange()
    final, steps, stats = refine_native(c, op.encode(f["rows"], f["molc"]), tree, alp, seed=1, maxitr=10, window=16, exchange=ex)
    c.close()
    q.put((rank, final.tobytes(), final.shape, [s["branch"] for s in steps], stats))
    dist.destroy_process_group()


def test_native_refinement_sharded_over_two_ranks():
    """g2g_refine with an exchange callback (torch.distributed all_gather, gloo here; RCCL with GPU tensors): two ranks -- two
    processes sharing this box's GPU, each scoring half of every window -- end with the reference's final MSA, and neither
    scored every division."""
    import multiprocessing as mp
    path = [p for p in FIX if "prot12x80" in p][0]
    f = json.load(open(path))
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
