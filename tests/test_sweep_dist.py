"""The N>1 path of a sweep on CPU: divisions dealt to ranks, result slots packed, all-gathered over gloo
(world_size 2) and unpacked -- every rank must end up with every division's result.  The DP itself needs a
GPU, so results here are synthetic skeletons; the sharding / slot / gather logic is the code bench.py runs."""
import os
import socket
import subprocess
import sys

import numpy as np

from prrn_aln_amd import sweep

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from prrn_aln_amd import sweep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, cap = 23, 64
rng = np.random.RandomState(5)
cells = rng.randint(10, 1000, size=n)
order = np.argsort(-cells, kind="stable")
mine = sweep.shard(order, world, rank)
def fake(k):      # what the GPU would return for division k
    r = np.random.RandomState(100 + k); m = r.randint(2, cap)
    return (float(r.rand() * 1000 - 500), r.randint(0, 5000, size=(m, 2)).astype(np.int32), 0)
res = [fake(k) for k in mine]
nslots = (n + world - 1) // world
slots = torch.from_numpy(sweep.pack_slots(mine, res, cap, nslots))
gathered = [torch.empty_like(slots) for _ in range(world)]
dist.all_gather(gathered, slots)
allres = sweep.unpack_slots(torch.stack(gathered).numpy())
ok = sorted(allres) == list(range(n))
for k, (scr, skl, st) in allres.items():
    s2, k2, st2 = fake(k)
    ok = ok and scr == s2 and np.array_equal(skl, k2) and st == st2
dist.barrier(); dist.destroy_process_group()
print(json.dumps({{"rank": rank, "ok": bool(ok), "mine": len(mine)}}))
'''


def test_shard_partitions_everything():
    order = list(np.argsort(-np.arange(37)))
    for world in (1, 2, 3, 8):
        parts = [sweep.shard(order, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(37))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_pack_unpack_roundtrip():
    skl = np.array([[0, 0], [3, 3], [3, 5], [9, 11]], np.int32)
    slots = sweep.pack_slots([7, 2], [(1.25, skl, 0), (-3.5e300, skl[:2], -6)], cap=8, nslots=3)
    out = sweep.unpack_slots(slots)
    assert set(out) == {7, 2}
    assert out[7][0] == 1.25 and np.array_equal(out[7][1], skl) and out[7][2] == 0
    assert out[2][0] == -3.5e300 and out[2][2] == -6


def test_two_rank_gloo_allgather():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=root)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    import json
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-2000:]
        r = json.loads(out.decode().strip().splitlines()[-1])
        assert r["ok"], r


# ---- a sharded refinement: two gloo ranks run g2g_refine (the product's C++ loop) with the exchange callback on
# torch.distributed; each scores half of every window (the CPU checker in the scorer's seat: no GPU here) and both must hold
# the reference's final MSA at the end.  With FAIL_RANK set, that rank's scorer fails in its third window: BOTH ranks must
# come back with the same error code -- a rank that fails locally still enters the collective (g2g.h).
REFINE_WORKER = r"""
import os, sys, json, hashlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch.distributed as dist
import refinelib
from prrn_aln_amd._lib import G2GError
from prrn_aln_amd.refine import refine_native, torch_exchange
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
f, tree, alp, start = refinelib.load({fixture!r})
fail = int(os.environ.get("FAIL_RANK", "-1")) == rank
res = {{"rank": rank}}
try:
    final, steps, stats = refine_native(None, start, tree, alp, window=8, exchange=torch_exchange(), slot_cap=1024,
                                        scorer=refinelib.oracle_scorer(fail_after=2 if fail else -1))
    refinelib.check_against_trace(f, final, steps, stats)
    res.update(ok=True, digest=hashlib.sha1(final.tobytes()).hexdigest(), scored_here=stats["divisions_scored_here"],
               dps=sum(1 for s in steps if not s["skipped"]))
except G2GError as e:
    res.update(ok=False, error=str(e))
dist.barrier(); dist.destroy_process_group()
print(json.dumps(res))
"""


def _run_ranks(extra_env):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fixture = os.path.join(root, "tests", "golden", "refine_prot12x80_s3.json")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-c", REFINE_WORKER.format(root=root, fixture=fixture)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-2000:]
        outs.append(json.loads(out.decode().strip().splitlines()[-1]))
    return outs


def test_two_rank_sharded_refinement_same_msa_on_every_rank():
    outs = _run_ranks({})
    assert all(o["ok"] for o in outs), outs
    assert outs[0]["digest"] == outs[1]["digest"]
    assert all(0 < o["scored_here"] < o["dps"] for o in outs), outs          # the work really was split


def test_a_rank_local_failure_ends_every_rank_with_the_same_code():
    outs = _run_ranks({"FAIL_RANK": "1"})
    assert not any(o["ok"] for o in outs), outs                               # nobody hangs in the collective, nobody "succeeds"
    codes = [o["error"].split("rc=")[1].split(":")[0] for o in outs]
    assert codes[0] == codes[1] != "0", outs
    assert "rank 1 failed" in outs[0]["error"]                                # rank 0 knows who
