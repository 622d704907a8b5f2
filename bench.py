#!/usr/bin/env python3
"""bench.py -- DP cells/s of the group-to-group DP hot path on one refinement sweep.

Workload (BASELINE.json configs[2], the one the metric is quoted on): a synthetic family of 256 proteins
x 1024 aa (prrn_aln_amd/synth.py, seed 1; true alignment as the start MSA), one full randiv sweep = the
2N-3 = 509 tree-branch divisions, each re-aligned group-vs-rest by align2()'s DP (Fwd2c::forwardB +
traceback, reference src/fwd2c.h:359,671).  A "step" = one pass over all 509 DPs.  Inputs (profiles, gap
profiles, bands) are built on the host and uploaded BEFORE the timed region; the timed region is kernels +
result fetch (+ the all-gather of result slots for N > 1).  Divisions are dealt round-robin by size to the
ranks; total work is fixed, so scaling is "strong".

    python bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline.achieved` = in-band cells x 33 B (A = 2*Noll*8 + 1)
/ forward-kernel time measured with HIP events on the stream the kernel runs on.  `cpu_baseline` = the
reference's own alignC<recd_t> (oracle/_ref, kind "reference") or, when that is absent, the C restatement
(oracle/, kind "port"), one host core, on a bounded sample of the same divisions."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per concurrent persistent launch (read at HIP initialisation)

BYTES_PER_CELL = {2: 33, 3: 49}          # 2*Noll*sizeof(double) + 1 direction byte
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(sw, budget_s, gpu_scores=None, gpu_sp=None):
    """Reference (or port) CPU path on a bounded sample, 1 core.  Returns dict for the JSON line.  When the GPU scores
    of this rank's divisions are given, the sampled divisions are also compared score by score (the second half of
    the metric: score delta vs the reference)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    order = list(sw.order)
    # spread over the size range: every k-th division of the size-ordered list
    pick = order[:: max(1, len(order) // 24)]
    cells = 0
    secs = 0.0
    used = []
    ref_scr = {}
    ref_sp = {}
    try:
        import refdump
        if not refdump.available():
            raise ImportError("oracle/_ref not built")
        from prrn_aln_amd.sweep import division_groups
        R = refdump.RefLib(molc=sw.alp.molc, ls=sw.alp.ls if sw.alp.ls != 1 else 0)
        inv = {v: k for k, v in __import__("prrn_aln_amd.operator", fromlist=["_AA"])._AA.items() if k.isupper() or k == "-"}
        kind = "reference"
        for k in pick:
            side = sw.branches[k]
            a, b, ia, ib = division_groups(sw.codes, side)
            rows = lambda arr: ["".join(inv.get(int(c), "X") for c in arr[:, j]) for j in range(arr.shape[1])]
            wa = None if sw.weights is None else [float(sw.weights[i]) for i in ia]
            wb = None if sw.weights is None else [float(sw.weights[i]) for i in ib]
            ga = R.group(["a%d" % i for i in ia], rows(a), wa)
            gb = R.group(["b%d" % i for i in ib], rows(b), wb)
            sec, c, mode, scr = R.forward_timed(ga, gb)
            if gpu_sp is not None and int(k) in gpu_sp and len(ref_sp) < 12:
                try:                                   # untimed: the reference's own align2 + calcSpScore for the SP delta
                    ref_sp[int(k)] = R.align_fstat(ga, gb)[1]
                except Exception:
                    pass
            R.free(ga); R.free(gb)
            cells += c; secs += sec; used.append(int(k)); ref_scr[int(k)] = scr
            if secs > budget_s:
                break
    except Exception as e:            # no reference build on this box: time the C restatement instead
        import ctypes as C
        import oraclelib
        from prrn_aln_amd import _abi
        L = oraclelib.load()
        kind = "port"
        cells, secs, used = 0, 0.0, []
        for k in pick:
            q = sw.pwds[k].problem
            res = _abi.Result()
            t = time.perf_counter()
            L.g2g_oracle_forward(C.byref(q), C.byref(res))
            secs += time.perf_counter() - t
            L.g2g_oracle_free(res.trace)
            cells += res.cells; used.append(int(k)); ref_scr[int(k)] = res.score
            if secs > budget_s:
                break
    out = {"value": cells / secs if secs else 0.0, "unit": "cells/s", "cores": 1, "kind": kind,
           "sample": "%d of %d divisions spread over the size range (%.3g cells, %.1f s): forward fill + traceback only"
                     % (len(used), len(sw), cells, secs)}
    if gpu_scores is not None:
        both = [k for k in used if k in gpu_scores]
        out["score_delta_vs_ref"] = {"divisions_compared": len(both),
                                     "max_abs_delta": max([abs(gpu_scores[k] - ref_scr[k]) for k in both] or [0.0])}
    if gpu_sp is not None and ref_sp:
        # sum-of-pairs score of the NEW alignment (Gsinfo.fstat.val): GPU calcSpScore vs the reference's align2 + calcSpScore
        out["sp_score_delta_vs_ref"] = {"divisions_compared": len(ref_sp),
                                        "max_abs_delta": max(abs(gpu_sp[k] - v) for k, v in ref_sp.items())}
    return out


def _allcore_worker(args):
    """One host core's share of the all-core reference figure: the reference's own alignC on its divisions."""
    molc, ls, jobs = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refdump
    R = refdump.RefLib(molc=molc, ls=ls)
    cells, secs = 0, 0.0
    for (na, ra, wa, nb, rb, wb) in jobs:
        ga, gb = R.group(na, ra, wa), R.group(nb, rb, wb)
        sec, c, mode, scr = R.forward_timed(ga, gb)
        R.free(ga); R.free(gb)
        cells += c; secs += sec
    return cells, secs


def cpu_baseline_allcores(sw, budget_s):
    """The reference's alignC<recd_t> on every host core at once (one process per core: the reference keeps its parameters in
    process globals), divisions spread over the size range; throughput = all cells / the slowest core's DP time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp
    import refdump
    if not refdump.available():
        return None
    from prrn_aln_amd.sweep import division_groups
    # the GPU box gives one GPU's job a share of 16 host cores (its cpu_count() is the whole machine's): never more than that
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    order = list(sw.order)
    per_core = max(1, int(budget_s * 1.6e7 / max(1.0, float(sw.cells.mean()))))        # ~1.6e7 cells/s per core
    pick = order[:: max(1, len(order) // (cores * per_core))][: cores * per_core]
    inv = {v: k for k, v in __import__("prrn_aln_amd.operator", fromlist=["_AA"])._AA.items() if k.isupper() or k == "-"}
    rows = lambda arr: ["".join(inv.get(int(c), "X") for c in arr[:, j]) for j in range(arr.shape[1])]
    jobs = [[] for _ in range(cores)]
    for i, k in enumerate(pick):
        a, b, ia, ib = division_groups(sw.codes, sw.branches[k])
        wa = None if sw.weights is None else [float(sw.weights[j]) for j in ia]
        wb = None if sw.weights is None else [float(sw.weights[j]) for j in ib]
        jobs[i % cores].append((["a%d" % j for j in ia], rows(a), wa, ["b%d" % j for j in ib], rows(b), wb))
    ls = sw.alp.ls if sw.alp.ls != 1 else 0
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_allcore_worker, [(sw.alp.molc, ls, j) for j in jobs])
    cells = sum(c for c, _ in res)
    slow = max(s for _, s in res)
    return {"value": cells / slow if slow else 0.0, "unit": "cells/s", "cores": cores, "kind": "reference",
            "sample": "%d of %d divisions spread over the size range on %d processes (%.3g cells, slowest core %.1f s of DP time): "
                      "forward fill + traceback only" % (len(pick), len(sw), cores, cells, slow)}


def reference_prefix(rows, names, seconds, extra_opts=()):
    """The reference's own refinement (oracle/_ref/prrn5_trace = the unmodified prrn5.cc + a tracing wrapper of align2) started on
    the same MSA on one host core of THIS box and stopped after `seconds`: how many align2() calls it got through.  (The whole run
    takes the better part of an hour for the 256 x 1024 family; its duration in the build container is in the fixture.)"""
    import signal
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refdump
    ref = os.path.join(ROOT, "oracle", "_ref")
    exe = os.path.join(ref, "prrn5_trace")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        refdump.write_multi(os.path.join(tmp, "fam.msa"), names, rows, "fam")
        tr = os.path.join(tmp, "trace.txt")
        env = dict(os.environ, ALN_TAB=os.path.join(ref, "table"), G2G_TRACE=tr)
        t0 = time.perf_counter()
        p = subprocess.Popen([exe, "-YH0", "-R1"] + list(extra_opts) + ["fam.msa"], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            p.wait(timeout=seconds)
        except subprocess.TimeoutExpired:
            p.send_signal(signal.SIGKILL)
            p.wait()
        dt = time.perf_counter() - t0
        calls = 0
        if os.path.exists(tr):
            with open(tr) as fd:
                calls = sum(1 for l in fd if l.startswith("A "))
    return {"align2_calls": calls, "seconds": dt, "finished": p.returncode == 0}


def refinement_readout(ctx, args):
    """The second half of the metric on the named workload itself: Prrn::rir (reference src/prrn5.cc:633-666) of the 256 x 1024 aa
    family from the reference's progressive MSA, through g2g_refine, against the committed trace of the reference's own run
    (tests/golden/refine_prot256x1024_prog.json.gz, tools/make_refine_golden.py): same branch sequence, same DP score and
    fstat.val at every align2(), same final MSA => the SP-score delta of the refined alignment is exactly 0; pairsum_ss of
    both final MSAs is computed anyway.  The reference itself is timed beside it on one host core for a bounded prefix."""
    import gzip
    import numpy as np
    from prrn_aln_amd import operator as op
    from prrn_aln_amd.refine import KTree, refine_native, pairsum
    big = os.path.join(ROOT, "tests", "golden", "refine_prot256x1024_prog.json.gz")
    small = os.path.join(ROOT, "tests", "golden", "refine_prot48x300_s5.json")
    use_big = os.path.exists(big) and not args.dna and not os.environ.get("G2G_BENCH_SMALL_REFINE")
    f = json.load(gzip.open(big, "rt")) if use_big else json.load(open(small))
    t = f["tree"]
    tree = KTree(t["left"], t["right"], t["parent"], t["vol"], t["cur"])
    ralp = op.AlnParam(ls=f["ls"], molc=f["molc"], max_code=25 if f["molc"] == 1 else 17)
    start = op.encode(f["rows"], f["molc"])
    refine_native(ctx, start, tree, ralp, seed=1, maxitr=1, window=2) if not use_big else None      # (warm: first use of the small-batch paths)
    c0 = ctx.counters()
    t1 = time.perf_counter()
    # (g2g_refine starts every window of the big family at 16 speculative divisions, of the small one at 4: g2g_refine.cpp)
    final, rsteps, rstats = refine_native(ctx, start, tree, ralp, seed=1, maxitr=10, window=16)
    rt = time.perf_counter() - t1
    c1 = ctx.counters()
    want = op.encode(f["final_rows"], f["molc"])
    same = bool(np.array_equal(final, want))
    dps = [x for x in rsteps if not x["skipped"]]
    scr_same = len(dps) == len(f["align2"]) and all(x["scr"] == r["scr"] and x["val_new"] == r["val"] for x, r in zip(dps, f["align2"]))
    sp_gpu = pairsum(ctx, final, tree, ralp)
    # the REFERENCE's own Ssrel::pairsum_ss values of this run (fixture key pairsum_ss, tools/make_refine_pairsum.py: traced out of
    # `prrn5 -YH0 -R1 -O4` itself): start and refined MSA on the refinement's tree, and the -O4 read-outs on the tree it rebuilt
    ref_ps = {(c["msa"], c["use_pw"], c["tree"] is None): c for c in f.get("pairsum_ss", [])}
    sp_ref = ref_ps[("final", 1, True)]["value"] if ("final", 1, True) in ref_ps else None
    o4 = []
    for c in f.get("pairsum_ss", []):
        if c["tree"] is None:
            continue
        t2 = c["tree"]
        ours = pairsum(ctx, final, KTree(t2["left"], t2["right"], t2["parent"], t2["vol"], t2["cur"]), ralp, use_pw=bool(c["use_pw"]))
        o4.append({"use_pw": c["use_pw"], "ours_on_our_refined_msa": ours, "reference": c["value"], "abs_delta": abs(ours - c["value"])})
    out = {"family": "%d %s x %s, start MSA %d columns (%s: trace of the reference's own Prrn::rir, serial, -YH0 -R1)"
                     % (len(f["rows"]), "proteins" if f["molc"] == 1 else "DNA sequences", "1024 aa" if use_big else "300 aa", len(f["rows"][0]), os.path.basename(big if use_big else small)),
           "engine": "g2g_refine (C++ behind the C ABI): windows of speculative divisions batched on the GPU, in-order acceptance",
           "wall_s": rt, "divisions_evaluated": rstats["divisions"], "align2_calls": len(dps), "accepted_moves": rstats["accepted"],
           "gpu_batches": rstats["batches"], "divisions_recomputed": rstats["divisions_wasted"],
           "wait_timeouts": int(c1["wait_timeouts"] - c0["wait_timeouts"]), "recovered_dps": int(c1["recovered_dps"] - c0["recovered_dps"]),
           "last_timeout_report": (ctx.last_timeout()[:4000] if c1["recovered_dps"] > c0["recovered_dps"] else None),
           "wait_gaps": list(ctx.wait_gaps()),
           "same_branch_sequence_as_reference": [x["branch"] for x in rsteps] == f["branches"],
           "every_dp_score_and_fstat_val_equal_to_reference": bool(scr_same),
           "final_msa_identical_to_reference": same,
           # Ssrel::pairsum_ss (g2g_pairsum, on the trace's tree) of the start MSA, of OUR refined MSA and of the REFERENCE's refined MSA
           "pairsum_ss_start": pairsum(ctx, start, tree, ralp), "pairsum_ss_refined": sp_gpu,
           # what the reference itself printed / computed for ITS refined MSA (not our kernel on its MSA): Ssrel::pairsum_ss, src/fspscore.cc:896
           "pairsum_ss_reference_start": ref_ps[("start", 1, True)]["value"] if ("start", 1, True) in ref_ps else None,
           "pairsum_ss_reference_refined": sp_ref,
           "sp_delta_vs_reference": (abs(sp_gpu - sp_ref) if sp_ref is not None else None),
           "sp_readout_O4_on_the_reference_s_rebuilt_tree": o4}
    ref = {"whole_run_s_one_core_build_container": f.get("reference_seconds")}
    if not args.no_cpu:
        names = [("s%03d" if len(f["rows"]) > 99 else "s%02d") % i for i in range(len(f["rows"]))]
        live = reference_prefix(list(f["rows"]), names, 30.0 if use_big else 20.0, ["-yl3"] if f["ls"] == 3 else [])
        if live:
            n = live["align2_calls"]
            ours = dps[n - 1]["t_ms"] / 1e3 if 0 < n <= len(dps) else None
            ref.update({"live_prefix_on_this_box": live, "g2g_refine_s_for_the_same_prefix": ours,
                        "speedup_on_the_prefix": (live["seconds"] / ours) if ours else None})
    out["reference_prrn5"] = ref
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=256)
    ap.add_argument("--length", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--limit", type=int, default=0, help="use only the first LIMIT divisions (debug)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dna", action="store_true", help="non-default workload: DNA family, long-gap penalty (ls=3, Noll 3 kernels)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="rehearsal on ONE GPU: run only rank 0's share of an N-rank job (prints a line marked rehearsal)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # G2G_BENCH_REHEARSE=1: all ranks on GPU 0 with the gloo backend -- a rehearsal of the N>1 code path on a one-GPU box
    # (RCCL refuses two ranks on one device); the number it prints is not a scaling result and says so.
    rehearse = world > 1 and os.environ.get("G2G_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)

    from prrn_aln_amd import engine, operator as op, sweep
    from prrn_aln_amd.synth import make_family

    if args.dna:
        from prrn_aln_amd.synth import DNA
        fam = make_family(args.nseq, args.length, args.seed, alphabet=DNA)
        alp = op.AlnParam(ls=3, molc=op.DNA, max_code=17)
    else:
        fam = make_family(args.nseq, args.length, args.seed)
        alp = op.AlnParam()
    ctx = engine.Context(local_rank)
    # the divisions' inputs: groups split out of the MSA on host threads, then thickness / vectors / gap profiles of all 1018 groups
    # built on the DEVICE in one batch (g2g_pwdm_create_batch, csrc/g2g_build.hip: SURVEY section 8 rows a8 / a9).  Outside the timed
    # region; reported as config.builders (with the host builders timed beside them at N = 1).
    sw = sweep.Sweep(fam, alp, weighted=True, limit=args.limit or None, ctx=ctx)
    builders = {"groups_split_ms": 1e3 * sw.t_split, "device_batch_ms": 1e3 * sw.t_batch, "groups": 2 * len(sw)}
    if world == 1 and not args.shard_of and not args.no_cpu:
        t1 = time.perf_counter()
        sw_host = sweep.Sweep(fam, alp, weighted=True, limit=args.limit or None)
        builders["host_threads_ms_incl_split"] = 1e3 * (time.perf_counter() - t1)
        builders["host_threads"] = min(16, os.cpu_count() or 1)
        del sw_host
    mine = sweep.shard(sw.order, args.shard_of, 0) if args.shard_of > 1 and world == 1 else sweep.shard(sw.order, world, rank)
    holders = [sw.pwds[k] for k in mine]

    # inputs resident in HBM before the timed region
    class _H:                                   # adapter: engine.Context wants objects with a `.c` Problem
        def __init__(self, q): self.c = q
    hs = [_H(p.problem) for p in holders]
    try:
        batch = ctx.prepare(hs)
    except Exception as e:
        # a sweep whose state does not fit one GPU's HBM (e.g. BASELINE configs[4]-like DNA families) cannot be resident:
        # it is streamed in chunks by g2g_forward_batch (pack + upload + kernels + fetch per chunk, all inside the clock)
        if "out of memory" not in str(e):
            raise
        batch = None
    cap = 4096                                  # corners per result slot (longer skeletons flag -99)
    nslots = (len(sw) + world - 1) // world

    def step():
        if batch is None:
            res = ctx.forward_batch(hs)
        else:
            batch.run()
            res = batch.fetch()
        out = [(scr, engine.stdskl(tr), st) for (scr, cells, tr, st) in res]
        if world > 1:
            slots = torch.from_numpy(sweep.pack_slots(mine, out, cap, nslots))
            if not rehearse:
                slots = slots.cuda()
            gathered = [torch.empty_like(slots) for _ in range(world)]
            dist.all_gather(gathered, slots)
            # the consumer of the exchange, inside the timed step: every rank unpacks ALL ranks' slots into the complete table of the
            # sweep's results -- what Prrn::best_of_n looks at (reference src/prrn5.cc:594-631) -- and takes the same decision on it
            # (here: the division with the best DP score; ties to the lower id).  No rank acts on numbers another rank does not hold.
            table = sweep.unpack_slots(torch.stack(gathered).cpu().numpy())
            if len(table) != len(sw) or any(v[2] != 0 for v in table.values()):
                raise RuntimeError("exchange: %d of %d divisions arrived, %d failed" % (len(table), len(sw), sum(1 for v in table.values() if v[2] != 0)))
            best = min(table, key=lambda k: (-table[k][0], k))
            return out, {"divisions_on_every_rank": len(table), "best_division": int(best), "best_score": float(table[best][0]),
                         "bytes_gathered_per_rank": int(slots.numel() * slots.element_size() * world)}
        return out, None

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fwd_ms = tb_ms = 0.0
    wait_timeouts, recovered_dps, gaps_step = [], [], []  # per timed step: scheduler waits that ran into their limit / DPs re-run for it /
    cnt_before = ctx.counters()                           # waiting waves that found themselves off the machine for > 4 ms
    gaps_before = ctx.wait_gaps()[0]
    for _ in range(args.steps):
        out, gathered = step()
        f, t = batch.times_ms() if batch is not None else (0.0, 0.0)
        fwd_ms += f; tb_ms += t
        cnt_now = ctx.counters()
        wait_timeouts.append(int(cnt_now["wait_timeouts"] - cnt_before["wait_timeouts"]))
        recovered_dps.append(int(cnt_now["recovered_dps"] - cnt_before["recovered_dps"]))
        cnt_before = cnt_now
        g_now = ctx.wait_gaps()[0]
        gaps_step.append(int(g_now - gaps_before))
        gaps_before = g_now
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # f1 (not part of the metric): PreSpScore::calcSpScore of every new alignment on the resident batch
    sp_ms = None
    try:
        if batch is None:
            raise RuntimeError("sweep not resident")
        sps = [op.spparams(p) for p in holders]
        t1 = time.perf_counter()
        fst = batch.spscore(sps, [skl for (_, skl, _) in out])
        sp_ms = 1e3 * (time.perf_counter() - t1)
        sp_bad = sum(1 for (_, _, st) in fst if st != 0)
        gpu_sp = {int(k): float(v) for k, (v, _, st) in zip(mine, [x[:3] for x in fst]) if st == 0}
    except Exception as e:                                   # never let the extra row break the benchmark line
        sp_ms, sp_bad = None, str(e)
        gpu_sp = None
    # (not part of the metric either) the whole operator from host memory: g2g_align2_batch = pack + upload + kernels +
    # fetch + stdskl + end check, i.e. what a caller pays per sweep when nothing is resident
    e2e_ms = None
    if world == 1 and not args.shard_of and not args.no_cpu:     # (--no-cpu = the timed region only: profiling runs)
        try:
            e2e_ms = float("inf")
            for _ in range(2):                             # (the first call also allocates its device arena: steady state = min)
                t1 = time.perf_counter()
                full = op.align2_batch(ctx, holders)
                e2e_ms = min(e2e_ms, 1e3 * (time.perf_counter() - t1))
            if any(st != 0 for (_, _, st) in full) or any(a[0] != b_[0] for a, b_ in zip(full, out)):
                e2e_ms = -e2e_ms                           # scores differ from the resident-batch run: flag it
        except Exception:
            e2e_ms = None
    # second workload (not the metric): the same family, sweep started from the REFERENCE's progressive MSA (fixture made by
    # tools/make_progressive_fixture.py) -- shorter, rougher, smaller DPs than the synthetic true alignment
    prog = None
    ppath = os.path.join(ROOT, "tests", "golden", "msa", "prog256x1024.npz")
    if world == 1 and not args.shard_of and not args.dna and not args.limit and args.nseq == 256 and args.length == 1024 and args.seed == 1 \
            and os.path.exists(ppath) and not args.no_cpu:
        try:
            sw2 = sweep.Sweep(fam, alp, weighted=True, codes=np.load(ppath)["codes"])
            b2 = ctx.prepare([_H(p.problem) for p in sw2.pwds])
            b2.run(); b2.fetch()
            t1 = time.perf_counter()
            for _ in range(3):
                b2.run()
                r2 = b2.fetch()
            dt2 = (time.perf_counter() - t1) / 3
            prog = {"start_msa": "reference progressive alignment (prrn5 -YH0 -S0), %d columns" % sw2.codes.shape[0],
                    "divisions": len(sw2), "cells_per_step": int(sw2.cells.sum()), "ms_per_step": 1e3 * dt2,
                    "cells_per_s": float(sw2.cells.sum()) / dt2, "failed_items": sum(1 for r in r2 if r[3] != 0)}
            del b2
        except Exception as e:
            prog = {"error": str(e)[:200]}
    # two more read-outs of the same context (not the metric, outside the timed region):
    #  - the whole refinement loop (f2: Randiv order, windows of speculative divisions batched on the GPU, in-order acceptance)
    #    on a family the reference's own trajectory is committed for: wall time, and whether the final MSA is the reference's
    #    byte for byte (then the SP-score delta of the refined alignment vs the reference is exactly 0)
    #  - the guide-tree stage (f3: alnScoreD for all pairs of the bench family, one launch)
    refinement = guide_tree = None
    if world == 1 and not args.shard_of and not args.limit and not args.no_cpu:
        try:
            refinement = refinement_readout(ctx, args)
        except Exception as e:
            refinement = {"error": str(e)[:300]}
        try:
            from prrn_aln_amd import guide
            if not args.dna:
                prm, _mtx_keep = alp.to_c()
                gseqs = [op.encode([row.replace("-", "")], alp.molc)[:, 0].copy() for row in fam.msa]
                gia, gib = guide.all_pairs(len(gseqs))
                guide.alnscored_batch(ctx, prm, gseqs, gia[:8], gib[:8])
                t1 = time.perf_counter()
                gsc, gst = guide.alnscored_batch(ctx, prm, gseqs, gia, gib)
                gt = time.perf_counter() - t1
                glen = np.array([len(x) for x in gseqs], np.int64)
                guide_tree = {"what": "alnScoreD (Fwd2d::forwardD) for all pairs of the bench family, one call from host buffers",
                              "pairs": int(len(gia)), "wall_ms": 1e3 * gt, "full_matrix_cells": int((glen[gia] * glen[gib]).sum()),
                              "failed": int((gst != 0).sum())}
        except Exception as e:
            guide_tree = {"error": str(e)[:200]}
    bad = [mine[i] for i, (scr, skl, st) in enumerate(out) if st != 0 or len(skl) < 2]
    my_cells = int(sum(sw.cells[k] for k in mine))
    total_cells = int(sw.cells.sum())
    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = total_cells * args.steps / dt
        fwd_avg_ms = fwd_ms / args.steps
        noll = holders[0].problem.noll if holders else 2
        line_tag = "NON-DEFAULT workload (DNA, ls=3): not the BASELINE metric" if args.dna else None
        if batch is None:
            line_tag = (line_tag + "; " if line_tag else "") + "sweep does not fit HBM: streamed in chunks, upload and packing inside the timed region"
        if rehearse:
            line_tag = "REHEARSAL: %d ranks share GPU 0 over gloo; exercises the N>1 code path, not a scaling result" % world
        ach = my_cells * BYTES_PER_CELL[noll] / (fwd_avg_ms * 1e-3) / 1e9 if fwd_avg_ms else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                # measured on the whole default sweep; a rank / a non-default workload launches a different number of
                # cells: scaled by cells (the PMC traffic is proportional to cells to within a few per cent), null for
                # the workload the counters were not taken on
                if traffic is not None:
                    traffic = None if (args.dna or args.nseq != 256 or args.length != 1024 or args.limit) else traffic * my_cells / total_cells
            except Exception:
                traffic = None
        line = {
            "metric": "DP cells/s (profile-profile fwd2) + SP-score delta vs ref, 256x1024aa refinement",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "prrn refinement sweep: %d proteins x %d aa (seed %d), start MSA %d columns, "
                                   "%d tree-branch divisions (HLF/RHF %d, GPF %d), %.4g in-band cells/sweep, band sh=-60"
                                   % (args.nseq, args.length, args.seed, len(fam.msa[0]), len(sw),
                                      sum(1 for p in sw.pwds if p.alnmode in (7, 8)),
                                      sum(1 for p in sw.pwds if p.alnmode == 9), total_cells),
                       "divisions": len(sw), "cells_per_step": total_cells, "parallelism": "divisions round-robin by size over %d GPU(s)" % world,
                       "exchange": gathered,
                       "failed_items": bad, "align2_batch_from_host_ms": e2e_ms, "builders": builders,
                       # every run is evidence about the scheduler's waits: per timed step, the waits that ran into
                       # their wall-clock limit and the DPs re-run because of it (rank 0's share); an ordinary run shows zeros
                       "wait_timeouts": wait_timeouts, "recovered_dps": recovered_dps,
                       # ... and the stretches of more than 4 ms a waiting wave spent off the machine since the context was created
                       # (count, longest in ms: g2g_ctx_wait_gaps)
                       "wait_gaps": list(ctx.wait_gaps()), "wait_gaps_per_timed_step": gaps_step,
                       # checksum of the last timed step's results (this rank's divisions): the same workload must give the same
                       # two numbers in every run, with or without a recovered time-out in it
                       "score_sum": float(sum(scr for (scr, _, st) in out if st == 0)),
                       "skeleton_corners": int(sum(len(skl) for (_, skl, st) in out if st == 0)),
                       "arena_bytes_per_cell": (batch.arena_bytes() / my_cells) if (batch is not None and my_cells) else None},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "g2g_v3r_hf2 + g2g_v6_pf2 (persistent strip kernels incl. boundary chains and their own column scores, concurrent)", "kernel_ms": fwd_avg_ms, "traceback_ms": tb_ms / args.steps,
                         "bytes_per_cell": BYTES_PER_CELL[noll], "cells_per_launch": my_cells},
        }
        if prog is not None:
            line["config"]["progressive_start"] = prog
        if refinement is not None:
            line["config"]["refinement"] = refinement
        if guide_tree is not None:
            line["config"]["guide_tree"] = guide_tree
        if sp_ms is not None:
            line["config"]["calcSpScore_ms"] = sp_ms
            line["config"]["calcSpScore_failed"] = sp_bad
        if line_tag:
            line["config"]["note"] = line_tag
            if args.dna:
                line["config"]["workload"] = line["config"]["workload"].replace("proteins", "DNA sequences").replace(" aa ", " nt ")
        if args.shard_of > 1 and world == 1:
            line["rehearsal"] = "rank 0's share of a %d-rank job on one GPU: %d divisions, %.4g cells, %.1f ms per step" % (
                args.shard_of, len(mine), my_cells, ms_per_step)
        if not args.no_cpu and world == 1:                # (rank 0 at N = 1 only)
            one = cpu_baseline(sw, args.cpu_seconds, {int(k): float(o[0]) for k, o in zip(mine, out)}, gpu_sp)
            allc = None
            try:
                allc = cpu_baseline_allcores(sw, args.cpu_seconds) if not args.dna else None
            except Exception as e:                           # (never let the extra figure break the line)
                allc = None
            if allc and allc["value"]:
                # the contract's cpu_baseline = the reference on ALL host cores; the one-core figure and the parity deltas ride along
                allc["one_core"] = {k: one[k] for k in ("value", "sample", "kind")}
                for k in ("score_delta_vs_ref", "sp_score_delta_vs_ref"):
                    if k in one:
                        allc[k] = one[k]
                line["cpu_baseline"] = allc
                line["config"]["gpu_over_cpu_allcores"] = value / allc["value"]
            else:
                line["cpu_baseline"] = one
            if one["value"]:
                line["config"]["gpu_over_cpu_1core"] = value / one["value"]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
