#!/bin/bash
# Hunt for the rare scheduler stall of DESIGN.md section 4.2 ON THE GPU BOX (through gpurun, from the repo root):
#   G2G_V6_SMALL_KB=62 HUNT_N=45 bash tools/hunt_stall.sh > gpurun_out/hunt_all.log
# Runs bench.py HUNT_N times (1 warm-up + 6 timed sweeps each, G2G_WARN=1) and prints per run: ms per sweep, the checksum of
# the last sweep's results (config.score_sum / skeleton_corners: the same in every run, event or not) and, if a wait timed
# out, the engine's report with the first time-out's snapshot.  One process at a time, bounded by `timeout`.
mkdir -p gpurun_out
for i in $(seq 1 ${HUNT_N:-40}); do
  G2G_WARN=1 timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu 2> gpurun_out/hunt_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($i, round(d['ms_per_step'],1), repr(d['config']['score_sum']), d['config']['skeleton_corners'], d['config']['failed_items'])"
  grep -A1 "timed out" gpurun_out/hunt_$i.err | cut -c1-900
done
