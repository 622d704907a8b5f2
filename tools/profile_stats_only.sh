#!/bin/bash
# kernel trace + stats of bench.py only (no counters), with the engine's per-launch log:  bash tools/profile_stats_only.sh <tag> [steps]
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export G2G_WARN=1
[ -n "$DEBUGLOG" ] && export G2G_DEBUG=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps ${2:-4} --warmup 1 --no-cpu > $O/bench.json 2> $O/bench.err || exit 1
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv
grep -c "timed out" $O/bench.err
grep "timed out" $O/bench.err | head -5
head -8 $O/kernel_stats.csv
