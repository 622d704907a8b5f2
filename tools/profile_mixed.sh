#!/bin/bash
# rocprofv3 kernel trace of tools/mixed_profile.py ON THE GPU BOX:  bash tools/profile_mixed.sh <tag> -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-rXX_mixed}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/tools/mixed_profile.py > $O/mixed.log 2> $O/mixed.err || exit 1
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv
cat $O/mixed.log
