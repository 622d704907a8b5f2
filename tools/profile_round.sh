#!/bin/bash
# Collects the rocprofv3 evidence of one round ON THE GPU BOX (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>        ->  gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json,pmc.json,traffic.json}
# Kernel trace + stats in one run; the counters in their own runs (never together with a trace), one --pmc set per
# pass as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/bench.json 2> $O/bench.err || exit 1
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv
echo "[profile] kernel stats done"
k=0
# QUICK=1: instruction / wait counters only (one pass) -- for iterating on a kernel
if [ -n "$QUICK" ]; then
    timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_3 -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/pmc_3.err || exit 1
    python3 $R/tools/summarize_pmc.py $O
    exit 0
fi
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"; do
    k=$((k + 1))
    timeout -k 10 400 rocprofv3 --pmc $set -d $O/pmc_$k -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/pmc_$k.err || exit 1
    echo "[profile] pmc pass $k done"
done
python3 $R/tools/summarize_pmc.py $O
