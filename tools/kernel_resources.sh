#!/bin/bash
# Compiler resource lines (registers, spills, scratch, occupancy) of every kernel of libg2g.so -> profiles/<tag>_kernel_resources.txt.
# Runs without a GPU: hipcc cross-compiles gfx950; the flags are prrn_aln_amd/build.py's.
#   bash tools/kernel_resources.sh r03
set -o pipefail
TAG=${1:-rXX}
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -mllvm -amdgpu-promote-alloca-to-vector-limit=4096 -DG2G_FWD_THREADS=512 -DG2G_V2_THREADS=256 -DG2G_V2_MINWAVES=3 -DG2G_V2_TILE_COLS=512 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result"
T=$(mktemp -d)
for u in g2g_engine g2g_tu_v2 g2g_tu_v3 g2g_tu_v6 g2g_tu_v78; do
    /opt/rocm/bin/hipcc $F --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o $T/$u.o $R/prrn_aln_amd/csrc/$u.hip 2>&1 |
        grep -E "Function Name|SGPRs:|VGPRs:|AGPRs|ScratchSize|Occupancy|Spill|LDS Size" | sed 's/.*remark: //' | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - - -
done > $T/all.txt
python3 - "$T/all.txt" "$R/profiles/${TAG}_kernel_resources.txt" <<'PY'
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", l)
    if not m:
        continue
    g = lambda k: int(re.search(k + r": (\d+)", l).group(1))
    rows.append((m.group(1), g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill")))
out = ["# Compiler resource lines of every kernel of libg2g.so (hipcc 7.2, -Rpass-analysis=kernel-resource-usage, the flags of prrn_aln_amd/build.py;",
       "# runs in the GPU-less container: tools/kernel_resources.sh).  SGPR spills live in VGPR lanes (v_writelane / v_readlane), not in memory.",
       "%-28s %5s %5s %5s %8s %10s %7s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "waves/SIMD", "sspill", "vspill")]
out += ["%-28s %5d %5d %5d %8d %10d %7d %7d" % r for r in rows]
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $T
