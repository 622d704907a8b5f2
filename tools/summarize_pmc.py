"""Sums the rocprofv3 counter CSVs of tools/profile_round.sh per kernel and writes pmc.json / traffic.json."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(O, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("g2g"):
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
fwd = [k for k in per if k not in ("g2g_traceback_kernel", "g2g_spscore_kernel")]
hbm = sum((per[k].get("FETCH_SIZE", 0) + per[k].get("WRITE_SIZE", 0)) * 1024 for k in fwd)
out = {"per_kernel": {k: dict(v) for k, v in per.items()},
       "units": "FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3 derived counters, summed over XCDs; one sweep = bench.py "
                "--steps 1 --warmup 0); SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles",
       "command": "tools/profile_round.sh: rocprofv3 --pmc <set> -- python3 bench.py --steps 1 --warmup 0 --no-cpu, one set per pass",
       "hbm_bytes_forward_kernels": hbm}
json.dump(out, open(os.path.join(O, "pmc.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": hbm,
           "note": "sum of FETCH_SIZE+WRITE_SIZE (KiB*1024) over the forward kernels of ONE sweep (%s); "
                   "algorithmic: 2.986e9 cells * 33 B = 98.5 GB" % " + ".join(sorted(fwd))},
          open(os.path.join(O, "traffic.json"), "w"), indent=1)
print("[profile] hbm bytes per sweep: %.4g" % hbm)
for k in sorted(per):
    v = per[k]
    print(k, {c: v[c] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY") if c in v})
