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
# the forward kernels of a sweep: the strip kernels (and the v1 kernel); not the traceback, the calcSpScore walks, the device
# builders (they run once, before the timed region) or the clock probe
fwd = [k for k in per if k.startswith(("g2g_v2_", "g2g_v3", "g2g_v6_", "g2g_v7_", "g2g_v8_", "g2g_forward_kernel"))]
hbm = sum((per[k].get("FETCH_SIZE", 0) + per[k].get("WRITE_SIZE", 0)) * 1024 for k in fwd)
out = {"per_kernel": {k: dict(v) for k, v in per.items()},
       "units": "FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3 derived counters, summed over XCDs; one sweep = bench.py "
                "--steps 1 --warmup 0); SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles",
       "command": "tools/profile_round.sh: rocprofv3 --pmc <set> -- python3 bench.py --steps 1 --warmup 0 --no-cpu, one set per pass",
       "hbm_bytes_forward_kernels": hbm}
json.dump(out, open(os.path.join(O, "pmc.json"), "w"), indent=1)
fetch = sum(per[k].get("FETCH_SIZE", 0) * 1024 for k in fwd)
json.dump({"hbm_bytes_per_launch": hbm,
           "fetch_bytes": fetch, "write_bytes": hbm - fetch,
           "note": "sum of FETCH_SIZE+WRITE_SIZE (KiB*1024, raw: separate --pmc passes) over the forward kernels of ONE sweep (%s); "
                   "algorithmic: 2.986e9 cells * 33 B = 98.5 GB.  /opt/skills/guides/MI355X_MICROARCH.md calibrates FETCH_SIZE only for "
                   "16-B-per-lane streaming reads (x 2 there); these kernels read bytes, dwords and 8-byte words: the raw sum is "
                   "reported, and with every fetched byte doubled the upper bound is hbm_bytes_if_fetch_doubled" % " + ".join(sorted(fwd)),
           "hbm_bytes_if_fetch_doubled": hbm + fetch},
          open(os.path.join(O, "traffic.json"), "w"), indent=1)
print("[profile] hbm bytes per sweep: %.4g" % hbm)
for k in sorted(per):
    v = per[k]
    print(k, {c: v[c] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY") if c in v})
