#!/usr/bin/env python3
"""profiles/rNN_xtx_pmc_traffic.json from the PMC passes of tools/xtx_pmc.sh at K = 4096 and K = 14336
(FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, per xtx_kernel launch; the bench's `roofline.traffic`
is the average over the 4 Gram launches of a step: 3 x K=4096 + 1 x K=14336)."""
import csv
import glob
import json
import sys

out = {}
N = 512 * 384
for K, d in ((4096, sys.argv[1]), (14336, sys.argv[2])):
    vals = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "xtx" in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:   # xtx_kernel / xtx16_kernel
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    per = {}
    for k, v in vals.items():
        big = [x for x in v if x > 0.5 * max(v)] or v      # the full launches (the first one is a short warm-up)
        per[k] = sum(big) / len(big)
    fetch = per["FETCH_SIZE"] * 1024 * 2
    write = per["WRITE_SIZE"] * 1024
    alg = N * K * 2 + K * K * 4
    out[str(K)] = {"fetch_corrected": fetch, "write": write, "total": fetch + write, "algorithmic": alg,
                   "ratio": (fetch + write) / alg,
                   "l2_hit_rate": per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])}
doc = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, tools/xtx_pmc.sh) on "
              "`python3 tools/xtx_only.py K 2` (N = 196608 tokens)",
    "unit_note": "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); "
                 "WRITE_SIZE as is; these are L2 fabric-side bytes: Infinity-Cache hits are counted, so this is an upper "
                 "bound of the HBM bytes",
    "per_launch_bytes": out,
    "avg_bytes_per_launch_over_a_step": (3 * out["4096"]["total"] + out["14336"]["total"]) / 4,
}
print(json.dumps(doc, indent=1))
