#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc csv files written by tools/xtx_pmc.sh (per xtx_kernel launch)."""
import csv
import glob
import sys
from collections import defaultdict

out, K = sys.argv[1], int(sys.argv[2])
N = 512 * 384
vals = defaultdict(list)
dur = []
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "xtx" not in row.get("Kernel_Name", "") or "reduce" in row.get("Kernel_Name", ""):
            continue
        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "xtx" in row.get("Kernel_Name", "") and "reduce" not in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
# the first launch of xtx_only.py is a short warm-up on 8192 tokens: keep the long ones
full = [d for d in dur if d > 0.5 * max(dur)] if dur else []
ms = sum(full) / len(full) if full else float("nan")
print(f"# xtx_kernel PMC summary, K = {K}, N = {N} (per full launch; {len(full)} launches, avg {ms:.3f} ms under the profiler)")
print()
print("| counter | per launch |")
print("|---|---|")
per = {}
for name, v in sorted(vals.items()):
    big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
    per[name] = sum(big) / len(big)
    print(f"| {name} | {per[name]:.4g} |")
print()
flops = N * K * (K + 1)
print(f"- {flops / ms / 1e9:.1f} TFLOP/s under the profiler")
if "GRBM_GUI_ACTIVE" in per:
    clk = per["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
    print(f"- effective clock {clk:.3f} GHz (GRBM_GUI_ACTIVE / 8 / time)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        util = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * per["GRBM_GUI_ACTIVE"] / 8)
        print(f"- MFMA utilisation at that clock {100 * util:.1f} % (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles))")
    if "SQ_LDS_IDX_ACTIVE" in per:
        print(f"- LDS array busy {100 * per['SQ_LDS_IDX_ACTIVE'] / (256 * per['GRBM_GUI_ACTIVE'] / 8):.1f} %")
if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per:
    print(f"- SQ_WAIT_ANY / SQ_WAVE_CYCLES = {100 * per['SQ_WAIT_ANY'] / per['SQ_WAVE_CYCLES']:.1f} %")
if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
    print(f"- L2 hit rate {100 * per['TCC_HIT_sum'] / (per['TCC_HIT_sum'] + per['TCC_MISS_sum']):.1f} %")
alg = N * K * 2 + K * K * 4
if "FETCH_SIZE" in per:
    fb = per["FETCH_SIZE"] * 1024 * 2      # KiB; x2 per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)
    wb = per.get("WRITE_SIZE", 0.0) * 1024
    print(f"- fabric traffic: FETCH_SIZE x2 = {fb / 1e9:.2f} GB, WRITE_SIZE = {wb / 1e9:.2f} GB, total {(fb + wb) / 1e9:.2f} GB "
          f"= {(fb + wb) / alg:.1f}x the algorithmic {alg / 1e9:.2f} GB")
