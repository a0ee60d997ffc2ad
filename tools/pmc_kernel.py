#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counters over the launches of one kernel (name substring) that last >= min_us:
    pmc_kernel.py <dir with counter_collection.csv + kernel_trace.csv> <kernel substring> [min_us]
Prints summed counters and the usual quotients (clock, MFMA-busy, LDS busy, wait share, L2 hit rate, fabric bytes)."""
import csv
import glob
import sys
from collections import defaultdict

d, pat = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(f.rsplit("/", 2)[-3] if "/" in f else "", r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = defaultdict(float)
t_us, n = defaultdict(float), defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    key0 = f.rsplit("/", 2)[-3] if "/" in f else ""
    seen = set()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = (key0, r["Dispatch_Id"])
        if k not in dur or dur[k] < min_us:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Counter_Name"]) not in seen and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
            pass
        if k not in seen:
            seen.add(k)
            t_us[key0] += dur[k]
            n[key0] += 1
print(f"{pat}: launches >= {min_us} us per pass: {dict(n)}; time per pass (us): { {k: round(v, 1) for k, v in t_us.items()} }")
for k in sorted(tot):
    print(f"  {k:28s} {tot[k]:.4g}")
g = tot.get("GRBM_GUI_ACTIVE")
if g:
    pass_t = next((t_us[k] for k in t_us if t_us[k] > 0), 0.0)
    cyc = g / 8
    print(f"  clock {cyc / pass_t / 1e3:.3f} GHz" + (f", MFMA busy {100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} %" if "SQ_VALU_MFMA_BUSY_CYCLES" in tot else "")
          + (f", LDS busy {100 * tot['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.1f} %" if "SQ_LDS_IDX_ACTIVE" in tot else "")
          + (f", LDS bank conflict cycles / LDS active {100 * tot['SQ_LDS_BANK_CONFLICT'] / max(1.0, tot.get('SQ_LDS_IDX_ACTIVE', 1.0)):.1f} %" if "SQ_LDS_BANK_CONFLICT" in tot else ""))
if "SQ_WAIT_ANY" in tot and "SQ_WAVE_CYCLES" in tot:
    print(f"  SQ_WAIT_ANY / SQ_WAVE_CYCLES {100 * tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']:.1f} %, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {100 * tot.get('SQ_WAIT_INST_ANY', 0) / tot['SQ_WAVE_CYCLES']:.1f} %")
if "TCC_HIT_sum" in tot:
    print(f"  L2 hit rate {100 * tot['TCC_HIT_sum'] / (tot['TCC_HIT_sum'] + tot['TCC_MISS_sum']):.1f} %")
if "FETCH_SIZE" in tot:
    print(f"  fabric read {tot['FETCH_SIZE'] * 1024 * 2 / 1e9:.2f} GB (FETCH_SIZE x 2), write {tot.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB")
