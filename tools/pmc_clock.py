#!/usr/bin/env python3
"""Clock and MFMA-pipe utilisation per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
--kernel-trace run:  pmc_clock.py <dir> <kernel-name substring> [min_us]
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles).
GRBM_GUI_ACTIVE also counts the cycles around a dispatch (command processor, cache flushes), so for short launches the
quotient is not a clock (round 3 printed 2.9-3.1 GHz for 27-34 us kernels on a 2.4 GHz part): below 200 us average the
clock and the MFMA-busy figure derived from it are withheld."""
import csv
import glob
import sys
from collections import defaultdict

d, pat = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cnt = defaultdict(dict)
name = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot_t = tot_c = tot_m = 0.0
n = 0
for k, c in cnt.items():
    if pat not in name[k] or k not in dur or dur[k] < min_us or "GRBM_GUI_ACTIVE" not in c:
        continue
    tot_t += dur[k]
    tot_c += c["GRBM_GUI_ACTIVE"] / 8
    tot_m += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    n += 1
MIN_CLOCK_US = 200.0
if n and tot_t / n < MIN_CLOCK_US:
    print(f"{pat}: {n} launches >= {min_us} us, {tot_t / n:.1f} us avg -- too short for a GRBM-derived clock "
          f"(< {MIN_CLOCK_US:.0f} us): clock and MFMA-busy withheld; MFMA-busy cycles per launch {tot_m / n:.3e}")
elif n:
    print(f"{pat}: {n} launches >= {min_us} us, {tot_t / n:.1f} us avg, clock {tot_c / tot_t / 1e3:.3f} GHz, "
          f"MFMA busy {100 * tot_m / (1024 * tot_c):.1f} %")
else:
    print(f"{pat}: no launches")
