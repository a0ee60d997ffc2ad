#!/usr/bin/env python3
"""Print per-launch durations (us) of the kernels whose name contains any of the given substrings, from a
rocprofv3 --kernel-trace output directory:  kernel_durations.py <dir> <substr> [<substr> ...]"""
import csv
import glob
import sys

d, pats = sys.argv[1], sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(p in n for p in pats):
            print(f"{n[:70]:70s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:10.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X', ''))}")
