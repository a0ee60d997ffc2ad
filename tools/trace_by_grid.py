#!/usr/bin/env python3
"""Totals per (kernel, grid size) of a rocprofv3 kernel trace:  trace_by_grid.py <dir> <kernel-name substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        k = (r.get("Grid_Size_X", "") or r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:25]:
    print(f"grid {k}: {c:5d} launches {t:10.1f} us total {t / c:8.1f} us avg")
