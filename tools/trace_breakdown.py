#!/usr/bin/env python3
"""Per-kernel totals of the LAST segment of a rocprofv3 kernel trace, a segment starting at the last launch
whose name contains <marker>:  trace_breakdown.py <dir> <marker>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
seg = rows[idx[-1]:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:62s} {c:5d} launches {t:10.1f} us")
print(f"sum of kernel durations {sum(v[1] for v in agg.values()):.1f} us; first start to last end "
      f"{(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.1f} us")
