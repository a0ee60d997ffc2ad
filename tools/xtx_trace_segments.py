#!/usr/bin/env python3
"""Split the xtx_kernel launches of a rocprofv3 --kernel-trace of `bench.py` into warm-up / timed /
isolated segments (launch order: 4 per step, then 4 isolated) and print their average durations next
to the figure bench.py measured with HIP events in the same run.

usage: xtx_trace_segments.py <kernel_trace.csv> <bench.json> [warmup_steps] [steps]"""
import csv
import json
import sys

trace, bench = sys.argv[1], sys.argv[2]
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = [r for r in csv.DictReader(open(trace)) if "xtx" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
a, b = 4 * warm, 4 * (warm + steps)
seg = {"warmup": dur[:a], "timed": dur[a:b], "isolated": dur[b:]}
line = json.loads(open(bench).read().strip().splitlines()[-1])
print("| segment | launches | avg ms (rocprofv3 kernel trace) |")
print("|---|---|---|")
for k, v in seg.items():
    if v:
        print(f"| {k} | {len(v)} | {sum(v) / len(v):.3f} |")
r = line["roofline"]
print(f"\nbench.py in the same run (HIP events around the kernel): timed region {r['launches']} launches, "
      f"avg {r['avg_launch_ms']:.3f} ms; ms_per_step {line['ms_per_step']:.2f} (profiler attached)")
