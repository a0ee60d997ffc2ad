#!/usr/bin/env python3
"""Split the Gram-kernel launches (xtx_kernel / xtx16_kernel) of a rocprofv3 --kernel-trace of `bench.py` into set-up /
warm-up / timed / isolated segments and print their average durations next to the figure bench.py measured with HIP
events in the same run.  The counts come from the bench line itself: `roofline.launches / steps` Gram launches per step,
`layers_in_flight` set-up steps (one per lane, before the warm-up), `warmup` warm-up steps, `steps` timed steps; what
follows are the isolated launches.

usage: xtx_trace_segments.py <kernel_trace.csv> <bench.json>"""
import csv
import json
import sys

trace, bench = sys.argv[1], sys.argv[2]
line = json.loads(open(bench).read().strip().splitlines()[-1])
r = line["roofline"]
steps, warm = int(line["steps"]), int(line["warmup"])
per = int(r["launches"]) // steps
setup = int(line.get("layers_in_flight", 0)) if line.get("config", {}).get("method", "gptq") == "gptq" else 0
rows = [x for x in csv.DictReader(open(trace)) if "xtx" in x["Kernel_Name"] and "reduce" not in x["Kernel_Name"]]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in rows]
a = per * setup
b = a + per * warm
c = b + per * steps
seg = {"set-up (one step per lane)": dur[:a], "warm-up": dur[a:b], "timed": dur[b:c], "isolated": dur[c:]}
print("| segment | launches | avg ms (rocprofv3 kernel trace) |")
print("|---|---|---|")
for k, v in seg.items():
    if v:
        print(f"| {k} | {len(v)} | {sum(v) / len(v):.3f} |")
print(f"\nbench.py in the same run (HIP events around the kernel): timed region {r['launches']} launches, "
      f"avg {r['avg_launch_ms']:.3f} ms; ms_per_step {line['ms_per_step']:.2f} (profiler attached)")
t = seg["timed"]
if t and len(t) == int(r["launches"]):
    print(f"trace vs HIP events over the timed launches: {sum(t) / len(t):.3f} vs {r['avg_launch_ms']:.3f} ms "
          f"({100 * (sum(t) / len(t) / r['avg_launch_ms'] - 1):+.2f} %)")
else:
    print(f"WARNING: {len(t)} launches in the timed segment of the trace, bench.py counted {r['launches']}")
