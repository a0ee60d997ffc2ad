#!/bin/bash
for m in all4 two; do for l in 2 3; do
  QT_BENCH_GROUPING=$m timeout -k 10 400 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --lanes $l 2>/dev/null | grep "^{" > /tmp/g_$m$l.json
  python3 - <<PY
import json
d=json.load(open("/tmp/g_$m$l.json")); r=d["roofline"]
print("$m lanes $l", round(d["ms_per_step"],1), "ms/step", round(d["value"]/1e9,3), "Gw/s live frac", r["frac"])
PY
done; done
