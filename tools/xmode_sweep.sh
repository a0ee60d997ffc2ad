#!/bin/bash
# bench.py under the Gram-pass stream policies (diagnostic)
for m in group lane prio shared; do
  QT_BENCH_XTX_STREAM=$m timeout -k 10 400 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep "^{" > /tmp/xm_$m.json
  python3 - <<PY
import json
d=json.load(open("/tmp/xm_$m.json"))
r=d["roofline"]
print("$m", round(d["ms_per_step"],1), "ms/step", round(d["value"]/1e9,3), "Gw/s live frac", r["frac"], "isolated", r["frac_isolated"], "avg launch ms", r["avg_launch_ms"])
PY
done
