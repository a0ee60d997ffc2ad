#!/bin/bash
# bench.py with 1..4 layers in flight (diagnostic)
for l in 1 2 3 4; do
  timeout -k 10 400 python3 bench.py --steps 12 --warmup 4 --lanes $l --no-cpu-baseline --no-stage-split 2>/dev/null | grep "^{" > /tmp/ln.json
  python3 - "$l" <<'PY'
import json, sys
d = json.load(open("/tmp/ln.json"))
r = d["roofline"]
print("lanes", sys.argv[1], round(d["ms_per_step"], 2), "ms/step", round(d["value"] / 1e9, 3), "Gw/s live frac", r["frac"], "isolated", r["frac_isolated"])
PY
done
