#!/bin/bash
# bench.py with 2/3/4 layers in flight (diagnostic)
for l in 2 3 4; do
  timeout -k 10 400 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --lanes $l 2>/dev/null | grep "^{" > /tmp/lanes_$l.json
  python3 - <<PY
import json
d=json.load(open("/tmp/lanes_$l.json"))
print("lanes", $l, round(d["ms_per_step"],1), "ms/step", round(d["value"]/1e9,3), "Gw/s live frac", d["roofline"]["frac"], "isolated", d["roofline"]["frac_isolated"])
PY
done
