#!/bin/bash
# bench.py under the Gram-pass stream policies (diagnostic): QT_BENCH_XTX_STREAM x QT_BENCH_XTX_ORDER
for m in group lane:big lane:small shared:big shared:small prio:small; do
  QT_BENCH_XTX_STREAM=${m%%:*} QT_BENCH_XTX_ORDER=${m##*:} timeout -k 10 400 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-stage-split 2>/dev/null | grep "^{" > /tmp/xm.json
  python3 - "$m" <<'PY'
import json, sys
d = json.load(open("/tmp/xm.json"))
r = d["roofline"]
print(sys.argv[1], round(d["ms_per_step"], 1), "ms/step", round(d["value"] / 1e9, 3), "Gw/s live frac", r["frac"], "isolated", r["frac_isolated"])
PY
done
