#!/bin/bash
# usage: tools/pmc_counters.sh <out-subdir under gpurun_out> "<counters>" <kernel-substr> -- <python tool and args>
# one rocprofv3 --pmc pass (with --kernel-trace only); prints the per-launch average of every counter for the kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; CNT=$2; KERN=$3; shift 4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/$1" "${@:2}" > "$OUT/run.log" 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2]
v = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(v.items()):
    print(f"{k:32s} launches {len(x):5d} avg {sum(x) / len(x):14.1f}")
PY
