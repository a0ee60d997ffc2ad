#!/bin/bash
# Lab builds of the library with -DQT_SWEEP_LAB=n (timing-only ablations of sweep_quad_kernel; results are wrong):
#   1 = prologue + epilogue only, 2 = no stores inside the loop, 3 = no scale / zero-point loads inside the loop,
#   4 = a step's LDS reads at its own start instead of one step ahead (correct results: the round-3/4 schedule)
# usage (in the build container): tools/sweep_lab.sh build      -> quantool_amd/lib/lab/libquantool_hip_lab{1,2,3,4}.so
# usage (on the GPU box):         tools/sweep_lab.sh run <out-subdir under gpurun_out> K R [variants, default "0 1 2 3 4"]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -I$R/include"
if [ "$1" = build ]; then
    mkdir -p "$R/quantool_amd/lib/lab"
    for n in 1 2 3 4; do
        /opt/rocm/bin/hipcc $FLAGS -DQT_SWEEP_LAB=$n -c "$R/quantool_amd/csrc/sweep.hip" -o /tmp/sweep_lab$n.o || exit 1
        objs=$(ls "$R"/quantool_amd/csrc/_obj/*.o | grep -v '/sweep.o')
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/quantool_amd/lib/lab/libquantool_hip_lab$n.so" $objs /tmp/sweep_lab$n.o || exit 1
    done
    ls -la "$R/quantool_amd/lib/lab"
    exit 0
fi
OUT=$R/gpurun_out/$2; K=$3; ROWS=$4; VARS=${5:-0 1 2 3 4}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in $VARS; do
    QT_LAB_LIB=$n rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lab$n" -- python3 "$R/tools/sweep_lab.py" $K $ROWS > "$OUT/lab$n.log" 2>&1
    echo "lab $n: $(grep -h 'sweep K=' "$OUT/lab$n.log" | tail -1)" >> "$OUT/summary.txt"
    python3 - "$OUT/lab$n" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_quad" in r["Kernel_Name"] or "sgemm_ring" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v3 = v[len(v) // 3:]          # launches of the timed repetitions (the first run warms up)
    print(f"   {k:40s} n {len(v3):4d} avg {sum(v3) / len(v3):8.1f} us  min {min(v3):8.1f}  max {max(v3):8.1f}")
PY
done
cat "$OUT/summary.txt"
