#!/bin/bash
# usage: tools/prof_kernels.sh <out-subdir under gpurun_out> <kernel-name substrings, comma separated> -- <python tool and args>
# kernel-trace of one tool run, then per-launch durations of the named kernels in <out>/durations.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; PATS=$2; shift 3
mkdir -p "$OUT/prof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$R/$1" "${@:2}" > "$OUT/run.log" 2>&1
python3 "$R/tools/kernel_durations.py" "$OUT/prof" ${PATS//,/ } > "$OUT/durations.txt"
wc -l "$OUT/durations.txt"
