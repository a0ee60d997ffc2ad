#!/bin/bash
# usage: tools/pmc_clock.sh <out-subdir under gpurun_out> -- <python tool and args>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/$1" "${@:2}" > "$OUT/run.log" 2>&1
