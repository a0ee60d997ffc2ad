#!/bin/bash
# PMC passes over the Gram kernel alone (tools/xtx_only.py K reps), separate runs per counter set
# (guide: TCC has 4 slots, FETCH_SIZE takes 3; never combine --pmc with --sys-trace).
#   tools/xtx_pmc.sh <outdir> <K> [env assignments are inherited]
set -u
OUT=$1; K=${2:-14336}
mkdir -p "$OUT"
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/xtx_only.py "$K" 2 > "$OUT/$name.log" 2>&1 || echo "pass $name failed" >> "$OUT/errors.log"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/xtx_pmc_summary.py "$OUT" "$K" > "$OUT/summary_K$K.md" 2>&1
cat "$OUT/summary_K$K.md"
