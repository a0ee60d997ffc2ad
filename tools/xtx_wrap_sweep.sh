#!/bin/bash
# Locality ablation sweep (timing only; results are wrong by construction): time + L2 hit rate + fabric
# bytes of xtx_kernel<true> for several wrap windows.  usage: tools/xtx_wrap_sweep.sh <outdir> <K> w1 w2 ...
OUT=$1; K=$2; shift 2
mkdir -p "$OUT"
for W in "$@"; do
  export QT_XTX_ABLATE_WRAP=$W
  python3 tools/xtx_only.py "$K" 3 2>/dev/null | tail -1 | sed "s/^/wrap=$W  /"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/w$W/fetch" -- python3 tools/xtx_only.py "$K" 2 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/w$W/tcc" -- python3 tools/xtx_only.py "$K" 2 > /dev/null 2>&1
  python3 tools/xtx_pmc_summary.py "$OUT/w$W" "$K" | grep -E "TFLOP|clock|MFMA util|L2 hit|fabric" | sed "s/^/    /"
done
