set -u
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm3.py tests/test_gpu_factor_fullsize.py tests/test_gpu_fp32_activations.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/tests.log
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3; do
  for K in 14336 8192; do
    QT_CHOL_G3_SPANS=0 python3 tools/chol_only.py $K 3 | tail -1 | sed 's/^/per-tile /' >> $O/chol_spans_ab.txt
    python3 tools/chol_only.py $K 3 | tail -1 | sed 's/^/spans    /' >> $O/chol_spans_ab.txt
  done
done
cat $O/chol_spans_ab.txt
for i in 1 2 3; do
  for K in 4096 8192; do
    QT_XTX_ORDER=1 python3 tools/xtx_only.py $K 5 | tail -1 | sed 's/^/pairs   /' >> $O/xtx_order_ab.txt
    QT_XTX_ORDER=2 python3 tools/xtx_only.py $K 5 | tail -1 | sed 's/^/aligned /' >> $O/xtx_order_ab.txt
  done
done
cat $O/xtx_order_ab.txt
tools/prof_kernels.sh r3c/chol14336 "" -- tools/chol_only.py 14336 2 > /dev/null
python3 tools/trace_breakdown.py $O/chol14336/prof build_flipped > $O/chol_kernel_breakdown_K14336.txt; head -8 $O/chol_kernel_breakdown_K14336.txt
rm -rf $O/chol14336
python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench.json 2>/dev/null; cut -c1-260 $O/bench.json
QT_CHOL_G3_SPANS=0 python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench_nospans.json 2>/dev/null; cut -c1-260 $O/bench_nospans.json
