set -u
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gemm3.py tests/test_gpu_kernels.py -x -q -k "chol or gemm3 or factor" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/tests.log
tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2 3; do
  for K in 14336 8192; do
    QT_CHOL_G3_SPANS=0 python3 tools/chol_only.py $K 3 | tail -1 | sed 's/^/per-tile /' >> $O/chol_spans_ab.txt
    python3 tools/chol_only.py $K 3 | tail -1 | sed 's/^/spans    /' >> $O/chol_spans_ab.txt
  done
done
cat $O/chol_spans_ab.txt
for i in 1 2; do
  for shape in "14336 4096" "4096 28672" "4096 6144"; do
    for b in 4 8; do
      QT_SWEEP_BATCH=$b python3 tools/sweep_only.py $shape 3 | tail -1 | sed "s/^/batch $b /" >> $O/sweep_batch_ab.txt
    done
  done
done
cat $O/sweep_batch_ab.txt
tools/prof_kernels.sh r3d/chol14336 "" -- tools/chol_only.py 14336 2 > /dev/null
python3 tools/trace_breakdown.py $O/chol14336/prof build_flipped > $O/chol_kernel_breakdown_K14336.txt; head -4 $O/chol_kernel_breakdown_K14336.txt
rm -rf $O/chol14336
python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench.json 2>/dev/null; cut -c1-260 $O/bench.json
QT_CHOL_G3_SPANS=0 python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench_nospans.json 2>/dev/null; cut -c1-260 $O/bench_nospans.json
QT_SWEEP_BATCH=8 python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench_batch8.json 2>/dev/null; cut -c1-260 $O/bench_batch8.json
python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench2.json 2>/dev/null; cut -c1-260 $O/bench2.json
