set -u
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_token_split.py tests/test_gpu_oneshot_dist.py tests/test_gpu_kernels.py tests/test_gpu_fp32_activations.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for i in 1 2 3; do
  QT_XTX_ORDER=1 python3 tools/xtx_only.py 14336 3 | tail -1 | sed 's/^/pairs   /' >> $O/xtx_order_ab.txt
  python3 tools/xtx_only.py 14336 3 | tail -1 | sed 's/^/aligned /' >> $O/xtx_order_ab.txt
done
for i in 1 2; do
  QT_XTX_ORDER=1 python3 tools/xtx_only.py 4096 5 | tail -1 | sed 's/^/pairs   /' >> $O/xtx_order_ab.txt
  python3 tools/xtx_only.py 4096 5 | tail -1 | sed 's/^/aligned /' >> $O/xtx_order_ab.txt
done
cat $O/xtx_order_ab.txt
bash tools/xtx_pmc.sh $O/xtx_pmc_K14336 14336 > /dev/null 2>&1
cat $O/xtx_pmc_K14336/summary_K14336.md
rm -rf $O/xtx_pmc_K14336/sq $O/xtx_pmc_K14336/fetch $O/xtx_pmc_K14336/write
python3 bench.py --no-cpu-baseline --no-stage-split > $O/bench.json 2>/dev/null; cat $O/bench.json | cut -c1-300
