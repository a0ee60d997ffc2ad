set -u
O=gpurun_out/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit 1
bash tools/collect_r03.sh > $O/collect.log 2>&1; echo "collect rc=$?"; tail -5 $O/collect.log
