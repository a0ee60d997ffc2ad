#!/bin/bash
# Round-2 artefacts of the final code (run on the GPU box through gpurun; outputs under gpurun_out/r2final)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2final
mkdir -p "$O"
cd "$R"
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
python3 bench.py --accumulate per-sample --no-cpu-baseline > "$O/bench_persample.json" 2>/dev/null || exit 1
python3 bench.py --method awq --steps 4 --warmup 1 --no-cpu-baseline > "$O/bench_awq.json" 2>/dev/null || exit 1
python3 tools/stage_times.py > "$O/stage_times.txt" 2>&1 || exit 1
python3 tools/gemm3_bench.py > "$O/gemm3_bench.txt" 2>&1 || exit 1
for K in 4096 8192 14336; do
  QT_CHOL_G3=0 python3 tools/chol_only.py $K 3 2>&1 | grep chol | tail -1 | sed "s/^/f32 chain (QT_CHOL_G3=0): /" >> "$O/chol_times.txt"
  python3 tools/chol_only.py $K 3 2>&1 | grep chol | tail -1 | sed "s/^/default:                   /" >> "$O/chol_times.txt"
done
tools/prof_kernels.sh r2final/chol "" -- tools/chol_only.py 14336 2 > /dev/null || exit 1
python3 tools/trace_breakdown.py "$O/chol/prof" build_flipped > "$O/chol_kernel_breakdown.txt" || exit 1
tools/pmc_clock.sh r2final/pmc_stage -- tools/stage_times.py --groups mlp_down || exit 1
for k in "sgemm_tn_kernel<128, 128, 0, true" "gemm3_kernel<0>" "gemm3_kernel<1>" "xtx_kernel" "sweep_block_kernel"; do
  python3 tools/pmc_clock.py "$O/pmc_stage" "$k" 20 >> "$O/gemm_pmc.txt"
done
mkdir -p "$O/benchprof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/benchprof" -- python3 "$R/bench.py" --no-cpu-baseline > "$O/bench_profiled.json" 2> "$O/bench_profiled.err" || exit 1
cd "$R"
python3 tools/xtx_trace_segments.py $(ls "$O"/benchprof/*/*kernel_trace.csv | head -1) "$O/bench_profiled.json" > "$O/bench_xtx_segments.md" 2>&1
cp $(ls "$O"/benchprof/*/*kernel_stats.csv | head -1) "$O/bench_kernel_stats.csv"
rm -rf "$O/benchprof" "$O/chol/prof" "$O/pmc_stage"
ls -la "$O"
