#!/bin/bash
# Round-3 artefacts of the final code (run on the GPU box through gpurun; outputs under gpurun_out/r3final).
# Every step writes its own file, so a late failure loses nothing.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final
mkdir -p "$O"
cd "$R"
echo "bench default" && python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
echo "bench per-sample" && python3 bench.py --accumulate per-sample --no-cpu-baseline > "$O/bench_persample.json" 2>/dev/null || exit 1
echo "bench awq" && python3 bench.py --method awq --no-cpu-baseline > "$O/bench_awq.json" 2>/dev/null || exit 1
echo "bench 70b" && python3 bench.py --model llama-3-70b --steps 4 --warmup 2 > "$O/bench_70b.json" 2>/dev/null || exit 1
echo "bench mixtral" && python3 bench.py --model mixtral-8x7b --steps 4 --warmup 2 > "$O/bench_mixtral.json" 2>/dev/null || exit 1
echo "bench 2 ranks (gloo rehearsal, self-launched)" && QT_BENCH_REHEARSE_GLOO=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench_rehearsal_2ranks_one_gpu.json" 2>/dev/null || exit 1
echo "full model" && python3 tools/full_model.py 32 2>&1 | grep -v "amdgpu.ids\|huggingface\|offline-mode" > "$O/full_model.txt" || exit 1
echo "stage times" && python3 tools/stage_times.py > "$O/stage_times.txt" 2>&1 || exit 1
echo "sweep a/b" && python3 tools/sweep_ab.py 3 > "$O/sweep_ab.txt" 2>&1 || exit 1
echo "sgemm k sweep" && { QT_SGEMM_RING=0 python3 tools/sgemm_k_sweep.py 2>&1 | sed "s/^/register-staged: /"; python3 tools/sgemm_k_sweep.py 2>&1 | sed "s/^/ring:            /"; } > "$O/sgemm_k_sweep.txt" || exit 1
for K in 4096 8192 14336; do
  python3 tools/chol_only.py $K 3 2>&1 | grep chol | tail -1 >> "$O/chol_times.txt"
done
echo "chol kernel breakdown" && tools/prof_kernels.sh r3final/chol14336 "" -- tools/chol_only.py 14336 2 > /dev/null || exit 1
python3 tools/trace_breakdown.py "$O/chol14336/prof" build_flipped > "$O/chol_kernel_breakdown_K14336.txt" || exit 1
tools/prof_kernels.sh r3final/chol4096 "" -- tools/chol_only.py 4096 2 > /dev/null || exit 1
python3 tools/trace_breakdown.py "$O/chol4096/prof" build_flipped > "$O/chol_kernel_breakdown_K4096.txt" || exit 1
echo "sweep kernel breakdown" && tools/prof_kernels.sh r3final/sweep14336 "" -- tools/sweep_only.py 14336 4096 1 > /dev/null || exit 1
python3 tools/trace_breakdown.py "$O/sweep14336/prof" gather_f32 > "$O/sweep_kernel_breakdown_K14336.txt" || exit 1
echo "pmc clock" && tools/pmc_clock.sh r3final/pmc_stage -- tools/stage_times.py --groups mlp_down || exit 1
for k in "sgemm_ring_kernel" "gemm3_kernel<0>" "gemm3_kernel<1>" "xtx_kernel" "sweep_quad_kernel" "potf2_kernel"; do
  python3 tools/pmc_clock.py "$O/pmc_stage" "$k" 20 >> "$O/gemm_pmc.txt"
done
echo "xtx pmc" && bash tools/xtx_pmc.sh "$O/xtx_pmc_K14336" 14336 > /dev/null 2>&1
bash tools/xtx_pmc.sh "$O/xtx_pmc_K4096" 4096 > /dev/null 2>&1
cp "$O/xtx_pmc_K14336/summary_K14336.md" "$O/xtx_pmc_K14336.md" 2>/dev/null
cp "$O/xtx_pmc_K4096/summary_K4096.md" "$O/xtx_pmc_K4096.md" 2>/dev/null
python3 tools/xtx_traffic_json.py "$O/xtx_pmc_K4096" "$O/xtx_pmc_K14336" > "$O/xtx_pmc_traffic.json" 2> "$O/xtx_pmc_traffic.err"
echo "bench profiled" && mkdir -p "$O/benchprof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/benchprof" -- python3 "$R/bench.py" --no-cpu-baseline --no-stage-split > "$O/bench_profiled.json" 2> "$O/bench_profiled.err" || exit 1
cd "$R"
python3 tools/xtx_trace_segments.py $(ls "$O"/benchprof/*/*kernel_trace.csv | head -1) "$O/bench_profiled.json" > "$O/bench_xtx_segments.md" 2>&1
cp $(ls "$O"/benchprof/*/*kernel_stats.csv | head -1) "$O/bench_kernel_stats.csv"
rm -rf "$O/benchprof" "$O/chol14336/prof" "$O/chol4096/prof" "$O/sweep14336/prof" "$O/pmc_stage" "$O"/xtx_pmc_K*/sq "$O"/xtx_pmc_K*/fetch "$O"/xtx_pmc_K*/write
ls -la "$O"
