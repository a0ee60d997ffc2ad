#!/bin/bash
# Round-4 artefacts of the final code (run on the GPU box through gpurun; outputs under gpurun_out/r4final).
# Every step writes its own file, so a late failure loses nothing.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${COLLECT_DIR:-r4final}
mkdir -p "$O"
cd "$R"
echo "bench default" && python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
echo "bench 32 steps" && python3 bench.py --steps 32 --warmup 3 --no-cpu-baseline --no-stage-split > "$O/bench_32steps.json" 2>/dev/null || exit 1
echo "bench a/b: a chain per group (round 3), 2 layers in flight" && QT_BATCH_CHAINS=0 python3 bench.py --steps 20 --warmup 3 --lanes 2 --no-cpu-baseline --no-stage-split > "$O/bench_ab_unbatched_lanes2.json" 2>/dev/null
echo "bench a/b: batched, 2 layers in flight" && python3 bench.py --steps 20 --warmup 3 --lanes 2 --no-cpu-baseline --no-stage-split > "$O/bench_ab_batched_lanes2.json" 2>/dev/null
echo "bench a/b: unbatched, 3 layers in flight" && QT_BATCH_CHAINS=0 python3 bench.py --steps 20 --warmup 3 --lanes 3 --no-cpu-baseline --no-stage-split > "$O/bench_ab_unbatched_lanes3.json" 2>/dev/null
echo "diag gram only / chains only" && QT_BENCH_SKIP=chain python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-stage-split > "$O/diag_gram_only.json" 2>/dev/null
QT_BENCH_SKIP=gram python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-stage-split > "$O/diag_chains_only.json" 2>/dev/null
echo "bench per-sample" && python3 bench.py --accumulate per-sample --no-cpu-baseline --no-stage-split > "$O/bench_persample.json" 2>/dev/null || exit 1
echo "bench awq" && python3 bench.py --method awq --no-cpu-baseline > "$O/bench_awq.json" 2>/dev/null || exit 1
echo "bench 70b" && python3 bench.py --model llama-3-70b --steps 4 --warmup 2 --no-cpu-baseline > "$O/bench_70b.json" 2>/dev/null || exit 1
echo "bench mixtral" && python3 bench.py --model mixtral-8x7b --steps 4 --warmup 2 --no-cpu-baseline > "$O/bench_mixtral.json" 2>/dev/null || exit 1
echo "bench mixtral unbatched" && QT_BATCH_CHAINS=0 python3 bench.py --model mixtral-8x7b --steps 4 --warmup 2 --no-cpu-baseline --no-stage-split > "$O/bench_mixtral_unbatched.json" 2>/dev/null
echo "bench 2 ranks (gloo rehearsal, self-launched)" && QT_BENCH_REHEARSE_GLOO=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-stage-split > "$O/bench_rehearsal_2ranks_one_gpu.json" 2>/dev/null || exit 1
echo "full model 8b" && python3 tools/full_model.py 32 2>&1 | grep -v "amdgpu.ids\|huggingface\|offline-mode" > "$O/full_model_8b.txt" || exit 1
echo "full model mixtral" && python3 tools/full_model.py 32 512 384 smoothquant W4A8 mixtral 2>&1 | grep -v "amdgpu.ids\|huggingface\|offline-mode" > "$O/full_model_mixtral.txt"
echo "stage times" && python3 tools/stage_times.py > "$O/stage_times_8b.txt" 2>&1 || exit 1
python3 tools/stage_times.py --model mixtral-8x7b 2>&1 | grep -v amdgpu | tail -4 > "$O/stage_times_mixtral_tail.txt"
python3 tools/stage_times.py --model llama-3-70b 2>&1 | grep -v amdgpu > "$O/stage_times_70b.txt"
echo "chol batched" && for a in "4096 3" "4096 10" "8192 3" "14336 8"; do python3 tools/chol_batched.py $a 2>&1 | grep -v amdgpu; done > "$O/chol_batched_vs_single.txt"
echo "chol batched kernel breakdown" && mkdir -p "$O/cholprof"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/cholprof" -- python3 "$R/tools/chol_batched.py" 4096 3 1 > "$O/cholprof.log" 2>&1)
DB=$(ls "$O"/cholprof/*/*results.db | head -1)
{ echo "# one batched chain of 3 x K = 4096 (rocprofv3 --kernel-trace, tools/rocpd_stats.py)"; python3 tools/rocpd_stats.py "$DB" --last-chain flat_reverse --gridz 3; echo; echo "# one single-problem chain, K = 4096"; python3 tools/rocpd_stats.py "$DB" --last-chain flat_reverse --gridz 1; } > "$O/chol_kernel_stats_K4096_batched.txt"
if [ "${COLLECT_PMC:-1}" = 1 ]; then
echo "xtx pmc" && bash tools/xtx_pmc.sh "$O/xtx_pmc_K14336" 14336 > /dev/null 2>&1
bash tools/xtx_pmc.sh "$O/xtx_pmc_K4096" 4096 > /dev/null 2>&1
cp "$O/xtx_pmc_K14336/summary_K14336.md" "$O/xtx_pmc_K14336.md" 2>/dev/null
cp "$O/xtx_pmc_K4096/summary_K4096.md" "$O/xtx_pmc_K4096.md" 2>/dev/null
python3 tools/xtx_traffic_json.py "$O/xtx_pmc_K4096" "$O/xtx_pmc_K14336" > "$O/xtx_pmc_traffic.json" 2> "$O/xtx_pmc_traffic.err"
fi
echo "bench profiled" && mkdir -p "$O/benchprof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/benchprof" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-stage-split > "$O/bench_profiled.json" 2> "$O/bench_profiled.err" || exit 1
cd "$R"
python3 tools/xtx_trace_segments.py $(ls "$O"/benchprof/*/*kernel_trace.csv | head -1) "$O/bench_profiled.json" > "$O/bench_xtx_segments.md" 2>&1
cp $(ls "$O"/benchprof/*/*kernel_stats.csv | head -1) "$O/bench_kernel_stats.csv"
rm -rf "$O/benchprof" "$O/cholprof" "$O"/xtx_pmc_K*/sq "$O"/xtx_pmc_K*/fetch "$O"/xtx_pmc_K*/write
ls -la "$O"
