"""Readiness for the first RCCL run (BASELINE.json metric "@1/2/4/8 GPU"; configs[3] / configs[4] are 8-GPU
configurations): `bench.py --gpus 2` END TO END with two ranks sharing the box's one GPU (QT_BENCH_REHEARSE_GLOO=1:
gloo for the host side, RCCL refuses two ranks on one device), at the group structure of Llama-3-70B and of
Mixtral-8x7B with every width divided by 16 (QT_BENCH_SHAPE_SCALE): which groups exist, the batched chains per
in_features, ragged per-rank expert routing, and the final gather of every Linear's packed state to rank 0.
What this cannot show is the transport itself: no collective of this repo has run over RCCL / xGMI yet (the driver has
had no 8-GPU node); the launch path, rendezvous and control flow are what is rehearsed."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _run(args, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 prints the one line
    return lines[0]


@pytest.mark.parametrize("model,n_linears", [("llama-3-70b", 7), ("mixtral-8x7b", 28)])
def test_two_ranks_on_one_gpu_run_the_big_configurations_scaled_down(dev, model, n_linears):
    line = _run(["--gpus", "2", "--model", model, "--steps", "2", "--warmup", "1", "--samples", "16", "--no-cpu-baseline",
                 "--no-stage-split"], {"QT_BENCH_SHAPE_SCALE": "16"})
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["metric"].startswith("REHEARSAL") and "rehearsal" in line["data"]
    assert len(line["per_rank_compute_ms_per_step"]) == 2 and line["gather_ms"] >= 0.0
    assert line["value"] > 0 and line["roofline"]["launches"] > 0
    # the gather carried every Linear of both ranks' layers (bench.py asserts len(merged) == len(local) * world itself)
    assert f"{n_linears} Linears" in line["config"]["workload"]
