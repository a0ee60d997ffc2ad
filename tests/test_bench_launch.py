"""`python bench.py --gpus N` must produce an N-rank job BY ITSELF (BASELINE.json metric "@1/2/4/8 GPU"):
with no launcher around it, it starts its own N ranks; with fewer than N GPUs visible it refuses with
a non-zero exit instead of printing an `n_gpus: 1` line."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus2_spawns_two_ranks_without_a_launcher():
    r = _run(["--gpus", "2", "--launch-probe"], {"QT_BENCH_REHEARSE_GLOO": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    assert lines[0] == {"probe": True, "n_gpus": 2, "rank_sum": 3, "self_launched": True}


def test_refuses_when_fewer_gpus_are_visible():
    import torch

    if torch.cuda.device_count() >= 64:
        return
    r = _run(["--gpus", "64", "--launch-probe"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and r.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--launch-probe"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "8", "--launch-probe"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
