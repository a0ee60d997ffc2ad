"""SURVEY 8e on the real backend, as far as one GPU goes: every collective of ``engine/sharding.py`` and of
``bench.py``'s N>1 path through a ONE-rank RCCL process group (``backend="nccl"``).

The two-rank tests (test_gpu_oneshot_dist.py, test_gpu_token_split.py, test_sharding_gloo.py) run over gloo because
RCCL refuses two ranks on one device; what they cannot show is that the tensors handed to the collectives live on
the device and carry dtypes RCCL accepts, and that bench.py's ``nccl`` branch (process-group creation bound to the
device, device-side barrier, the gather warm-up, max-over-ranks timing) runs at all.  One rank shows that; the
transfers between GPUs themselves stay unmeasured until an 8-GPU node runs SCALE."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
WORKER = Path(__file__).resolve().parent / "rccl_one_rank_worker.py"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharding_collectives_over_rccl(dev):
    p = subprocess.run([sys.executable, str(WORKER), str(_free_port())], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=420)
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-4000:]
    assert "[rccl] one-rank worker: ok" in p.stdout


def test_bench_line_through_a_one_rank_rccl_group(dev):
    env = dict(os.environ, QT_BENCH_ONE_RANK_RCCL="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-stage-split"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2
    assert "one-rank RCCL group" in line["config"]["sharding"]
    assert line["gather_ms"] >= 0.0 and len(line["per_rank_compute_ms_per_step"]) == 1
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
