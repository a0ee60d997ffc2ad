"""SURVEY 8e behind the plugin API: ``oneshot`` under torch.distributed with two ranks.

* per-Linear calibration set: the heavy group is split over the ranks (partitioning B), the light ones
  are whole units on different ranks (A), rank 0 gathers and writes the complete state -- packed words
  and scales bit-identical to single-process runs (for the split group: on the Gram sums added in
  rank order);
* nn.Module path through the ``gptq`` plugin: the calibration samples are split over the ranks, every
  rank ends with the same quantised model, rank 0 writes it.

The box has one GPU: both ranks use it and gloo carries the collectives (RCCL refuses two ranks on one
device); on a multi-GPU node the same code runs over RCCL with one rank per GPU."""
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
WORKER = Path(__file__).resolve().parent / "oneshot_dist_worker.py"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["linears", "module"])
def test_oneshot_two_ranks(dev, tmp_path, mode):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(WORKER), mode, str(r), "2", str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exited {p.returncode}:\n" + "\n".join(o[-3000:] for o in outs)
