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


@pytest.mark.parametrize("mode", ["linears", "smooth", "module"])
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


def test_flatten_takes_a_result_dict_with_host_and_device_tensors(dev):
    """The RCCL branch of the final gather hands `_flatten` what `result_tensors` returns: packed words,
    scales on the GPU next to `weight_shape` on the host."""
    import torch

    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs
    from quantool_amd.engine.serialization import result_tensors
    from quantool_amd.engine.sharding import _flatten, _unflatten

    g = torch.Generator().manual_seed(0)
    acc = HessianAccumulator(256, dev)
    acc.add(torch.randn((4, 64, 256), generator=g).to(torch.bfloat16).to(dev))
    w = (torch.randn((40, 256), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    r = gptq_quantize_linear(w, acc, QuantArgs(num_bits=4, symmetric=False, group_size=128, actorder="group"))
    t = result_tensors(r)
    assert {v.device.type for v in t.values()} == {"cpu", "cuda"}
    flat, meta = _flatten({f"lin::{k}": v for k, v in t.items()}, dev)
    assert flat.device.type == "cuda" and flat.numel() % 16 == 0
    back = _unflatten(flat, meta)
    for k, v in t.items():
        assert torch.equal(back[f"lin::{k}"].cpu(), v.cpu()) and back[f"lin::{k}"].dtype == v.dtype
    flat2, _ = _flatten({f"lin::{k}": v for k, v in t.items()})          # device inferred: the accelerator
    assert flat2.device.type == "cuda"
