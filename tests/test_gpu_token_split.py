"""SURVEY 8e partitioning B: the calibration tokens of ONE Linear group split over two ranks --
per-rank Gram sums, one all-reduce, replicated factorisation, row-split sweep, all-gather of the
packed rows -- against a single-process run on the same summed Gram matrix (bit-exact).  The box has
one GPU: both ranks use it and gloo carries the collectives (RCCL refuses two ranks on one device)."""
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
WORKER = Path(__file__).resolve().parent / "token_split_worker.py"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("actorder,symmetric", [("static", True), ("group", False)])
def test_token_split_two_ranks_matches_single_process(dev, actorder, symmetric):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(WORKER), str(r), "2", str(port), actorder, "1" if symmetric else "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exited {p.returncode}:\n{out[-3000:]}"
