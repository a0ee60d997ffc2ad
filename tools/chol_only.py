#!/usr/bin/env python3
"""Cholesky-inverse chain alone (for rocprofv3 per-kernel breakdowns)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.randn(2 * K, K, device=dev).to(torch.bfloat16)
G = torch.zeros(K, K, device=dev)
ops.xtx_accumulate(X, G)
for _ in range(reps):
    A, dead, _ = ops.hessian_prepare(G, 8, 0.01, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    U, info = ops.cholesky_inverse_upper(A)
    e1.record()
    torch.cuda.synchronize()
    print(f"K={K} chol {e0.elapsed_time(e1):.2f} ms info={int(info.item())}", flush=True)
