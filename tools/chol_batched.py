#!/usr/bin/env python3
"""n equal-size factorisations: n single-problem chains one after another vs ONE batched chain (same stream).
usage: chol_batched.py K n [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

K, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((2 * K, K), generator=g, device=dev).to(torch.bfloat16)
G = torch.zeros((K, K), dtype=torch.float32, device=dev)
ops.xtx_accumulate(X, G)
A0, _, _ = ops.hessian_prepare(G, 8, 0.01, None)
Ab = torch.empty((n, K, K), dtype=torch.float32, device=dev)
Ub = torch.empty((n, K, K), dtype=torch.float32, device=dev)


def timed(fn):
    best = 1e9
    for _ in range(reps + 1):          # first repetition: workspace allocation, item tables
        for b in range(n):
            Ab[b].copy_(A0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def singles():
    for b in range(n):
        ops.cholesky_inverse_upper(Ab[b], U_out=Ub[b])


t1 = timed(singles)
U1 = Ub.clone()
t2 = timed(lambda: ops.cholesky_inverse_upper_batched(Ab, Ub))
same = torch.equal(U1, Ub)
print(f"K={K} n={n}: {n} single chains {t1:.2f} ms ({t1 / n:.2f} each) | one batched chain {t2:.2f} ms ({t2 / n:.2f} per problem) | "
      f"x{t1 / t2:.2f} | bit-identical: {same}")
