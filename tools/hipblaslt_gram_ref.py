#!/usr/bin/env python3
"""Reference point for the Gram kernel: the same contraction as a plain library GEMM (torch.mm -> hipBLASLt), full
K x K output in bf16 (no symmetry, no fp32 result) -- what rate the vendor library sustains on this box for
X^T X at the bench's shapes.  Not used by the product.  usage: hipblaslt_gram_ref.py [K ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations
from quantool_amd.hip import ops

dev = torch.device("cuda:0")
N = 512 * 384
for K in [int(a) for a in sys.argv[1:]] or [4096, 14336]:
    X = synth_activations(N, K, seed=1, device=dev)
    Xt = X.t().contiguous()
    for name, fn, flops in (("torch.mm(X^T, X)  [K,N]x[N,K] bf16 out", lambda: torch.mm(Xt, X), 2.0 * N * K * K),
                            ("torch.mm(X.t(), X) (transposed view)     ", lambda: torch.mm(X.t(), X), 2.0 * N * K * K)):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"K={K:6d} {name}: {best:8.3f} ms  {flops / best / 1e9:8.1f} TFLOP/s (full square)", flush=True)
    G = torch.zeros(K, K, device=dev)
    ops.xtx_accumulate(X[:8192], G)
    best = 1e9
    for _ in range(3):
        G.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.xtx_accumulate(X, G)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"K={K:6d} qt_xtx_accumulate (lower triangle, fp32 out)      : {best:8.3f} ms  {N * K * (K + 1) / best / 1e9:8.1f} TFLOP/s "
          f"(algorithmic; {2.0 * N * K * K / best / 1e9:.0f} if it were credited the full square)", flush=True)
    del X, Xt, G
