#!/usr/bin/env python3
"""fp32 TN GEMM micro-benchmark on the shapes the Cholesky chain and the sweep's trailing update use."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

dev = torch.device("cuda:0")
shapes = [  # (M, N, k, mode, split, label)
    (4096, 4096, 4096, 1, False, "square 4096^3"),
    (128, 7168, 7168, 0, True, "potrf row-panel mid (split-K)"),
    (128, 12288, 2048, 0, True, "potrf early"),
    (128, 2048, 12288, 0, True, "potrf late"),
    (4096, 7168, 128, 0, False, "trailing down_proj mid"),
    (28672, 2048, 128, 0, False, "trailing gate_up mid"),
    (6144, 2048, 128, 0, False, "trailing qkv mid"),
]
for M, N, k, mode, split, label in shapes:
    A = torch.randn(k, M, device=dev)
    B = torch.randn(k, N, device=dev)
    C = torch.randn(M, N, device=dev)
    out = torch.empty_like(C)
    for _ in range(2):
        ops.sgemm_tn(A, B, C, mode, allow_split_k=split, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.sgemm_tn(A, B, C, mode, allow_split_k=split, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:32s} M={M:6d} N={N:6d} k={k:6d}: {ms * 1e3:8.1f} us  {2.0 * M * N * k / ms / 1e9:7.1f} TFLOP/s  "
          f"C-traffic {(2 if mode == 0 else 1) * M * N * 4 / ms / 1e6:7.1f} GB/s", flush=True)
