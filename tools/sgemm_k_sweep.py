#!/usr/bin/env python3
"""MODE_SUB fp32 TN GEMM at the sweep far update's shape for growing k: separates the k-loop's MFMA rate from the
per-workgroup prologue / epilogue / launch-tail cost (diagnostic)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

dev = torch.device("cuda:0")
K = 14336
W = torch.randn(4096, K, device=dev)
for M, N in ((4096, 13824), (4096, 8192), (4096, 2560), (28672, 3584)):
    Wm = torch.randn(M, K, device=dev) if M != 4096 else W
    C = Wm[:, K - N:]
    for k in (128, 512, 2048, 8192):
        A = torch.randn(k, M, device=dev)
        B = torch.randn(k, K, device=dev)[:, K - N:]
        for _ in range(2):
            ops.sgemm_tn(A, B, C, 0, out=C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ops.sgemm_tn(A, B, C, 0, out=C)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"M={M:6d} N={N:6d} k={k:5d}: {ms * 1e3:9.1f} us  {2.0 * M * N * k / ms / 1e9:7.1f} TFLOP/s", flush=True)
        C.normal_()
