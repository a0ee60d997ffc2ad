#!/usr/bin/env python3
"""One fp32 TN GEMM shape, a few launches (for rocprofv3 --pmc passes).  usage: M N k mode split reps"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

M, N, k, mode, split, reps = (int(x) for x in sys.argv[1:7])
dev = torch.device("cuda:0")
A = torch.randn(k, M, device=dev)
B = torch.randn(k, N, device=dev)
C = torch.randn(M, N, device=dev)
out = torch.empty_like(C)
for _ in range(reps):
    ops.sgemm_tn(A, B, C, mode, allow_split_k=bool(split), out=out)
torch.cuda.synchronize()
