#!/usr/bin/env python3
"""Same-process timing of the bf16x3 product (csrc/gemm3_tn.hip) against the f32-MFMA GEMM (sgemm_tn) on the
two shapes that carry the K^3 of the Cholesky chain: a k = 256 update of a big square (C -= A^T B) and a
256-row block-row product with long k.  Times include the plane split of the test face."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

dev = torch.device("cuda:0")


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (k, M, N, kind) in ((256, 14080, 14080, 0), (256, 8192, 8192, 0), (512, 14080, 14080, 0), (14080, 256, 14080, 1),
                        (8192, 256, 8192, 1)):
    A = torch.randn(k, M, device=dev)
    B = torch.randn(k, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    flop = 2.0 * k * M * N
    d3 = t(lambda: ops.gemm3_tn(A, B, C, kind))
    if kind == 0:
        ds = t(lambda: ops.sgemm_tn(A, B, C, 0, out=C))
    else:
        ds = t(lambda: ops.sgemm_tn(A, B, None, 1, allow_split_k=True, out=C))
    print(f"k={k:6d} M={M:6d} N={N:6d} kind={kind}: gemm3 {d3 * 1e3:8.3f} ms {flop / d3 / 1e12:7.1f} TF/s | "
          f"sgemm {ds * 1e3:8.3f} ms {flop / ds / 1e12:7.1f} TF/s", flush=True)
