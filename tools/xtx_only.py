#!/usr/bin/env python3
"""X^T X alone at one K (for rocprofv3 --pmc passes and mapping experiments)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations
from quantool_amd.hip import ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
X = synth_activations(512 * 384, K, seed=1, device=dev)
G = torch.zeros(K, K, device=dev)
ops.xtx_accumulate(X[:8192], G)
for _ in range(reps):
    G.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.xtx_accumulate(X, G)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"K={K} xtx {ms:.3f} ms  {X.shape[0] * K * (K + 1) / ms / 1e9:.1f} TFLOP/s", flush=True)
