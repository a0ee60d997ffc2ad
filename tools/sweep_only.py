#!/usr/bin/env python3
"""One sweep stage (after a warm-up run) for kernel traces.  usage: sweep_only.py K R [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.hip import ops

K, R = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
X = synth_activations(4 * K, K, seed=K, device=dev)
W = synth_weight(R, K, seed=R, device=dev)
G = torch.zeros((K, K), dtype=torch.float32, device=dev)
ops.xtx_accumulate(X, G)
perm, inv = ops.argsort_desc(ops.hessian_diag(G, 8))
A, dead, _ = ops.hessian_prepare(G, 8, 0.01, perm)
U, info = ops.cholesky_inverse_upper(A)
sc, zp, sct, zpt = ops.group_minmax_qparams(W, 128, True, 4)
g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
for i in range(reps + 1):
    Wf = ops.weight_gather_f32(W, perm, dead)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    Qt, loss = ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4)
    e1.record()
    torch.cuda.synchronize()
    print(f"sweep K={K} R={R}: {e0.elapsed_time(e1):.3f} ms", flush=True)
