#!/usr/bin/env python3
"""The sweep with its far update on the f32-MFMA fmaf chain (the parity contract, default) and as a three-plane bf16
product (QT_SWEEP_FAR=bf16x3, opt-in), same process, same inputs: stage time and the share of integer levels that come
out different.  usage: sweep_far_ab.py [reps]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.hip import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
for K, R in ((4096, 4096), (4096, 28672), (14336, 4096), (8192, 10240)):
    X = synth_activations(4 * K, K, seed=K, device=dev)
    W = synth_weight(R, K, seed=R, device=dev)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    perm, inv = ops.argsort_desc(ops.hessian_diag(G, 8))
    A, dead, _ = ops.hessian_prepare(G, 8, 0.01, perm)
    U, info = ops.cholesky_inverse_upper(A)
    sc, zp, sct, zpt = ops.group_minmax_qparams(W, 128, True, 4)
    g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
    out = {}
    for mode in ("chain", "bf16x3"):
        if mode == "bf16x3":
            os.environ["QT_SWEEP_FAR"] = "bf16x3"
        else:
            os.environ.pop("QT_SWEEP_FAR", None)
        best = 1e9
        for _ in range(reps + 1):
            Wf = ops.weight_gather_f32(W, perm, dead)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            Qt, loss = ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[mode] = (best, Qt.clone(), loss.clone())
    os.environ.pop("QT_SWEEP_FAR", None)
    diff = float((out["chain"][1] != out["bf16x3"][1]).float().mean())
    rows = float(((out["chain"][1] != out["bf16x3"][1]).sum(0) == 0).float().mean())      # Qt is [K, R]
    dl = float((out["chain"][2] - out["bf16x3"][2]).abs().max() / out["chain"][2].abs().max())
    print(f"K={K:6d} R={R:6d}: chain {out['chain'][0]:7.3f} ms   bf16x3 far update {out['bf16x3'][0]:7.3f} ms   "
          f"levels that differ {diff:.2e} (rows equal in every column: {100 * rows:.1f} %), loss rel. diff {dl:.1e}", flush=True)
    del X, W, G, A, U
