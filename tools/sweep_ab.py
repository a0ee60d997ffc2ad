#!/usr/bin/env python3
"""A/B of the sweep stage's block kernel forms (QT_SWEEP_BLOCK=row|quad) on Llama-3-8B-shaped groups: same
process, interleaved rounds, outputs compared bit for bit.  usage: sweep_ab.py [rounds]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.hip import ops


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    for K, R in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 57344)):
        n_tok = 4 * K
        X = synth_activations(n_tok, K, seed=K, device=dev)
        W = synth_weight(R, K, seed=R, device=dev)
        G = torch.zeros((K, K), dtype=torch.float32, device=dev)
        ops.xtx_accumulate(X, G)
        diag = ops.hessian_diag(G, 8)
        perm, inv = ops.argsort_desc(diag)
        A, dead, _ = ops.hessian_prepare(G, 8, 0.01, perm)
        U, info = ops.cholesky_inverse_upper(A)
        sc, zp, sct, zpt = ops.group_minmax_qparams(W, 128, True, 4)
        g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
        out = {}
        times = {"row": [], "quad": []}
        for rd in range(rounds + 1):
            for mode in ("row", "quad"):
                os.environ["QT_SWEEP_BLOCK"] = mode
                Wf = ops.weight_gather_f32(W, perm, dead)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                Qt, loss = ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4)
                e1.record()
                torch.cuda.synchronize()
                if rd:
                    times[mode].append(e0.elapsed_time(e1))
                out[mode] = (Qt, loss, Wf)
        same = all(torch.equal(a, b) for a, b in zip(out["row"], out["quad"]))
        print(f"K={K:6d} R={R:6d}: row {min(times['row']):8.3f} ms  quad {min(times['quad']):8.3f} ms  "
              f"(medians {sorted(times['row'])[len(times['row']) // 2]:.3f} / {sorted(times['quad'])[len(times['quad']) // 2]:.3f})  "
              f"Qt/loss/W identical: {same}", flush=True)
        del X, G, A, U, W


if __name__ == "__main__":
    main()
