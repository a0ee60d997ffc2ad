#!/usr/bin/env python3
"""Per-stage device times of the per-Linear GPTQ path on Llama-3-8B-shaped groups (diagnostic)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.engine.model_shapes import MODEL_SHAPES
from quantool_amd.hip import ops


def timed(fn, reps=1):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--groups", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    shape = MODEL_SHAPES[args.model]
    n_tokens = args.samples * 384
    tot = {}
    for gi, (gname, K, lins) in enumerate(shape.groups):
        if args.groups and gname not in args.groups.split(","):
            continue
        X = synth_activations(n_tokens, K, seed=gi, device=dev)
        Ws = [synth_weight(R, K, seed=10 + i, device=dev) for i, (_, R) in enumerate(lins)]
        R = sum(w.shape[0] for w in Ws)
        G = torch.zeros((K, K), dtype=torch.float32, device=dev)
        ops.xtx_accumulate(X[:4096], G)  # warm
        G.zero_()
        _, t_xtx = timed(lambda: ops.xtx_accumulate(X, G))
        diag, t_diag = timed(lambda: ops.hessian_diag(G, args.samples))
        perm, t_sort = timed(lambda: torch.argsort(diag, descending=True, stable=True).to(torch.int32))
        (A, dead, _), t_prep = timed(lambda: ops.hessian_prepare(G, args.samples, 0.01, perm))
        ops.cholesky_inverse_upper(A)           # warm: workspace allocation and the per-K item tables (A is consumed)
        (A, dead, _) = ops.hessian_prepare(G, args.samples, 0.01, perm)
        (U, info), t_chol = timed(lambda: ops.cholesky_inverse_upper(A))
        W = torch.cat(Ws, 0)
        Wf, t_gather = timed(lambda: ops.weight_gather_f32(W, perm, dead))
        (sc, zp, sct, zpt), t_qp = timed(lambda: ops.group_minmax_qparams(W, 128, True, 4))
        g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
        (Qt, loss), t_sweep = timed(lambda: ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4))
        inv = torch.empty_like(perm)
        inv[perm.long()] = torch.arange(K, dtype=torch.int32, device=dev)
        _, t_pack = timed(lambda: ops.pack_int4(Qt, inv))
        flops = n_tokens * K * (K + 1)
        row = dict(xtx=t_xtx, diag=t_diag, sort=t_sort, prep=t_prep, chol=t_chol, gather=t_gather, qparams=t_qp,
                   sweep=t_sweep, pack=t_pack)
        print(f"{gname:9s} K={K:5d} R={R:5d} info={int(info.item())} " + " ".join(f"{k}={v:8.2f}ms" for k, v in row.items())
              + f" | xtx {flops / t_xtx / 1e9:7.1f} TFLOP/s  chol {2 / 3 * K ** 3 / t_chol / 1e9:6.1f} TFLOP/s"
              + f"  sweep {R * K * K / t_sweep / 1e9:6.1f} TFLOP/s", flush=True)
        for k, v in row.items():
            tot[k] = tot.get(k, 0.0) + v
        del X, G, A, U, Wf, Qt
    s = sum(tot.values())
    print("layer total " + " ".join(f"{k}={v:8.2f}ms" for k, v in tot.items()) + f" | sum={s:.1f} ms -> "
          f"{shape.weights_per_layer / s / 1e6:.2f} Gweights/s", flush=True)


if __name__ == "__main__":
    main()
