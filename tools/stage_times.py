#!/usr/bin/env python3
"""Per-stage device times of the per-Linear GPTQ path on a model's Linear groups (diagnostic), one stream, nothing
overlapped.  Every stage is run once untimed before it is timed (first calls pay for workspaces, item tables and
kernel-attribute set-up: round 3's table carried a 69 ms "sort" that was such a first call).  Second part: the groups of
equal in_features through the BATCHED chain (one batched factorisation + one stacked sweep) next to the sum of their
single-group times."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.engine.model_shapes import MODEL_SHAPES
from quantool_amd.hip import ops


def timed(fn, prepare=None, reps=1):
    """min over `reps` timed runs of fn(), after one untimed run; `prepare` re-creates inputs fn destroys."""
    best, out = 1e30, None
    for i in range(reps + 1):
        arg = prepare() if prepare is not None else None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(arg) if prepare is not None else fn()
        e1.record()
        torch.cuda.synchronize()
        if i > 0:
            best = min(best, e0.elapsed_time(e1))
    return out, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--groups", default="")
    ap.add_argument("--no-batched", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    shape = MODEL_SHAPES[args.model]
    n_tokens = args.samples * 384
    tot = {}
    kept = {}        # K -> list of (gname, G, perm, dead, W, R): inputs of the batched part
    for gi, (gname, K, lins) in enumerate(shape.groups):
        if args.groups and gname not in args.groups.split(","):
            continue
        X = synth_activations(n_tokens, K, seed=gi, device=dev)
        Ws = [synth_weight(R, K, seed=10 + i, device=dev) for i, (_, R) in enumerate(lins)]
        R = sum(w.shape[0] for w in Ws)
        G = torch.zeros((K, K), dtype=torch.float32, device=dev)

        def gram():
            G.zero_()
            ops.xtx_accumulate(X, G)
        _, t_xtx = timed(gram)
        t_xtx -= timed(lambda: G.zero_())[1]
        ops.xtx_accumulate(X, G.zero_())
        diag, t_diag = timed(lambda: ops.hessian_diag(G, args.samples))
        (perm, inv), t_sort = timed(lambda: ops.argsort_desc(diag))
        (A, dead, _), t_prep = timed(lambda: ops.hessian_prepare(G, args.samples, 0.01, perm))
        (U, info), t_chol = timed(lambda A_: ops.cholesky_inverse_upper(A_), prepare=lambda: ops.hessian_prepare(G, args.samples, 0.01, perm)[0])
        W = torch.cat(Ws, 0)
        Wf, t_gather = timed(lambda: ops.weight_gather_f32(W, perm, dead))
        (sc, zp, sct, zpt), t_qp = timed(lambda: ops.group_minmax_qparams(W, 128, True, 4))
        g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
        (Qt, loss), t_sweep = timed(lambda Wf_: ops.gptq_sweep(Wf_, U, sct, zpt, g_sweep, 128, 4),
                                    prepare=lambda: ops.weight_gather_f32(W, perm, dead))
        _, t_pack = timed(lambda: ops.pack_int4(Qt, inv))
        flops = n_tokens * K * (K + 1)
        row = dict(xtx=t_xtx, diag=t_diag, sort=t_sort, prep=t_prep, chol=t_chol, gather=t_gather, qparams=t_qp,
                   sweep=t_sweep, pack=t_pack)
        print(f"{gname:12s} K={K:5d} R={R:5d} info={int(info.item())} " + " ".join(f"{k}={v:7.2f}ms" for k, v in row.items())
              + f" | xtx {flops / t_xtx / 1e9:7.1f} TFLOP/s  chol {2 / 3 * K ** 3 / t_chol / 1e9:6.1f} TFLOP/s"
              + f"  sweep {R * K * K / t_sweep / 1e9:6.1f} TFLOP/s", flush=True)
        for k, v in row.items():
            tot[k] = tot.get(k, 0.0) + v
        if not args.no_batched:
            kept.setdefault(K, []).append(dict(name=gname, G=G, perm=perm, dead=dead, W=W, R=R, sct=sct, zpt=zpt, g_sweep=g_sweep,
                                               t_chol=t_chol, t_sweep=t_sweep))
        del X, A, U, Wf, Qt
    s = sum(tot.values())
    print("layer total " + " ".join(f"{k}={v:7.2f}ms" for k, v in tot.items()) + f" | sum={s:.1f} ms -> "
          f"{shape.weights_per_layer / s / 1e6:.2f} Gweights/s", flush=True)

    # ---- the groups of equal in_features through the batched chain ----
    saved = 0.0
    for K, grp in kept.items():
        for c0 in range(0, len(grp), ops.MAX_BATCH):
            chunk = grp[c0:c0 + ops.MAX_BATCH]
            n = len(chunk)
            if n < 2:
                continue
            Ab = torch.empty((n, K, K), dtype=torch.float32, device=dev)
            Ub = torch.empty((n, K, K), dtype=torch.float32, device=dev)

            def fill():
                for b, g in enumerate(chunk):
                    ops.hessian_prepare(g["G"], args.samples, 0.01, g["perm"], A_out=Ab[b])
            _, t_b = timed(lambda _: ops.cholesky_inverse_upper_batched(Ab, Ub), prepare=fill)
            R = sum(g["R"] for g in chunk)
            row_end = [sum(g["R"] for g in chunk[:i + 1]) for i in range(n)]
            sct = torch.cat([g["sct"] for g in chunk], 1).contiguous()
            zpt = torch.cat([g["zpt"] for g in chunk], 1).contiguous()
            gidx = torch.stack([g["g_sweep"] for g in chunk]).contiguous()

            def stacked():
                return torch.cat([ops.weight_gather_f32(g["W"], g["perm"], g["dead"]) for g in chunk], 0)
            _, t_s = timed(lambda Wf_: ops.gptq_sweep_grouped(Wf_, Ub, row_end, sct, zpt, gidx, 128, 4), prepare=stacked)
            t1c, t1s = sum(g["t_chol"] for g in chunk), sum(g["t_sweep"] for g in chunk)
            saved += t1c + t1s - t_b - t_s
            print(f"batched K={K:5d} x{n:2d} ({'+'.join(g['name'] for g in chunk)}; {R} rows): factor {t_b:7.2f} ms (singles {t1c:7.2f}) | "
                  f"sweep {t_s:7.2f} ms (singles {t1s:7.2f})", flush=True)
            del Ab, Ub
    if kept:
        print(f"layer total with batched chains: sum={s - saved:.1f} ms -> {shape.weights_per_layer / (s - saved) / 1e6:.2f} Gweights/s",
              flush=True)


if __name__ == "__main__":
    main()
