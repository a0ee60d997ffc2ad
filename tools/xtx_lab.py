#!/usr/bin/env python3
"""Gram-kernel lab: correctness screen + same-process A/B timing of xtx_kernel variants (diagnostic).

Variants are selected through the environment knobs the library reads per call
(QT_XTX_MAP, QT_XTX_ABLATE_WRAP).  Timings are interleaved
rounds in ONE process (guide rule 24); every variant is checked against an fp64 Gram matrix and
for run-to-run bitwise determinism (a race in the hand-ordered LDS-DMA pipeline shows as either).

  python tools/xtx_lab.py check            # sizes incl. ragged / tiny / edge-tile cases
  python tools/xtx_lab.py time [K ...]     # N = 196608 tokens, default K = 4096 14336
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from quantool_amd.hip import ops

DEV = torch.device("cuda:0")
VARIANTS = {
    "default": {},
    "ring": {"QT_XTX_SHAPE": "32"},
    "nothrottle": {"QT_XTX_SHAPE": "32", "QT_XTX_THROTTLE": "0"},
    "shape16": {"QT_XTX_SHAPE": "16"},
    "ring_map1": {"QT_XTX_MAP": "1"},
    "wrap8": {"QT_XTX_ABLATE_WRAP": "8"},          # timing-only ablations (wrong results)
    "wrap64": {"QT_XTX_ABLATE_WRAP": "64"},
}
CHECKED = [v for v in os.environ.get("XTX_LAB_VARIANTS", "default,ring,shape16").split(",") if v]


def setenv(v):
    for k in ("QT_XTX_MAP", "QT_XTX_ABLATE_WRAP", "QT_XTX_THROTTLE", "QT_XTX_SHAPE"):
        os.environ.pop(k, None)
    os.environ.update(VARIANTS[v])


def synth(n, K, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    X = torch.empty((n, K), dtype=torch.bfloat16, device=DEV)
    gain = torch.ones(K, device=DEV)
    gain[torch.randperm(K, generator=g, device=DEV)[: max(1, K // 100)]] = 10.0
    for t0 in range(0, n, 16384):
        t1 = min(n, t0 + 16384)
        X[t0:t1] = (torch.randn((t1 - t0, K), generator=g, device=DEV) * gain).to(torch.bfloat16)
    return X


def gram(X, K, variant):
    setenv(variant)
    G = torch.zeros((K, K), dtype=torch.float32, device=DEV)
    ops.xtx_accumulate(X, G)
    return G


def ref_err(X, G, K, blocks=12, seed=0):
    """max |G - fp64 Gram| / sqrt(G_ii G_jj) over sampled 128x128 blocks of the lower triangle."""
    g = torch.Generator().manual_seed(seed)
    d = None
    worst = 0.0
    nb = (K + 127) // 128
    picks = {(nb - 1, nb - 1), (0, 0), (nb - 1, 0)}
    while len(picks) < min(blocks, nb * (nb + 1) // 2):
        i = int(torch.randint(0, nb, (1,), generator=g))
        j = int(torch.randint(0, i + 1, (1,), generator=g))
        picks.add((i, j))
    for bi, bj in picks:
        i0, i1, j0, j1 = bi * 128, min(K, bi * 128 + 128), bj * 128, min(K, bj * 128 + 128)
        ref = torch.zeros((i1 - i0, j1 - j0), dtype=torch.float64, device=DEV)
        for t0 in range(0, X.shape[0], 65536):
            xs = X[t0:t0 + 65536]
            ref += xs[:, i0:i1].double().t() @ xs[:, j0:j1].double()
        if d is None:
            d = torch.zeros(K, dtype=torch.float64, device=DEV)
            for t0 in range(0, X.shape[0], 65536):
                d += (X[t0:t0 + 65536].double() ** 2).sum(0)
        got = G[i0:i1, j0:j1].double()
        if bi == bj:
            got, ref = torch.tril(got), torch.tril(ref)
        scale = torch.sqrt(d[i0:i1, None] * d[None, j0:j1]).clamp_min(1e-30)
        worst = max(worst, float(((got - ref).abs() / scale).max()))
    return worst


def check():
    cases = [(1000, 512), (64, 256), (17, 264), (4096 + 17, 768), (384, 1024), (8192, 1096), (6 * 64, 4096),
             (32768, 4096), (12288, 3072), (196608, 4096), (16384 + 40, 5120)]
    if "--big" in sys.argv:
        cases.append((196608, 14336))
    ok = True
    for n, K in cases:
        X = synth(n, K, seed=n + K)
        row = []
        for v in CHECKED:
            G0 = gram(X, K, v)
            e = ref_err(X, G0, K)
            same = True
            for _ in range(3 if n * K < 5e8 else 1):
                same &= bool(torch.equal(G0, gram(X, K, v)))
            # accumulate twice into the same G: second call adds on top of the first
            setenv(v)
            G2 = G0.clone()
            ops.xtx_accumulate(X, G2)
            dbl = float((torch.tril(G2) - 2 * torch.tril(G0)).abs().max() / torch.tril(G0).abs().max())
            good = e <= 1e-5 and same and dbl <= 1e-6
            ok &= good
            row.append(f"{v}: {e:.1e} {'ok' if good else f'FAIL det={same} 2x={dbl:.1e}'}")
        # strided input (ldx > K)
        Xw = torch.zeros((n, K + 64), dtype=torch.bfloat16, device=DEV)
        Xw[:, :K] = X
        setenv(CHECKED[-1])
        Gs = torch.zeros((K, K), dtype=torch.float32, device=DEV)
        ops.xtx_accumulate(Xw[:, :K], Gs)
        es = ref_err(X, Gs, K)
        ok &= es <= 1e-5
        print(f"N={n:7d} K={K:6d} | " + " | ".join(row) + f" | ring strided err {es:.2e}", flush=True)
        del X, Xw
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def timeit(Ks, n=196608, rounds=5, variants=None):
    variants = variants or [v for v in os.environ.get("XTX_LAB_TIME", "").split(",") if v] or CHECKED + ["wrap8", "wrap64"]
    for K in Ks:
        X = synth(n, K, seed=K)
        G = torch.zeros((K, K), dtype=torch.float32, device=DEV)
        res = {v: [] for v in variants}
        for v in variants:       # warm
            setenv(v)
            ops.xtx_accumulate(X, G)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for v in variants:
                setenv(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.xtx_accumulate(X, G)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1))
        flops = n * K * (K + 1)
        for v in variants:
            ts = sorted(res[v])
            med = ts[len(ts) // 2]
            print(f"K={K:6d} N={n} {v:10s} median {med:8.3f} ms  min {ts[0]:8.3f} ms  "
                  f"{flops / med / 1e9:7.1f} TFLOP/s (median)  {flops / ts[0] / 1e9:7.1f} (min)", flush=True)
        del X, G


def staged(K=14336, n=196608, T=384):
    """Per-sample accumulation through HessianAccumulator's token staging vs one launch."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator

    setenv("default")
    X = synth(n, K, seed=K + 1)
    one = HessianAccumulator(K, DEV, stage_tokens=0)
    st = HessianAccumulator(K, DEV)
    for acc, per_sample in ((one, False), (st, True), (one, False), (st, True)):
        acc.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if per_sample:
            for i in range(0, n, T):
                acc.add(X[i:i + T])
        else:
            acc.add(X, num_samples=n // T)
        _ = acc.G
        torch.cuda.synchronize()
        print(f"K={K} {'per-sample staged' if per_sample else 'single launch':18s} {1e3 * (time.perf_counter() - t0):8.2f} ms "
              f"(n={acc.n})", flush=True)
    rel = float((torch.tril(one.G) - torch.tril(st.G)).abs().max() / torch.tril(one.G).abs().max())
    print(f"staged vs single: max rel diff {rel:.2e}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(0 if check() else 1)
    elif mode == "time":
        Ks = [int(a) for a in sys.argv[2:] if a.isdigit()] or [4096, 14336]
        timeit(Ks)
    elif mode == "staged":
        for K in ([int(a) for a in sys.argv[2:] if a.isdigit()] or [4096, 14336]):
            staged(K)
