#!/usr/bin/env python3
"""BASELINE.json configs[2]: Llama-3-8B-shaped AWQ W4A16 g128, 20-point scale search, one decoder
layer's four mappings on one MI355X (diagnostic; the headline bench is GPTQ)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import synth_activations, synth_weight
from quantool_amd.engine.awq_linear import awq_quantize_group
from quantool_amd.engine.model_shapes import MODEL_SHAPES
from quantool_amd.engine.schemes import QuantArgs

dev = torch.device("cuda:0")
shape = MODEL_SHAPES["llama-3-8b"]
n_tokens = 512 * 384
qa = QuantArgs()
tot = 0.0
for gi, (gname, K, lins) in enumerate(shape.groups):
    X = synth_activations(n_tokens, K, seed=gi, device=dev)
    Ws = [synth_weight(R, K, seed=10 + i, device=dev) for i, (_, R) in enumerate(lins)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = awq_quantize_group(Ws, [X], qa)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    R = sum(w.shape[0] for w in Ws)
    print(f"{gname:9s} K={K:5d} R={R:5d}: {dt * 1e3:8.1f} ms  best ratio {int(res[0].best_ratio_idx)}/20  "
          f"({20 * R * K * (K + 1) / dt / 1e12:6.1f} TFLOP/s of D^T D Gram work incl. everything else)", flush=True)
    tot += dt
    del X, Ws, res
print(f"layer total {tot * 1e3:.1f} ms -> {shape.weights_per_layer / tot / 1e9:.3f} Gweights/s", flush=True)
