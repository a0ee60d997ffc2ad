#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/reference_path.py).

SELF-GENERATED, UPSTREAM-UNVERIFIED: the reference's implementation of this path
(llmcompressor / compressed-tensors) is not available offline and the reference holds no golden
vectors for it, so these fixtures pin the oracle against regressions and give the
HIP path fixed inputs/outputs; they do not pin it against upstream.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import reference_path as rp  # noqa: E402

OUT = Path(__file__).resolve().parent


def case(name, R, K, S, T, sym, actorder, seed):
    rng = np.random.default_rng(seed)
    W = rp.bf16_bits_to_f32(rp.f32_to_bf16_bits((rng.standard_normal((R, K)) * 0.02).astype(np.float32)))
    X = rng.standard_normal((S * T, K)).astype(np.float32)
    cols = rng.choice(K, size=max(1, K // 100), replace=False)
    X[:, cols] *= 10
    xb = rp.f32_to_bf16_bits(X)
    G = rp.gram_f64(xb)
    H = rp.hessian_from_gram(G, S)
    o = rp.quantize_weight(W, H, group_size=128, symmetric=sym, num_bits=4, actorder=actorder, inverse="lapack")
    np.savez_compressed(
        OUT / f"{name}.npz", W=W, X_bf16=xb, n_samples=np.int64(S), H=H, q=o["q"], packed=rp.pack_int4(o["q"]),
        scale=o["scale"], zp=o["zp"], g_idx=(o["g_idx"] if o["g_idx"] is not None else np.zeros(0, np.int32)),
        perm=(o["perm"].astype(np.int32) if o["perm"] is not None else np.zeros(0, np.int32)), U=o["U"],
        w_dq=o["w_dq"], loss=o["loss"], symmetric=np.bool_(sym), actorder=np.str_(str(actorder)))


def awq_case(name, R, K, N, seed):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((R, K)) * 0.03).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, rng.choice(K, size=3, replace=False)] *= 20
    xb = rp.f32_to_bf16_bits(X)
    r = rp.awq_best_scale(xb, [W], 128)
    np.savez_compressed(OUT / f"{name}.npz", W=W, X_bf16=xb, x_mean=r["x_mean"], w_mean=r["w_mean"],
                        losses=r["losses"], best_ratio_idx=np.int64(r["best_ratio_idx"]), best_scales=r["best_scales"])


if __name__ == "__main__":
    case("gptq_w4a16_static_16x256", 16, 256, 4, 128, True, "static", 11)
    case("gptq_w4a16_none_16x256", 16, 256, 4, 128, True, None, 12)
    case("gptq_w4a16_group_24x384", 24, 384, 4, 160, True, "group", 13)
    case("gptq_w4a16_asym_static_16x256", 16, 256, 4, 128, False, "static", 14)
    awq_case("awq_w4a16_32x256", 32, 256, 768, 15)
    print("golden fixtures written to", OUT)
