"""Helpers shared by the tests: bf16 <-> numpy bit patterns, synthetic calibration data."""
import numpy as np
import torch


def bf16_tensor_to_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_bf16_tensor(b: np.ndarray, device=None) -> torch.Tensor:
    t = torch.from_numpy(b.view(np.int16).copy()).view(torch.bfloat16)
    return t.to(device) if device is not None else t


def synth_activations(n_tokens: int, K: int, seed: int = 2, outlier_frac: float = 0.01) -> np.ndarray:
    """X ~ N(0,1) with a fraction of channels scaled x10, as bf16 bit patterns."""
    from oracle import reference_path as rp

    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_tokens, K)).astype(np.float32)
    n_out = max(1, int(round(K * outlier_frac)))
    cols = rng.choice(K, size=n_out, replace=False)
    X[:, cols] *= 10.0
    return rp.f32_to_bf16_bits(X)


def synth_weight(R: int, K: int, seed: int = 0, std: float = 0.02) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((R, K)) * std).astype(np.float32)
