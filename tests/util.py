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


def hook_inputs(model, module, data, dev, batched=False):
    """What a plain forward-pre-hook on ``module`` sees over the calibration rows: list of [T, K] (or [B * T, K]).
    ``batched``: the (equal-length) rows go through the model in ONE forward, the way the sequential driver's
    default mode stacks equal-shape samples (``engine.sequential.merge_cache``)."""
    got = []
    hk = module.register_forward_pre_hook(lambda m, a: got.append(a[0].detach().reshape(-1, a[0].shape[-1]).clone()))
    with torch.no_grad():
        if batched:
            ids = torch.stack([row["input_ids"].reshape(-1) for row in data]).to(dev)
            model(input_ids=ids, use_cache=False)
        else:
            for row in data:
                model(input_ids=row["input_ids"].reshape(1, -1).to(dev), use_cache=False)
    hk.remove()
    return got


def oracle_group(oracle, acts, weights, k, actorder="static"):
    """``k``: one entry of ``engine.sequential.DEBUG_KEEP`` (G, U, perm, n of an input group).  The
    driver's Gram sum must be the Gram sum of what the plain hook saw (fp64 reference, 1e-5); then
    ``oracle.quantize_weight`` for every weight of the group, given the GPU's factor."""
    x = torch.cat(acts).double().cpu().numpy()
    Gt = x.T @ x
    Gl = np.tril(k["G"].cpu().numpy())
    d = np.sqrt(np.diag(Gt))
    assert np.all(np.abs(Gl - np.tril(Gt)) <= 1e-5 * np.tril(np.outer(d, d)) + 1e-30)
    H = oracle.hessian_from_gram_f32(Gl + np.tril(Gl, -1).T, k["n"])
    U = k["U"].cpu().numpy()
    return [oracle.quantize_weight(w.float().cpu().numpy(), H, actorder=actorder, U_override=U) for w in weights]
