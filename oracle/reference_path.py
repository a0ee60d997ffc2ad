"""CPU restatement of the GPTQ / AWQ / SmoothQuant per-linear hot path.

TEST INFRASTRUCTURE -- only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product path (``quantool_amd``) never does.

PARITY UNPINNED.  quantool delegates this arithmetic to ``llmcompressor>=0.8.1`` /
``compressed-tensors`` (reference ``pyproject.toml:49-51``; call site
``src/quantool/methods/llm_compressor/base.py:161``), neither of which is vendored in the
reference, installed in the build image, or fetchable.  No reference test holds a golden vector
for this path.  Everything here restates the published upstream algorithm
as recalled in SURVEY.md Appendix A, plus hand-derived known-answer tests (``tests/``).

Functions cite the SURVEY row (a7..a14) and the reference call site that reaches them.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

F32_EPS = np.float32(np.finfo(np.float32).eps)


def build(force: bool = False) -> Path:
    so = _HERE / "_build" / "liboracle.so"
    src = _HERE / "gptq_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_HERE)], stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = _HERE / "_build" / "liboracle.so"
        if not so.exists():
            build()
        _LIB = ctypes.CDLL(str(so))
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------
# bf16 helpers (numpy has no bf16: carry it as uint16 bit patterns)
# ---------------------------------------------------------------------------------------------
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


# ---------------------------------------------------------------------------------------------
# a7  make_empty_hessian / accumulate_hessian            [SURVEY A.2; base.py:161 -> GPTQ hook]
# ---------------------------------------------------------------------------------------------
def accumulate_hessian_reference(batches_bf16, K: int) -> np.ndarray:
    """Upstream's running update, one calibration sample per call, in fp32:
    ``H *= n/(n+1); n += 1; x = sqrt(2/n) * x.float(); H += x.T @ x``  (matmul order = BLAS's).
    ``batches_bf16``: iterable of uint16 arrays [T_b, K]."""
    H = np.zeros((K, K), dtype=np.float32)
    n = 0
    for xb in batches_bf16:
        x = bf16_bits_to_f32(xb).reshape(-1, K)
        H *= np.float32(n / (n + 1))
        n += 1
        x = np.float32(np.sqrt(2.0 / n)) * x
        H += x.T @ x
    return H


def gram_f64(x_bf16: np.ndarray) -> np.ndarray:
    """Exact-product / fp64-sum Gram matrix G = X^T X (full symmetric, float64)."""
    x = bf16_bits_to_f32(x_bf16).astype(np.float64)
    return x.T @ x


def gram_f64_c(x_bf16: np.ndarray) -> np.ndarray:
    """Same through the C restatement (lower triangle), used to cross-check the two."""
    n, K = x_bf16.shape
    x = np.ascontiguousarray(x_bf16)
    G = np.zeros((K, K), dtype=np.float64)
    lib().orc_xtx_accumulate_f64(_p(x), ctypes.c_int64(n), ctypes.c_int(K), ctypes.c_int64(K), _p(G))
    return G


def hessian_from_gram(G: np.ndarray, n_samples: int) -> np.ndarray:
    """H = (2/n) * sum_b X_b^T X_b, which is what the running update converges to."""
    return (G * (2.0 / n_samples)).astype(np.float32)


def hessian_from_gram_f32(G32: np.ndarray, n_samples: int) -> np.ndarray:
    """Same with the Gram sum already rounded to fp32 (the HIP path's stage boundary):
    one fp32 multiply by fp32(2/n) per element."""
    return np.asarray(G32, np.float32) * np.float32(2.0 / n_samples)


# ---------------------------------------------------------------------------------------------
# a8  dead columns, damping, Cholesky inverse, upper Cholesky          [SURVEY A.2]
# ---------------------------------------------------------------------------------------------
def hessian_dead_and_damp(H: np.ndarray, percdamp: float = 0.01):
    """Returns (H_damped fp32 copy, dead mask).  ``dead = diag==0; H[dead,dead]=1;
    damp = percdamp*mean(diag); H[diag] += damp``.  The mean is carried in fp64 and rounded to
    fp32 (torch's fp32 mean uses an unspecified pairwise order)."""
    H = np.array(H, dtype=np.float32, copy=True)
    d = np.diag(H).copy()
    dead = d == 0
    d[dead] = 1.0
    # torch: python scalar * 0-dim fp32 tensor -> one fp32 multiply of fp32(percdamp) by the mean
    damp = np.float32(percdamp) * np.float32(d.astype(np.float64).mean())
    d = d + damp
    H[np.diag_indices_from(H)] = d
    return H, dead, damp


def cholesky_inverse_upper_lapack(Hd: np.ndarray):
    """Upstream's exact op sequence through the same LAPACK routines torch-CPU calls, in fp32:
    ``L = cholesky(H); Hinv = cholesky_inverse(L); U = cholesky(Hinv, upper=True)``.
    Returns (U, ok).  ``ok=False`` mirrors torch's ``LinAlgError`` -> caller uses U = I."""
    from scipy.linalg import lapack

    A = np.array(Hd, dtype=np.float32, order="F", copy=True)
    c, info = lapack.spotrf(A, lower=1, clean=1, overwrite_a=1)
    if info != 0:
        return np.eye(Hd.shape[0], dtype=np.float32), False
    inv, info = lapack.spotri(c, lower=1, overwrite_c=1)
    if info != 0:
        return np.eye(Hd.shape[0], dtype=np.float32), False
    inv = np.tril(inv) + np.tril(inv, -1).T  # potri fills one triangle only
    u, info = lapack.spotrf(np.asfortranarray(inv), lower=0, clean=1)
    if info != 0:
        return np.eye(Hd.shape[0], dtype=np.float32), False
    return np.ascontiguousarray(u, dtype=np.float32), True


def cholesky_inverse_upper_f64(Hd: np.ndarray) -> np.ndarray:
    """fp64 'truth' for U = chol(H^-1, upper): used to size tolerances, not as a target."""
    A = np.asarray(Hd, dtype=np.float64)
    L = np.linalg.cholesky(A)
    Linv = np.linalg.inv(L)
    Hinv = Linv.T @ Linv
    return np.linalg.cholesky(Hinv).T.copy()


def cholesky_inverse_upper_f64_lapack(Hd: np.ndarray) -> np.ndarray:
    """The same fp64 'truth' through dpotrf / dpotri / dpotrf (4/3 K^3 flop instead of the dense
    inverse and product of ``cholesky_inverse_upper_f64``): what the full-size factor tests use
    at K = 8192 / 14336, where the numpy form would take minutes."""
    from scipy.linalg import lapack

    A = np.array(Hd, dtype=np.float64, order="F", copy=True)
    c, info = lapack.dpotrf(A, lower=1, clean=1, overwrite_a=1)
    assert info == 0, info
    inv, info = lapack.dpotri(c, lower=1, overwrite_c=1)
    assert info == 0, info
    # potri fills the lower triangle only; the upper factorisation below reads the upper one, which
    # in Fortran order is the transpose view of the lower triangle of the same buffer
    u, info = lapack.dpotrf(np.asfortranarray(inv.T), lower=0, clean=1, overwrite_a=1)
    assert info == 0, info
    return np.ascontiguousarray(u)


def cholesky_inverse_upper_ul(Hd: np.ndarray, dtype=np.float64) -> np.ndarray:
    """The algebraic shortcut the HIP path uses, restated on the CPU: with A = flip(H),
    A = R^T R (R upper), U = flip(R^-T).  Identical to the three-step sequence in exact
    arithmetic (uniqueness of the Cholesky factor of H^-1)."""
    A = np.asarray(Hd, dtype=dtype)[::-1, ::-1]
    R = np.linalg.cholesky(A.astype(np.float64)).T
    Y = np.linalg.inv(R).T  # R^-T, lower
    return np.ascontiguousarray(Y[::-1, ::-1]).astype(dtype)


# ---------------------------------------------------------------------------------------------
# a10  minmax observer -> calculate_qparams                              [SURVEY A.2]
# ---------------------------------------------------------------------------------------------
def calculate_range(num_bits: int):
    return float(-(2 ** (num_bits - 1))), float(2 ** (num_bits - 1) - 1)


def minmax_qparams(W: np.ndarray, group_size: int, symmetric: bool = True, num_bits: int = 4):
    """Group strategy.  ``group_size <= 0`` means channel-wise (one group per row)."""
    W = np.asarray(W, dtype=np.float32)
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    assert K % gs == 0
    qmin, qmax = calculate_range(num_bits)
    Wg = W.reshape(R, K // gs, gs)
    mn = np.minimum(Wg.min(axis=2), np.float32(0))
    mx = np.maximum(Wg.max(axis=2), np.float32(0))
    if symmetric:
        amax = np.maximum(np.abs(mn), np.abs(mx))
        scale = amax / np.float32((qmax - qmin) / 2.0)
        scale = np.maximum(scale, F32_EPS).astype(np.float32)
        zp = np.zeros_like(scale)
    else:
        scale = (mx - mn) / np.float32(qmax - qmin)
        scale = np.maximum(scale, F32_EPS).astype(np.float32)
        zp = np.float32(qmin) - mn / scale
        zp = np.clip(np.rint(zp), qmin, qmax).astype(np.float32)
    return scale, zp


def minmax_qparams_c(W, group_size, symmetric=True, num_bits=4):
    W = np.ascontiguousarray(W, dtype=np.float32)
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    qmin, qmax = calculate_range(num_bits)
    scale = np.empty((R, K // gs), np.float32)
    zp = np.empty((R, K // gs), np.float32)
    lib().orc_minmax_qparams(_p(W), R, K, gs, int(symmetric), ctypes.c_float(qmin), ctypes.c_float(qmax),
                             _p(scale), _p(zp))
    return scale, zp


def fake_quantize(x, scale, zp, num_bits=4):
    """``round_half_even(clamp(x/s + zp, qmin, qmax))`` -> (q_int, (q - zp) * s), all fp32."""
    qmin, qmax = calculate_range(num_bits)
    x = np.asarray(x, np.float32)
    t = x / np.asarray(scale, np.float32)
    t = t + np.asarray(zp, np.float32)
    t = np.clip(t, np.float32(qmin), np.float32(qmax))
    q = np.rint(t).astype(np.float32)
    return q, (q - zp) * scale


# ---------------------------------------------------------------------------------------------
# a11  column sweep                                                         [SURVEY A.2]
# ---------------------------------------------------------------------------------------------
def gptq_sweep_c(W, U, scale, zp, g_idx, blocksize=128, num_bits=4):
    """C restatement (the arithmetic of record).  Returns (Q int8, W_dq fp32, loss[R])."""
    W = np.array(W, dtype=np.float32, order="C", copy=True)
    U = np.ascontiguousarray(U, dtype=np.float32)
    scale = np.ascontiguousarray(scale, np.float32)
    zp = np.ascontiguousarray(zp, np.float32)
    g_idx = np.ascontiguousarray(g_idx, np.int32)
    R, K = W.shape
    G = scale.shape[1]
    qmin, qmax = calculate_range(num_bits)
    Q = np.empty((R, K), np.int8)
    loss = np.empty((R,), np.float32)
    lib().orc_gptq_sweep(_p(W), R, K, _p(U), _p(scale), _p(zp), G, _p(g_idx), blocksize,
                         ctypes.c_float(qmin), ctypes.c_float(qmax), _p(Q), _p(loss))
    return Q, W, loss


def gptq_sweep_numpy(W, U, scale, zp, g_idx, blocksize=128, num_bits=4):
    """Independent numpy restatement (vectorised over rows, same per-element op sequence).
    The trailing update's fmaf chain is emulated through float64 (exact product, one extra
    rounding in 2^-29 of cases) -- for small cross-checks of the C code only."""
    W = np.array(W, dtype=np.float32, copy=True)
    R, K = W.shape
    Q = np.zeros((R, K), np.int8)
    loss = np.zeros((R,), np.float32)
    rows = np.arange(R)
    for i1 in range(0, K, blocksize):
        i2 = min(i1 + blocksize, K)
        err_blk = np.zeros((R, i2 - i1), np.float32)
        blk_loss = np.zeros((R,), np.float32)
        for i in range(i2 - i1):
            c = i1 + i
            d = U[c, c]
            g = g_idx[c]
            w = W[:, c].copy()
            qi, q = fake_quantize(w, scale[rows, g], zp[rows, g], num_bits)
            Q[:, c] = qi.astype(np.int8)
            diff = w - q
            blk_loss = blk_loss + (diff * diff) / (d * d)
            e = diff / d
            err_blk[:, i] = e
            W[:, c] = q
            if c + 1 < i2:
                W[:, c + 1:i2] = W[:, c + 1:i2] - e[:, None] * U[c, c + 1:i2][None, :]
        loss = loss + blk_loss / np.float32(2.0)
        if i2 < K:
            P = np.zeros((R, K - i2), np.float32)
            for i in range(i2 - i1):
                P = (err_blk[:, i].astype(np.float64)[:, None] * U[i1 + i, i2:].astype(np.float64)[None, :]
                     + P.astype(np.float64)).astype(np.float32)
            W[:, i2:] = W[:, i2:] - P
    return Q, W, loss


# ---------------------------------------------------------------------------------------------
# a9 + a8 + a10 + a11 glued: quantize_weight                                [SURVEY A.2]
# ---------------------------------------------------------------------------------------------
def quantize_weight(W, H, *, group_size=128, symmetric=True, num_bits=4, blocksize=128,
                    percdamp=0.01, actorder="static", inverse="lapack", U_override=None,
                    sweep=gptq_sweep_c, perm_override=None):
    """Full per-Linear GPTQ as upstream's ``quantize_weight``.  ``actorder`` in
    {None, "static"/"weight", "group"}.  Returns dict with q (int8, original column order),
    scale, zp (fp32 [R,G]), g_idx (int32 [K] or None), w_dq, loss, perm, U, ok.
    ``perm_override`` (tests only): sweep in this order instead of ``argsort(diag H)`` -- the argsort of nearly
    equal diagonal entries flips with the last bit of H, i.e. with the order a Hessian was summed in; fixing the
    order separates that from everything else (tests/own_hessian_cases.py)."""
    W = np.array(W, dtype=np.float32, copy=True)
    H = np.array(H, dtype=np.float32, copy=True)
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    g_idx = (np.arange(K) // gs).astype(np.int32)
    perm = None
    if actorder in ("static", "weight"):
        scale, zp = minmax_qparams(W, group_size, symmetric, num_bits)
        perm = np.argsort(-np.diag(H), kind="stable") if perm_override is None else np.asarray(perm_override, np.int64)
        W = W[:, perm]
        H = H[perm][:, perm]
        g_idx = g_idx[perm]
    elif actorder == "group":
        perm = np.argsort(-np.diag(H), kind="stable") if perm_override is None else np.asarray(perm_override, np.int64)
        W = W[:, perm]
        H = H[perm][:, perm]
        scale, zp = minmax_qparams(W, group_size, symmetric, num_bits)
    else:
        scale, zp = minmax_qparams(W, group_size, symmetric, num_bits)
    Hd, dead, _ = hessian_dead_and_damp(H, percdamp)
    W[:, dead] = 0
    if U_override is not None:
        U, ok = np.asarray(U_override, np.float32), True
    elif inverse == "lapack":
        U, ok = cholesky_inverse_upper_lapack(Hd)
    else:
        U, ok = cholesky_inverse_upper_f64(Hd).astype(np.float32), True
    Q, Wdq, loss = sweep(W, U, scale, zp, g_idx, blocksize, num_bits)
    out_g_idx = None
    if perm is not None:
        inv = np.argsort(perm, kind="stable")
        Q = Q[:, inv]
        Wdq = Wdq[:, inv]
        if actorder == "group":
            out_g_idx = g_idx[inv].astype(np.int32)
    return dict(q=np.ascontiguousarray(Q), scale=scale, zp=zp, g_idx=out_g_idx,
                w_dq=np.ascontiguousarray(Wdq), loss=loss, perm=perm, U=U, ok=ok, dead=dead)


# ---------------------------------------------------------------------------------------------
# a14  pack_to_int32 / unpack                                                [SURVEY A.5]
# ---------------------------------------------------------------------------------------------
def pack_int4(Q: np.ndarray) -> np.ndarray:
    Q = np.asarray(Q, np.int8)
    R, K = Q.shape
    Kp = (K + 7) // 8 * 8
    u = np.zeros((R, Kp), np.uint32)
    u[:, :K] = (Q.astype(np.int32) + 8).astype(np.uint32) & 0xF
    u = u.reshape(R, Kp // 8, 8)
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    return (u << shifts).sum(axis=2, dtype=np.uint32).view(np.int32)


def pack_int4_c(Q):
    Q = np.ascontiguousarray(Q, np.int8)
    R, K = Q.shape
    out = np.empty((R, (K + 7) // 8), np.int32)
    lib().orc_pack_int4(_p(Q), R, K, _p(out))
    return out


def unpack_int4(packed: np.ndarray, K: int) -> np.ndarray:
    p = packed.view(np.uint32)
    R = p.shape[0]
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    u = (p[:, :, None] >> shifts) & 0xF
    return (u.reshape(R, -1)[:, :K].astype(np.int32) - 8).astype(np.int8)


def requantize_at_save(w_dq_bf16_bits, scale_bf16_bits, zp, g_idx_cols, num_bits=4):
    """Upstream's save-time step (A.5): q = clamp(round(W_dq(bf16)/scale(bf16) + zp)).
    Used to show it reproduces the sweep's integer levels ('pack-time requant')."""
    w = bf16_bits_to_f32(w_dq_bf16_bits)
    s = bf16_bits_to_f32(scale_bf16_bits)
    rows = np.arange(w.shape[0])[:, None]
    sc = s[rows, g_idx_cols[None, :]]
    z = np.asarray(zp, np.float32)[rows, g_idx_cols[None, :]]
    q, _ = fake_quantize(w, sc, z, num_bits)
    return q.astype(np.int8)


# ---------------------------------------------------------------------------------------------
# a12  AWQ                                                                    [SURVEY A.3]
# ---------------------------------------------------------------------------------------------
def awq_pseudo_quantize(W, group_size=128, symmetric=True, num_bits=4):
    """``_pseudo_quantize_tensor``: note /(2^(b-1)-1) = /7 for int4, unlike the observer's /7.5."""
    W = np.asarray(W, np.float32)
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    Wg = W.reshape(-1, gs)
    if symmetric:
        amax = np.maximum(np.abs(Wg).max(axis=1, keepdims=True), np.float32(1e-5))
        max_int = np.float32(2 ** (num_bits - 1) - 1)
        min_int = np.float32(-(2 ** (num_bits - 1)))
        sc = amax / max_int
        out = np.clip(np.rint(Wg / sc), min_int, max_int) * sc
    else:
        mx = Wg.max(axis=1, keepdims=True)
        mn = Wg.min(axis=1, keepdims=True)
        max_int = np.float32(2 ** num_bits - 1)
        sc = np.maximum(mx - mn, np.float32(1e-5)) / max_int
        z = np.clip(-np.rint(mn / sc), 0, max_int)
        out = (np.clip(np.rint(Wg / sc) + z, 0, max_int) - z) * sc
    return out.reshape(R, K).astype(np.float32)


def awq_weight_mean(W_list, group_size=128):
    """``w_mean``: mean over rows (all balance layers concatenated) of |W| / (group amax + 1e-6)."""
    W = np.concatenate([np.asarray(w, np.float32) for w in W_list], axis=0)
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    Wg = np.abs(W).reshape(-1, gs)
    Wn = Wg / (Wg.max(axis=1, keepdims=True) + np.float32(1e-6))
    return Wn.reshape(R, K).astype(np.float64).mean(axis=0).astype(np.float32)


def awq_scales_for_ratio(x_mean, w_mean, ratio, duo_scaling=True):
    x_mean = np.asarray(x_mean, np.float32)
    w_mean = np.asarray(w_mean, np.float32)
    r = np.float32(ratio)
    if duo_scaling:
        s = np.power(x_mean, r) / (np.power(w_mean, np.float32(1) - r) + np.float32(1e-4))
    else:
        s = np.power(x_mean, r)
    s = np.maximum(s, np.float32(1e-4)).astype(np.float32)
    s = s / np.sqrt(s.max() * s.min())
    s[~np.isfinite(s)] = 1.0
    return s.astype(np.float32)


def awq_best_scale(X_bf16, W_list, group_size=128, symmetric=True, num_bits=4, n_grid=20,
                   duo_scaling=True):
    """``_compute_best_scale`` for a mapping whose parent module is the balance Linear(s)
    themselves (single-consumer mappings).  Loss in float64."""
    X = bf16_bits_to_f32(X_bf16)
    x_mean = np.abs(X).astype(np.float64).mean(axis=0).astype(np.float32)
    w_mean = awq_weight_mean(W_list, group_size)
    W = np.concatenate([np.asarray(w, np.float32) for w in W_list], axis=0)
    Y = X.astype(np.float64) @ W.T.astype(np.float64)
    losses = []
    best = (np.inf, -1, None)
    for gi in range(n_grid):
        ratio = gi / n_grid
        s = awq_scales_for_ratio(x_mean, w_mean, ratio, duo_scaling)
        Wq = awq_pseudo_quantize(W * s[None, :], group_size, symmetric, num_bits) / s[None, :]
        Yq = X.astype(np.float64) @ Wq.T.astype(np.float64)
        loss = float(np.mean((Y - Yq) ** 2))
        losses.append(loss)
        if loss < best[0]:
            best = (loss, gi, s)
    return dict(best_ratio_idx=best[1], best_scales=best[2], losses=np.array(losses),
                x_mean=x_mean, w_mean=w_mean)


# ---------------------------------------------------------------------------------------------
# a13  SmoothQuant                                                             [SURVEY A.4]
# ---------------------------------------------------------------------------------------------
def smoothquant_scales(act_min, act_max, W_list, alpha=0.5):
    a = (np.asarray(act_max, np.float32) - np.asarray(act_min, np.float32))
    w = np.max(np.stack([np.abs(np.asarray(wi, np.float32)).max(axis=0) for wi in W_list]), axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.power(a, np.float32(alpha)) / np.power(w, np.float32(1.0 - alpha))
    s = np.where(w > 0, s, a).astype(np.float32)
    return s


def channel_minmax(X_bf16):
    X = bf16_bits_to_f32(X_bf16)
    return X.min(axis=0), X.max(axis=0)


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def num_threads() -> int:
    return int(lib().orc_num_threads())
