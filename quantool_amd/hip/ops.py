"""Thin torch-tensor wrappers over the C ABI (``include/quantool_amd.h``).

torch is plumbing here: device memory, the current HIP stream and shape checks before a raw
pointer crosses the boundary (a wrong shape in a hand-written kernel is a GPU fault, so every
assumption the kernels make is asserted on the host first).  No arithmetic happens in torch.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch

from . import _lib
from ._lib import QT_BF16, QT_F16, QT_F32, check, load

_WS_CACHE: dict = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return QT_F32
    if t.dtype == torch.bfloat16:
        return QT_BF16
    if t.dtype == torch.float16:
        return QT_F16
    raise TypeError(f"unsupported dtype {t.dtype} (fp32 / bf16 / fp16 only)")


def _act16(X: torch.Tensor, name: str) -> int:
    """Activations cross the boundary in the model's own 16-bit dtype."""
    if not X.is_cuda:
        raise ValueError(f"{name} must be a device tensor (no CPU path exists)")
    if X.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"{name} must be bf16 or fp16, got {X.dtype}")
    return QT_BF16 if X.dtype == torch.bfloat16 else QT_F16


_FP32_ACT_WARNED = False


def as_act16(X: torch.Tensor) -> torch.Tensor:
    """Activations as the kernels take them: the model's own 16-bit dtype, untouched.  Upstream
    accumulates ``inp.float()``; bf16 and fp16 widen exactly, so nothing is lost.

    fp32 activations (an fp32 checkpoint): the Gram accumulation has its own fp32-accurate path
    (``xtx_accumulate_f32`` / ``HessianAccumulator``) and does not come here.  The STATISTICS passes (AWQ's mean |x|,
    SmoothQuant's channel min / max) take 16-bit rows, so for them -- and for the Gram pass under
    ``QT_FP32_ACTIVATIONS=bf16`` -- fp32 activations are rounded to bf16 (range over mantissa: an fp16 cast could
    overflow on outlier channels).  That downgrade is never silent: it is logged once per process, and with
    ``QT_FP32_ACTIVATIONS=error`` it is refused (``ValueError``)."""
    if X.dtype in (torch.bfloat16, torch.float16):
        return X
    wide_activation_policy(X.dtype)
    return X.to(torch.bfloat16)


def wide_gram_mode() -> str:
    """How the Gram accumulation takes fp32 activations: "exact" (default: the three-plane fp32-accurate product,
    6x the MFMA work), "bf16" (round them, logged once) or "error" -- ``QT_FP32_ACTIVATIONS``."""
    mode = os.environ.get("QT_FP32_ACTIVATIONS", "exact").lower()
    if mode not in ("exact", "bf16", "warn", "error"):
        raise ValueError(f"QT_FP32_ACTIVATIONS={mode!r}: expected exact, bf16 or error")
    return "bf16" if mode == "warn" else mode


def wide_activation_policy(dtype) -> None:
    """Log (once) or refuse the bf16 rounding of activations wider than 16 bits: see ``as_act16``."""
    if dtype not in (torch.float32, torch.float64):
        raise TypeError(f"calibration activations must be floating point, got {dtype}")
    global _FP32_ACT_WARNED
    policy = os.environ.get("QT_FP32_ACTIVATIONS", "exact").lower()
    if policy == "error":
        raise ValueError(f"{dtype} calibration activations: this backend accumulates X^T X from 16-bit activations; "
                         "load the model in bf16 / fp16 (oneshot(precision=...)) or unset QT_FP32_ACTIVATIONS=error to "
                         "accept rounding them to bf16")
    if not _FP32_ACT_WARNED:
        _FP32_ACT_WARNED = True
        logging.getLogger(__name__).warning(
            f"{dtype} calibration activations are rounded to bf16 for the Gram / statistics passes (upstream accumulates "
            "them in fp32): Hessians differ at the 2^-9 level per activation.  Load the model in bf16 / fp16 to calibrate "
            "in its own dtype, or set QT_FP32_ACTIVATIONS=error to refuse")


def _req(t: torch.Tensor, dtype, name: str, ndim: Optional[int] = None):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-d, got shape {tuple(t.shape)}")


def workspace(nbytes: int, device, tag: str = "default") -> torch.Tensor:
    """Grow-only scratch buffer per (device, tag); reused across calls on the same stream."""
    # one scratch buffer per (device, stream, tag): groups running on different streams never share
    key = (str(device), _stream(), tag)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = buf
    return buf


def release_workspaces() -> None:
    _WS_CACHE.clear()


# ---- a7 -----------------------------------------------------------------------------------
def xtx_accumulate(X: torch.Tensor, G: torch.Tensor) -> None:
    """G[K,K] fp32 (lower triangle) += X^T X for X[..., K] bf16 / fp16 (leading dims flattened)."""
    lib = load()
    xdt = _act16(X, "X")
    _req(G, torch.float32, "G", 2)
    K = G.shape[0]
    if G.shape[1] != K or not G.is_contiguous():
        raise ValueError("G must be contiguous [K, K]")
    if X.shape[-1] != K:
        raise ValueError(f"X last dim {X.shape[-1]} != K {K}")
    X2 = X.reshape(-1, K)
    if X2.stride(1) != 1:
        X2 = X2.contiguous()
    n = X2.shape[0]
    if n == 0:
        return
    ldx = X2.stride(0) if n > 1 else K
    nbytes = lib.qt_xtx_workspace_bytes(n, K)
    ws = workspace(nbytes, X.device, "xtx")
    check("qt_xtx_accumulate", lib.qt_xtx_accumulate(X2.data_ptr(), xdt, n, K, ldx, G.data_ptr(), ws.data_ptr(),
                                                     ws.numel(), _stream()))


def xtx_accumulate_f32(X: torch.Tensor, G: torch.Tensor) -> None:
    """G[K,K] fp32 (lower triangle) += X^T X for fp32 X[..., K]: the fp32-accurate three-plane product (an fp32
    checkpoint's activations, which upstream accumulates as `inp.float()`)."""
    lib = load()
    _req(X, torch.float32, "X")
    _req(G, torch.float32, "G", 2)
    K = G.shape[0]
    if G.shape[1] != K or not G.is_contiguous():
        raise ValueError("G must be contiguous [K, K]")
    if X.shape[-1] != K:
        raise ValueError(f"X last dim {X.shape[-1]} != K {K}")
    if K % 4:
        raise ValueError(f"fp32 activations need in_features % 4 == 0 (K = {K})")
    X2 = X.reshape(-1, K)
    if X2.stride(1) != 1 or X2.stride(0) % 4 or X2.data_ptr() % 16:
        X2 = X2.contiguous()
    n = X2.shape[0]
    if n == 0:
        return
    ldx = X2.stride(0) if n > 1 else K
    ws = workspace(lib.qt_xtx_accumulate_f32_workspace_bytes(n, K), X.device, "xtx_f32")
    check("qt_xtx_accumulate_f32", lib.qt_xtx_accumulate_f32(X2.data_ptr(), n, K, ldx, G.data_ptr(), ws.data_ptr(),
                                                             ws.numel(), _stream()))


# ---- a12 / a13 ----------------------------------------------------------------------------
def act_stats_accumulate(X: torch.Tensor, abs_sum: Optional[torch.Tensor] = None,
                         cmin: Optional[torch.Tensor] = None, cmax: Optional[torch.Tensor] = None) -> None:
    lib = load()
    xdt = _act16(X, "X")
    K = X.shape[-1]
    X2 = X.reshape(-1, K)
    if X2.stride(1) != 1:
        X2 = X2.contiguous()
    n = X2.shape[0]
    for name, t in (("abs_sum", abs_sum), ("cmin", cmin), ("cmax", cmax)):
        if t is not None:
            _req(t, torch.float32, name, 1)
            if t.numel() != K or not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous [K]")
    if n == 0:
        return
    ldx = X2.stride(0) if n > 1 else K
    ws = workspace(lib.qt_act_stats_workspace_bytes(n, K), X.device, "stats")
    check("qt_act_stats_accumulate", lib.qt_act_stats_accumulate(
        X2.data_ptr(), xdt, n, K, ldx, _ptr(abs_sum), _ptr(cmin), _ptr(cmax), ws.data_ptr(), ws.numel(), _stream()))


# ---- a8 / a9 ------------------------------------------------------------------------------
def hessian_diag(G: torch.Tensor, n_samples: int) -> torch.Tensor:
    lib = load()
    _req(G, torch.float32, "G", 2)
    K = G.shape[0]
    out = torch.empty(K, dtype=torch.float32, device=G.device)
    check("qt_hessian_diag", lib.qt_hessian_diag(G.data_ptr(), K, n_samples, out.data_ptr(), _stream()))
    return out


def argsort_desc(values: torch.Tensor):
    """Stable descending argsort on the device: returns (perm int32[K], inv int32[K])."""
    lib = load()
    _req(values, torch.float32, "values", 1)
    K = values.numel()
    assert values.is_contiguous()
    perm = torch.empty(K, dtype=torch.int32, device=values.device)
    inv = torch.empty(K, dtype=torch.int32, device=values.device)
    check("qt_argsort_desc", lib.qt_argsort_desc(values.data_ptr(), K, perm.data_ptr(), inv.data_ptr(), _stream()))
    return perm, inv


def factor_buffer(K: int, device) -> torch.Tensor:
    """Flat fp32 buffer that holds the factor U ([K, K], its first K*K elements: ``factor_view``) and, before U is
    written, serves ``hessian_prepare`` as scratch (``scratch=``): the two-pass prepare needs a K x K symmetric copy
    of G -- 0.8 GB at K = 14336, 3.3 GB at K = 28672 -- which would otherwise sit in a grow-only per-stream workspace
    for the life of the process next to the factor it precedes."""
    return torch.empty(K * K + 256, dtype=torch.float32, device=device)


def factor_view(buf: torch.Tensor, K: int) -> torch.Tensor:
    return buf[:K * K].view(K, K)


def hessian_prepare(G: torch.Tensor, n_samples: int, percdamp: float, perm: Optional[torch.Tensor] = None,
                    A_out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None):
    """Returns (A flipped+damped [K,K] fp32 upper-valid, dead uint8[K], diag fp32[K]).  ``scratch``: a contiguous
    device tensor whose bytes may be used as the workspace (e.g. ``factor_buffer``: the buffer the factor will be
    written to afterwards); too small or absent: the cached per-stream workspace."""
    lib = load()
    _req(G, torch.float32, "G", 2)
    K = G.shape[0]
    if not G.is_contiguous() or G.shape[1] != K:
        raise ValueError("G must be contiguous [K, K]")
    if perm is not None:
        _req(perm, torch.int32, "perm", 1)
        if perm.numel() != K or not perm.is_contiguous():
            raise ValueError("perm must be contiguous int32[K]")
    A = A_out if A_out is not None else torch.empty((K, K), dtype=torch.float32, device=G.device)
    _req(A, torch.float32, "A", 2)
    if A.shape != (K, K) or not A.is_contiguous():
        raise ValueError("A must be contiguous [K, K]")
    dead = torch.empty(K, dtype=torch.uint8, device=G.device)
    diag = torch.empty(K, dtype=torch.float32, device=G.device)
    need = lib.qt_hessian_prepare_workspace_bytes(K)
    if scratch is not None and scratch.is_cuda and scratch.is_contiguous() and scratch.numel() * scratch.element_size() >= need:
        ws = scratch.view(torch.uint8).reshape(-1)
    else:
        ws = workspace(need, G.device, "prep")
    check("qt_hessian_prepare", lib.qt_hessian_prepare(G.data_ptr(), K, int(n_samples), float(percdamp), _ptr(perm),
                                                       A.data_ptr(), dead.data_ptr(), diag.data_ptr(), ws.data_ptr(),
                                                       ws.numel(), _stream()))
    return A, dead, diag


def cholesky_inverse_upper(A: torch.Tensor, U_out: Optional[torch.Tensor] = None):
    """A (flipped damped Hessian, destroyed) -> (U upper [K,K] fp32, info int32[1] device)."""
    lib = load()
    _req(A, torch.float32, "A", 2)
    K = A.shape[0]
    if A.shape != (K, K) or not A.is_contiguous():
        raise ValueError("A must be contiguous [K, K]")
    U = U_out if U_out is not None else torch.empty((K, K), dtype=torch.float32, device=A.device)
    if U.shape != (K, K) or not U.is_contiguous() or U.dtype != torch.float32:
        raise ValueError("U must be contiguous fp32 [K, K]")
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    ws = workspace(lib.qt_cholesky_inverse_upper_workspace_bytes(K), A.device, "chol")
    check("qt_cholesky_inverse_upper", lib.qt_cholesky_inverse_upper(A.data_ptr(), K, U.data_ptr(), info.data_ptr(),
                                                                     ws.data_ptr(), ws.numel(), _stream()))
    return U, info


MAX_BATCH = 16      # problems per batched factorisation / groups per grouped sweep (SG_MAX_BATCH / SG_MAX_GROUPS)


def cholesky_inverse_upper_batched(A: torch.Tensor, U: torch.Tensor):
    """n problems of one size in every launch of the chain.  ``A``: [n, K, K] fp32 (flipped damped Hessians, destroyed)
    -- or any tensor whose [b] slices are contiguous K x K matrices at a uniform stride -- ``U`` likewise (written).
    Returns info int32[n] (device).  Bit-identical per problem to ``cholesky_inverse_upper``."""
    lib = load()
    _req(A, torch.float32, "A", 3)
    _req(U, torch.float32, "U", 3)
    n, K, K2 = A.shape
    if K != K2 or tuple(U.shape) != (n, K, K) or not (1 <= n <= MAX_BATCH):
        raise ValueError(f"A / U must be [n <= {MAX_BATCH}, K, K], got {tuple(A.shape)} / {tuple(U.shape)}")
    for name, t in (("A", A), ("U", U)):
        if t.stride(2) != 1 or t.stride(1) != K or (n > 1 and (t.stride(0) < K * K or t.stride(0) % 4)):
            raise ValueError(f"{name}: every [b] slice must be a contiguous K x K matrix at a stride that is a multiple of 4")
    info = torch.zeros(n, dtype=torch.int32, device=A.device)
    ws = workspace(lib.qt_cholesky_inverse_upper_batched_workspace_bytes(K, n), A.device, "chol")
    check("qt_cholesky_inverse_upper_batched", lib.qt_cholesky_inverse_upper_batched(
        A.data_ptr(), A.stride(0) if n > 1 else K * K, K, U.data_ptr(), U.stride(0) if n > 1 else K * K, info.data_ptr(), n,
        ws.data_ptr(), ws.numel(), _stream()))
    return info


# ---- a10 ----------------------------------------------------------------------------------
def group_minmax_qparams(W: torch.Tensor, group_size: int, symmetric: bool = True, num_bits: int = 4):
    """Returns (scale[R,G], zp[R,G], scale_t[G,R], zp_t[G,R]) fp32."""
    lib = load()
    if W.dim() != 2 or not W.is_cuda or W.stride(1) != 1:
        raise ValueError("W must be a 2-d device tensor with unit column stride")
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    if K % gs:
        raise ValueError(f"K={K} not divisible by group_size={gs}")
    G = K // gs
    dev = W.device
    scale = torch.empty((R, G), dtype=torch.float32, device=dev)
    zp = torch.empty((R, G), dtype=torch.float32, device=dev)
    scale_t = torch.empty((G, R), dtype=torch.float32, device=dev)
    zp_t = torch.empty((G, R), dtype=torch.float32, device=dev)
    check("qt_group_minmax_qparams", lib.qt_group_minmax_qparams(
        W.data_ptr(), _dtype_code(W), R, K, W.stride(0), gs, int(bool(symmetric)), num_bits, scale.data_ptr(),
        zp.data_ptr(), scale_t.data_ptr(), zp_t.data_ptr(), _stream()))
    return scale, zp, scale_t, zp_t


def weight_gather_f32(W: torch.Tensor, perm: Optional[torch.Tensor] = None, dead: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load()
    if W.dim() != 2 or not W.is_cuda or W.stride(1) != 1:
        raise ValueError("W must be a 2-d device tensor with unit column stride")
    R, K = W.shape
    if perm is not None:
        _req(perm, torch.int32, "perm", 1)
        assert perm.numel() == K and perm.is_contiguous()
    if dead is not None:
        _req(dead, torch.uint8, "dead", 1)
        assert dead.numel() == K and dead.is_contiguous()
    if out is None:
        out = torch.empty((R, K), dtype=torch.float32, device=W.device)
    assert out.shape == (R, K) and out.is_contiguous() and out.dtype == torch.float32
    check("qt_weight_gather_f32", lib.qt_weight_gather_f32(W.data_ptr(), _dtype_code(W), R, K, W.stride(0),
                                                           _ptr(perm), _ptr(dead), out.data_ptr(), _stream()))
    return out


def weight_gather_qparams(W: torch.Tensor, perm: Optional[torch.Tensor], dead: Optional[torch.Tensor], group_size: int,
                          symmetric: bool, num_bits: int, out: torch.Tensor, scale: torch.Tensor, zp: torch.Tensor,
                          scale_t: Optional[torch.Tensor] = None, zp_t: Optional[torch.Tensor] = None) -> None:
    """``group_minmax_qparams(W)`` and ``weight_gather_f32(W, perm, dead)`` in one read of W.  ``out`` fp32 [R, K],
    ``scale`` / ``zp`` fp32 [R, G] contiguous; ``scale_t`` / ``zp_t``: [G, R] views with unit column stride (a column
    range of a table spanning more rows is fine: the row stride is passed on)."""
    lib = load()
    if W.dim() != 2 or not W.is_cuda or W.stride(1) != 1:
        raise ValueError("W must be a 2-d device tensor with unit column stride")
    R, K = W.shape
    gs = K if group_size <= 0 else group_size
    if K % gs:
        raise ValueError(f"K={K} not divisible by group_size={gs}")
    G = K // gs
    for name, t, shape in (("out", out, (R, K)), ("scale", scale, (R, G)), ("zp", zp, (R, G))):
        _req(t, torch.float32, name, 2)
        if tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous {shape}")
    ld_t = 0
    for name, t in (("scale_t", scale_t), ("zp_t", zp_t)):
        if t is not None:
            _req(t, torch.float32, name, 2)
            if tuple(t.shape) != (G, R) or t.stride(1) != 1 or (ld_t and t.stride(0) != ld_t):
                raise ValueError(f"{name} must be a [G, R] view with unit column stride (and the row stride of its twin)")
            ld_t = t.stride(0) if G > 1 else max(R, t.stride(0))
    for name, t, dt in (("perm", perm, torch.int32), ("dead", dead, torch.uint8)):
        if t is not None:
            _req(t, dt, name, 1)
            if t.numel() != K or not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous [K]")
    check("qt_weight_gather_qparams", lib.qt_weight_gather_qparams(
        W.data_ptr(), _dtype_code(W), R, K, W.stride(0), _ptr(perm), _ptr(dead), group_size, int(bool(symmetric)), num_bits,
        out.data_ptr(), scale.data_ptr(), zp.data_ptr(), _ptr(scale_t), _ptr(zp_t), ld_t, _stream()))


# ---- a11 ----------------------------------------------------------------------------------
def gptq_sweep(W: torch.Tensor, U: torch.Tensor, scale_t: torch.Tensor, zp_t: torch.Tensor, g_idx: torch.Tensor,
               blocksize: int = 128, num_bits: int = 4):
    """In-place on W (fp32 [R,K], sweep order).  Returns (Qt int8 [K,R], loss fp32 [R])."""
    lib = load()
    _req(W, torch.float32, "W", 2)
    _req(U, torch.float32, "U", 2)
    _req(scale_t, torch.float32, "scale_t", 2)
    _req(zp_t, torch.float32, "zp_t", 2)
    _req(g_idx, torch.int32, "g_idx", 1)
    R, K = W.shape
    G = scale_t.shape[0]
    if not (W.is_contiguous() and U.is_contiguous() and scale_t.is_contiguous() and zp_t.is_contiguous()
            and g_idx.is_contiguous()):
        raise ValueError("all sweep operands must be contiguous")
    if U.shape != (K, K) or scale_t.shape != (G, R) or zp_t.shape != (G, R) or g_idx.numel() != K:
        raise ValueError("sweep operand shapes inconsistent")
    Qt = torch.empty((K, R), dtype=torch.int8, device=W.device)
    loss = torch.empty(R, dtype=torch.float32, device=W.device)
    ws = workspace(lib.qt_gptq_sweep_workspace_bytes(R, K, blocksize), W.device, "sweep")
    check("qt_gptq_sweep", lib.qt_gptq_sweep(W.data_ptr(), R, K, U.data_ptr(), scale_t.data_ptr(), zp_t.data_ptr(), G,
                                             g_idx.data_ptr(), blocksize, num_bits, Qt.data_ptr(), loss.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _stream()))
    return Qt, loss


def gptq_sweep_grouped(W: torch.Tensor, U: torch.Tensor, row_end, scale_t: torch.Tensor, zp_t: torch.Tensor,
                       g_idx: torch.Tensor, blocksize: int = 128, num_bits: int = 4):
    """Several Linear groups of one in_features as ONE stacked sweep.  W fp32 [R_total, K] (in place), U [n, K, K] (its
    [g] slices contiguous K x K at a uniform stride), ``row_end``: list of n cumulative row counts (every boundary but
    the last a multiple of 128), g_idx int32 [n, K].  Returns (Qt int8 [K, R_total], loss fp32 [R_total])."""
    import ctypes

    lib = load()
    _req(W, torch.float32, "W", 2)
    _req(U, torch.float32, "U", 3)
    _req(scale_t, torch.float32, "scale_t", 2)
    _req(zp_t, torch.float32, "zp_t", 2)
    _req(g_idx, torch.int32, "g_idx", 2)
    R, K = W.shape
    n = U.shape[0]
    G = scale_t.shape[0]
    row_end = [int(r) for r in row_end]
    if not (W.is_contiguous() and scale_t.is_contiguous() and zp_t.is_contiguous() and g_idx.is_contiguous()):
        raise ValueError("all sweep operands must be contiguous")
    if (tuple(U.shape) != (n, K, K) or U.stride(2) != 1 or U.stride(1) != K or scale_t.shape != (G, R) or zp_t.shape != (G, R)
            or tuple(g_idx.shape) != (n, K) or len(row_end) != n or not (1 <= n <= MAX_BATCH)):
        raise ValueError("grouped sweep operand shapes inconsistent")
    if row_end[-1] != R or any(b <= a for a, b in zip([0] + row_end[:-1], row_end)) or any(r % 128 for r in row_end[:-1]):
        raise ValueError(f"row_end={row_end}: ascending, every boundary but the last a multiple of 128, last == R ({R})")
    Qt = torch.empty((K, R), dtype=torch.int8, device=W.device)
    loss = torch.empty(R, dtype=torch.float32, device=W.device)
    ws = workspace(lib.qt_gptq_sweep_workspace_bytes(R, K, blocksize), W.device, "sweep")
    ends = (ctypes.c_int32 * n)(*row_end)
    check("qt_gptq_sweep_grouped", lib.qt_gptq_sweep_grouped(
        W.data_ptr(), R, K, U.data_ptr(), U.stride(0) if n > 1 else K * K, n, ctypes.cast(ends, ctypes.c_void_p),
        scale_t.data_ptr(), zp_t.data_ptr(), G, g_idx.data_ptr(), blocksize, num_bits, Qt.data_ptr(), loss.data_ptr(),
        ws.data_ptr(), ws.numel(), _stream()))
    return Qt, loss


# ---- a14 ----------------------------------------------------------------------------------
def pack_int4(Qt: torch.Tensor, col_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load()
    _req(Qt, torch.int8, "Qt", 2)
    K, R = Qt.shape
    assert Qt.is_contiguous()
    if col_src is not None:
        _req(col_src, torch.int32, "col_src", 1)
        assert col_src.numel() == K and col_src.is_contiguous()
    packed = torch.empty((R, (K + 7) // 8), dtype=torch.int32, device=Qt.device)
    check("qt_pack_int4", lib.qt_pack_int4(Qt.data_ptr(), R, K, _ptr(col_src), packed.data_ptr(), _stream()))
    return packed


def dequantize(Qt: torch.Tensor, scale: torch.Tensor, zp: torch.Tensor, g_of_col: torch.Tensor,
               col_src: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    lib = load()
    _req(Qt, torch.int8, "Qt", 2)
    K, R = Qt.shape
    _req(scale, torch.float32, "scale", 2)
    _req(zp, torch.float32, "zp", 2)
    _req(g_of_col, torch.int32, "g_of_col", 1)
    G = scale.shape[1]
    assert scale.shape == (R, G) and zp.shape == (R, G) and g_of_col.numel() == K
    assert Qt.is_contiguous() and scale.is_contiguous() and zp.is_contiguous() and g_of_col.is_contiguous()
    if col_src is not None:
        _req(col_src, torch.int32, "col_src", 1)
        assert col_src.numel() == K and col_src.is_contiguous()
    out = torch.empty((R, K), dtype=dtype, device=Qt.device)
    check("qt_dequantize", lib.qt_dequantize(Qt.data_ptr(), R, K, _ptr(col_src), scale.data_ptr(), zp.data_ptr(), G,
                                             g_of_col.data_ptr(), out.data_ptr(), _dtype_code(out), out.stride(0),
                                             _stream()))
    return out


# ---- a12  AWQ ------------------------------------------------------------------------------
def _w2d(W: torch.Tensor):
    if W.dim() != 2 or not W.is_cuda or W.stride(1) != 1:
        raise ValueError("W must be a 2-d device tensor with unit column stride")
    return W.shape


def awq_weight_mean_accumulate(W: torch.Tensor, group_size: int, w_sum: torch.Tensor) -> None:
    lib = load()
    R, K = _w2d(W)
    _req(w_sum, torch.float32, "w_sum", 1)
    assert w_sum.numel() == K and w_sum.is_contiguous()
    ws = workspace(lib.qt_awq_weight_mean_workspace_bytes(R, K), W.device, "awq_wm")
    check("qt_awq_weight_mean_accumulate", lib.qt_awq_weight_mean_accumulate(
        W.data_ptr(), _dtype_code(W), R, K, W.stride(0), group_size, w_sum.data_ptr(), ws.data_ptr(), ws.numel(),
        _stream()))


def awq_scales(x_abs_sum: torch.Tensor, n_tokens: int, w_sum: torch.Tensor, n_rows: int, n_grid: int = 20,
               duo_scaling: bool = True) -> torch.Tensor:
    lib = load()
    _req(x_abs_sum, torch.float32, "x_abs_sum", 1)
    _req(w_sum, torch.float32, "w_sum", 1)
    K = x_abs_sum.numel()
    assert w_sum.numel() == K
    out = torch.empty((n_grid, K), dtype=torch.float32, device=x_abs_sum.device)
    check("qt_awq_scales", lib.qt_awq_scales(x_abs_sum.data_ptr(), int(n_tokens), w_sum.data_ptr(), int(n_rows), K,
                                             n_grid, int(bool(duo_scaling)), out.data_ptr(), _stream()))
    return out


def symmetrize_lower(G: torch.Tensor) -> None:
    lib = load()
    _req(G, torch.float32, "G", 2)
    assert G.is_contiguous() and G.shape[0] == G.shape[1]
    check("qt_symmetrize_lower", lib.qt_symmetrize_lower(G.data_ptr(), G.shape[0], _stream()))


def awq_loss(W: torch.Tensor, s: torch.Tensor, group_size: int, symmetric: bool, num_bits: int, Gfull: torch.Tensor,
             n_tokens: int, out: torch.Tensor, *, exact: bool = False, weight: float = 1.0,
             accumulate: bool = False) -> None:
    """out[0] (device fp32) = (accumulate ? out[0] : 0) + weight * search loss for the scales s[K]."""
    lib = load()
    R, K = _w2d(W)
    _req(s, torch.float32, "s", 1)
    _req(Gfull, torch.float32, "Gfull", 2)
    _req(out, torch.float32, "out")
    assert s.numel() == K and s.is_contiguous() and Gfull.shape == (K, K) and Gfull.is_contiguous()
    ws = workspace(lib.qt_awq_loss_workspace_bytes(R, K), W.device, "awq_loss")
    check("qt_awq_loss", lib.qt_awq_loss(W.data_ptr(), _dtype_code(W), R, K, W.stride(0), s.data_ptr(), group_size,
                                         int(bool(symmetric)), num_bits, Gfull.data_ptr(), int(n_tokens),
                                         int(bool(exact)), float(weight), int(bool(accumulate)), out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()))


def awq_losses(W: torch.Tensor, scales: torch.Tensor, group_size: int, symmetric: bool, num_bits: int,
               Gfull: torch.Tensor, n_tokens: int, out: torch.Tensor, *, weight: float = 1.0,
               accumulate: bool = False) -> None:
    """out[g] (device fp32 [n_grid]) = (accumulate ? out[g] : 0) + weight * fast search loss for scales[g]."""
    lib = load()
    R, K = _w2d(W)
    _req(scales, torch.float32, "scales", 2)
    _req(Gfull, torch.float32, "Gfull", 2)
    _req(out, torch.float32, "out", 1)
    n_grid = scales.shape[0]
    assert scales.shape[1] == K and scales.is_contiguous() and Gfull.shape == (K, K) and Gfull.is_contiguous()
    assert out.numel() == n_grid and out.is_contiguous()
    ws = workspace(lib.qt_awq_losses_workspace_bytes(R, K, n_grid), W.device, "awq_losses")
    check("qt_awq_losses", lib.qt_awq_losses(W.data_ptr(), _dtype_code(W), R, K, W.stride(0), scales.data_ptr(), n_grid,
                                             group_size, int(bool(symmetric)), num_bits, Gfull.data_ptr(), int(n_tokens),
                                             float(weight), int(bool(accumulate)), out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), _stream()))


def argmin_first(values: torch.Tensor) -> torch.Tensor:
    """Device int32[1]: index of the first minimum (ties keep the lowest index, as upstream's search loop)."""
    lib = load()
    _req(values, torch.float32, "values", 1)
    assert values.is_contiguous()
    out = torch.empty(1, dtype=torch.int32, device=values.device)
    check("qt_argmin_f32", lib.qt_argmin_f32(values.data_ptr(), values.numel(), out.data_ptr(), _stream()))
    return out


def awq_pseudo_quantize(W: torch.Tensor, s: torch.Tensor, group_size: int, symmetric: bool, num_bits: int,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pseudo_quant(W * s) / s in W's dtype (one grid point's trial weights)."""
    lib = load()
    R, K = _w2d(W)
    _req(s, torch.float32, "s", 1)
    assert s.numel() == K and s.is_contiguous()
    if out is None:
        out = torch.empty((R, K), dtype=W.dtype, device=W.device)
    assert out.shape == (R, K) and out.dtype == W.dtype and out.stride(1) == 1 and out.device == W.device
    check("qt_awq_pseudo_quantize",
          lib.qt_awq_pseudo_quantize(W.data_ptr(), _dtype_code(W), R, K, W.stride(0), s.data_ptr(), group_size,
                                     int(bool(symmetric)), num_bits, out.data_ptr(), out.stride(0), _stream()))
    return out


def scale_columns(W: torch.Tensor, s: torch.Tensor, divide: bool = False) -> torch.Tensor:
    lib = load()
    R, K = _w2d(W)
    _req(s, torch.float32, "s", 1)
    assert s.numel() == K and s.is_contiguous()
    out = torch.empty((R, K), dtype=W.dtype, device=W.device)
    check("qt_scale_columns", lib.qt_scale_columns(W.data_ptr(), _dtype_code(W), R, K, W.stride(0), s.data_ptr(),
                                                   int(bool(divide)), out.data_ptr(), out.stride(0), _stream()))
    return out


def rtn_quantize(W: torch.Tensor, scale: torch.Tensor, zp: torch.Tensor, group_size: int, num_bits: int = 4):
    """Plain round-to-nearest levels Qt[K, R] int8."""
    lib = load()
    R, K = _w2d(W)
    _req(scale, torch.float32, "scale", 2)
    _req(zp, torch.float32, "zp", 2)
    G = scale.shape[1]
    assert scale.shape == (R, G) and zp.shape == (R, G) and scale.is_contiguous() and zp.is_contiguous()
    Qt = torch.empty((K, R), dtype=torch.int8, device=W.device)
    check("qt_rtn_quantize", lib.qt_rtn_quantize(W.data_ptr(), _dtype_code(W), R, K, W.stride(0), scale.data_ptr(),
                                                 zp.data_ptr(), G, group_size, num_bits, Qt.data_ptr(), _stream()))
    return Qt


# ---- a13  SmoothQuant ----------------------------------------------------------------------
def col_absmax_accumulate(W: torch.Tensor, wmax: torch.Tensor) -> None:
    lib = load()
    R, K = _w2d(W)
    _req(wmax, torch.float32, "wmax", 1)
    assert wmax.numel() == K and wmax.is_contiguous()
    ws = workspace(lib.qt_col_absmax_workspace_bytes(R, K), W.device, "absmax")
    check("qt_col_absmax_accumulate", lib.qt_col_absmax_accumulate(W.data_ptr(), _dtype_code(W), R, K, W.stride(0),
                                                                   wmax.data_ptr(), ws.data_ptr(), ws.numel(),
                                                                   _stream()))


def smoothquant_scales(cmin: torch.Tensor, cmax: torch.Tensor, wmax: torch.Tensor, alpha: float) -> torch.Tensor:
    lib = load()
    for n, t in (("cmin", cmin), ("cmax", cmax), ("wmax", wmax)):
        _req(t, torch.float32, n, 1)
    K = cmin.numel()
    s = torch.empty(K, dtype=torch.float32, device=cmin.device)
    check("qt_smoothquant_scales", lib.qt_smoothquant_scales(cmin.data_ptr(), cmax.data_ptr(), wmax.data_ptr(), K,
                                                             float(alpha), s.data_ptr(), _stream()))
    return s


# ---- <H, X^T X>_F from the Gram kernel's accumulators (tests; the AWQ loss uses it internally) -------
def xtx_dot(X: torch.Tensor, H: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None,
            accumulate: bool = False) -> torch.Tensor:
    """scale * sum_ij H[i][j] (X^T X)[i][j] with H's lower triangle read; X [n, K] bf16 / fp16 contiguous."""
    lib = load()
    code = _act16(X, "X")
    _req(H, torch.float32, "H", 2)
    n, K = X.shape
    assert X.is_contiguous() and H.is_contiguous() and tuple(H.shape) == (K, K)
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=X.device)
    ws = workspace(lib.qt_xtx_dot_workspace_bytes(n, K), X.device, "xtx_dot")
    check("qt_xtx_dot", lib.qt_xtx_dot(X.data_ptr(), code, n, K, K, H.data_ptr(), float(scale),
                                       out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel(), _stream()))
    return out


# ---- fp32-accurate TN product on the bf16 MFMA (tests / micro-benchmarks) ---------------------
def gemm3_tn(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, kind: int = 0):
    """kind 0: C -= A^T B in place; kind 1: C = A^T B through k-split slabs.  A [k, M], B [k, N] fp32."""
    lib = load()
    _req(A, torch.float32, "A", 2)
    _req(B, torch.float32, "B", 2)
    _req(C, torch.float32, "C", 2)
    k, M = A.shape
    k2, N = B.shape
    assert k == k2 and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1 and tuple(C.shape) == (M, N)
    ws = workspace(lib.qt_gemm3_tn_f32_workspace_bytes(M, N, k), A.device, "gemm3")
    check("qt_gemm3_tn_f32", lib.qt_gemm3_tn_f32(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(),
                                                 C.stride(0), M, N, k, kind, ws.data_ptr(), ws.numel(), _stream()))
    return C


# ---- fp32 TN GEMM (tests / micro-benchmarks) ------------------------------------------------
def sgemm_tn(A: torch.Tensor, B: torch.Tensor, Cin: Optional[torch.Tensor] = None, mode: int = 1,
             skip_zero_k: bool = False, allow_split_k: bool = False, out: Optional[torch.Tensor] = None):
    """acc = A^T B with A [k, M], B [k, N] fp32 row-major; mode 0: Cin - acc, 1: acc, 2: -acc."""
    lib = load()
    _req(A, torch.float32, "A", 2)
    _req(B, torch.float32, "B", 2)
    k, M = A.shape
    k2, N = B.shape
    assert k == k2 and A.stride(1) == 1 and B.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    ws = workspace(lib.qt_sgemm_tn_f32_workspace_bytes(M, N) if allow_split_k else 1, A.device, "sgemm")
    check("qt_sgemm_tn_f32", lib.qt_sgemm_tn_f32(
        A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), _ptr(Cin), Cin.stride(0) if Cin is not None else 0,
        out.data_ptr(), out.stride(0), M, N, k, int(skip_zero_k), mode, int(allow_split_k),
        ws.data_ptr() if allow_split_k else None, ws.numel() if allow_split_k else 0, _stream()))
    return out
