"""ctypes binding of the C ABI declared in ``include/quantool_amd.h``.

The product path has no CPU fallback: if the HIP library is missing this module raises.
"""
from __future__ import annotations

import ctypes
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent.parent
LIB_PATH = _PKG / "lib" / "libquantool_hip.so"

QT_OK = 0
QT_ERR_INVALID = -1
QT_ERR_NOT_PD = -2
QT_ERR_WORKSPACE = -3
QT_ERR_HIP = -4
QT_ERR_UNSUPPORTED = -5

QT_F32, QT_BF16, QT_F16 = 0, 1, 2

# name -> (restype, argtypes); mirrors include/quantool_amd.h one to one
SIGNATURES = {
    "qt_version": (c_int, []),
    "qt_last_error": (c_char_p, []),
    "qt_xtx_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "qt_xtx_accumulate": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_xtx_accumulate_f32_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "qt_xtx_accumulate_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_act_stats_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "qt_act_stats_accumulate": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    "qt_hessian_prepare_workspace_bytes": (c_size_t, [c_int]),
    "qt_hessian_prepare": (c_int, [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "qt_hessian_diag": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "qt_argsort_desc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "qt_cholesky_inverse_upper_workspace_bytes": (c_size_t, [c_int]),
    "qt_cholesky_inverse_upper": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_cholesky_inverse_upper_batched_workspace_bytes": (c_size_t, [c_int, c_int]),
    "qt_cholesky_inverse_upper_batched": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                                  c_size_t, c_void_p]),
    "qt_group_minmax_qparams": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "qt_weight_gather_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "qt_weight_gather_qparams": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "qt_gptq_sweep_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "qt_gptq_sweep": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_gptq_sweep_grouped": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_pack_int4": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "qt_awq_weight_mean_workspace_bytes": (c_size_t, [c_int, c_int]),
    "qt_awq_weight_mean_accumulate": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p, c_void_p,
                                              c_size_t, c_void_p]),
    "qt_awq_scales": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "qt_symmetrize_lower": (c_int, [c_void_p, c_int, c_void_p]),
    "qt_awq_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "qt_awq_loss": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                            c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_awq_losses_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "qt_awq_losses": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                              c_int64, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "qt_argmin_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "qt_awq_pseudo_quantize": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p,
                                       c_int64, c_void_p]),
    "qt_scale_columns": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_void_p, c_int64,
                                 c_void_p]),
    "qt_rtn_quantize": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "qt_col_absmax_workspace_bytes": (c_size_t, [c_int, c_int]),
    "qt_col_absmax_accumulate": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    "qt_smoothquant_scales": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "qt_xtx_dot_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "qt_xtx_dot": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_double, c_void_p, c_int, c_void_p,
                           c_size_t, c_void_p]),
    "qt_gemm3_plan_check": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "qt_xtx_tile_table_check": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "qt_gemm3_tn_f32_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "qt_gemm3_tn_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                c_void_p, c_size_t, c_void_p]),
    "qt_sgemm_tn_f32_workspace_bytes": (c_size_t, [c_int, c_int]),
    "qt_sgemm_tn_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "qt_profile_enable": (c_int, [c_int]),
    "qt_profile_read": (c_int, [c_int, c_void_p, c_void_p]),
    "qt_dequantize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                              c_int, c_int64, c_void_p]),
}

_lib = None


class HipBackendError(RuntimeError):
    """Raised when a C-ABI call returns a negative status."""

    def __init__(self, fn: str, status: int, message: str):
        super().__init__(f"{fn} failed with status {status}: {message}")
        self.status = status


def load() -> ctypes.CDLL:
    """Load libquantool_hip.so (built by ``__graft_entry__.build()``).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"quantool_amd HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the quantization path."
        )
    # torch first: it ships its own libamdhip64, and the process must end up with ONE HIP runtime -- the one that
    # owns torch's device context.  Loaded before torch, this library binds the system runtime (LD_LIBRARY_PATH)
    # instead, whose hipHostMalloc then has no context to pin memory for (seen as "could not build the tile table"
    # when build() and smoke() ran in one process).
    import torch  # noqa: F401

    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(fn_name: str, status: int) -> None:
    if status != QT_OK:
        msg = load().qt_last_error()
        raise HipBackendError(fn_name, status, msg.decode() if msg else "")
