"""Per-mapping AWQ on the device (upstream AWQModifier, reached through
``src/quantool/methods/llm_compressor/awq/awq.py:81`` / ``base.py:161``).

Single-consumer mappings (the parent module is the balance Linear(s) themselves -- v->o, up->down and
the synthetic per-Linear mode): the 20-point grid loss is evaluated through the
Gram matrix (``awq.hip`` header), then the best scales are applied and the weights quantised by plain
round-to-nearest with the standard observer (/7.5), as upstream does after smoothing.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence

import torch

from ..hip import ops
from .schemes import QuantArgs


@dataclass
class AWQResult:
    weight_packed: Optional[torch.Tensor]
    weight_q: Optional[torch.Tensor]
    weight_scale: torch.Tensor
    weight_zero_point: Optional[torch.Tensor]
    weight_g_idx: Optional[torch.Tensor]
    weight_shape: torch.Tensor
    smoothing_scales: torch.Tensor            # fp32 [K]: W_balance *= s, previous op /= s
    best_ratio_idx: torch.Tensor              # int64 [] device
    losses: torch.Tensor                      # fp32 [n_grid] device
    scaled_weight: torch.Tensor = field(repr=False, default=None)   # W * s in model dtype
    scale_f32: torch.Tensor = field(repr=False, default=None)
    zp_f32: torch.Tensor = field(repr=False, default=None)
    Qt: torch.Tensor = field(repr=False, default=None)              # int8 [K, R] levels
    g_of_col: torch.Tensor = field(repr=False, default=None)        # int32 [K]

    def dequantized(self, dtype=torch.float32) -> torch.Tensor:
        """(q - zp) * scale: the weight upstream leaves in the module after the observer pass."""
        if self.Qt is None:
            raise RuntimeError("the integer levels of this result were released (sequential driver, "
                               "QT_RESULT_DETAIL_BYTES): the module's weight holds the dequantised values")
        return ops.dequantize(self.Qt, self.scale_f32, self.zp_f32, self.g_of_col, None, dtype)


def awq_search_enqueue(weights: Sequence[torch.Tensor], batches: Iterable[torch.Tensor], qargs: QuantArgs, *,
                       n_grid: int = 20, duo_scaling: bool = True, device=None) -> "PendingSearch":
    """Device half of ``awq_search``: Gram sum, channel statistics, the candidate scales and their fast losses."""
    K = weights[0].shape[1]
    dev = device or weights[0].device
    gs = qargs.kernel_group_size
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    x_sum = torch.zeros(K, dtype=torch.float32, device=dev)
    n_tokens = 0
    for xb in batches:
        xb = xb.to(dev)
        xb = ops.as_act16(xb)
        ops.xtx_accumulate(xb, G)
        ops.act_stats_accumulate(xb, abs_sum=x_sum)
        n_tokens += xb.numel() // K
    ops.symmetrize_lower(G)
    w_sum = torch.zeros(K, dtype=torch.float32, device=dev)
    n_rows = 0
    for w in weights:
        ops.awq_weight_mean_accumulate(w, gs, w_sum)
        n_rows += w.shape[0]
    scales = ops.awq_scales(x_sum, n_tokens, w_sum, n_rows, n_grid, duo_scaling)
    return search_losses_enqueue(weights, scales, G, n_tokens, qargs)


def awq_search(weights: Sequence[torch.Tensor], batches: Iterable[torch.Tensor], qargs: QuantArgs, *,
               n_grid: int = 20, duo_scaling: bool = True, device=None):
    """Returns (scales[n_grid, K], losses[n_grid], best index tensor, n_tokens)."""
    p = awq_search_enqueue(weights, batches, qargs, n_grid=n_grid, duo_scaling=duo_scaling, device=device)
    losses, best = p.resolve()
    return p.scales, losses, best, p.n_tokens


#: The fast search loss rounds D = W - Wq to bf16 (moves a loss by < 5e-4 relative).  Grid
#: points whose fast losses are within this relative distance of the best one are re-scored exactly.
NEAR_TIE_RTOL = 1.5e-3


@dataclass
class PendingSearch:
    """The device half of a mapping's search, enqueued: every grid point's fast loss and their arg-min.  The host
    half (``resolve``) reads the 20 numbers -- the mapping's one synchronisation -- and re-scores near-ties exactly.
    Split so that a driver can enqueue the searches of several mappings (on their streams) before it waits for
    the first of them."""
    weights: Sequence[torch.Tensor]
    scales: torch.Tensor
    G: torch.Tensor
    n_tokens: int
    qargs: QuantArgs
    losses: torch.Tensor
    best: torch.Tensor
    near_tie_rtol: Optional[float] = None

    def resolve(self):
        """(losses, best index) -- blocks the host until the device half is done."""
        weights, scales, G, qargs = self.weights, self.scales, self.G, self.qargs
        n_grid = scales.shape[0]
        gs = qargs.kernel_group_size
        n_rows = sum(int(w.shape[0]) for w in weights)
        losses, best = self.losses, self.best
        rtol = NEAR_TIE_RTOL if self.near_tie_rtol is None else self.near_tie_rtol
        host = losses.tolist()                                   # one small sync per mapping
        b = int(best.item())
        close = [i for i, l in enumerate(host) if l <= host[b] * (1.0 + rtol) or l != l]
        if len(close) > 1:
            exact_losses = torch.full((n_grid,), float("inf"), dtype=torch.float32, device=G.device)
            for gi in close:
                for j, w in enumerate(weights):
                    ops.awq_loss(w, scales[gi], gs, qargs.symmetric, qargs.num_bits, G, self.n_tokens,
                                 losses[gi:gi + 1], exact=True, weight=w.shape[0] / n_rows, accumulate=j > 0)
                exact_losses[gi] = losses[gi]
            best = ops.argmin_first(exact_losses)
        return losses, best.to(torch.int64).reshape(())


def search_losses_enqueue(weights: Sequence[torch.Tensor], scales: torch.Tensor, G: torch.Tensor, n_tokens: int,
                          qargs: QuantArgs, near_tie_rtol: Optional[float] = None) -> PendingSearch:
    """Fast pass of the search, device only: all grid points of a balance Linear in one Gram launch (its short D
    matrices -- the Linear's rows play the tokens -- fill the chip only together); the row-weighted mean over the
    mapping's balance Linears is accumulated by ``qt_awq_loss`` itself, the arg-min is ``qt_argmin_f32``."""
    n_grid = scales.shape[0]
    gs = qargs.kernel_group_size
    n_rows = sum(int(w.shape[0]) for w in weights)
    losses = torch.zeros(n_grid, dtype=torch.float32, device=G.device)
    sc = scales.contiguous()
    for j, w in enumerate(weights):
        ops.awq_losses(w, sc, gs, qargs.symmetric, qargs.num_bits, G, n_tokens, losses, weight=w.shape[0] / n_rows,
                       accumulate=j > 0)
    best = ops.argmin_first(losses)
    return PendingSearch(weights, scales, G, n_tokens, qargs, losses, best, near_tie_rtol)


def search_losses(weights: Sequence[torch.Tensor], scales: torch.Tensor, G: torch.Tensor, n_tokens: int,
                  qargs: QuantArgs, near_tie_rtol: Optional[float] = None):
    """Losses of every grid point and the arg-min, all on the device.  The host looks at the 20 numbers once, to
    decide whether any runner-up is closer to the winner than the bf16 rounding of D can resolve; those candidates
    (rare) are evaluated again with D and D^T D in fp32 and the arg-min is retaken among them -- so the chosen
    scales do not hinge on that rounding."""
    return search_losses_enqueue(weights, scales, G, n_tokens, qargs, near_tie_rtol).resolve()


def rtn_finalize(ws: torch.Tensor, qargs: QuantArgs, s: Optional[torch.Tensor] = None,
                 best: Optional[torch.Tensor] = None, losses: Optional[torch.Tensor] = None) -> AWQResult:
    """Standard observer (/7.5) + round-to-nearest + pack of an already smoothed weight ``ws``."""
    R, K = ws.shape
    gs = qargs.kernel_group_size
    scale, zp, _, _ = ops.group_minmax_qparams(ws, gs, qargs.symmetric, qargs.num_bits)
    Qt = ops.rtn_quantize(ws, scale, zp, gs, qargs.num_bits)
    packed = ops.pack_int4(Qt) if qargs.num_bits == 4 else None
    sdt = ws.dtype if ws.dtype in (torch.bfloat16, torch.float16) else torch.float32
    g_of_col = (torch.arange(K, dtype=torch.int32, device=ws.device) // (K if gs <= 0 else gs)).contiguous()
    return AWQResult(
        weight_packed=packed, weight_q=None if packed is not None else Qt.t().contiguous(),
        weight_scale=scale.to(sdt), weight_zero_point=None if qargs.symmetric else zp.to(torch.int8),
        weight_g_idx=None, weight_shape=torch.tensor([R, K], dtype=torch.int64), smoothing_scales=s,
        best_ratio_idx=best, losses=losses, scaled_weight=ws, scale_f32=scale, zp_f32=zp, Qt=Qt, g_of_col=g_of_col)


def awq_finalize_group(p: "PendingSearch") -> List[AWQResult]:
    """Host half of a mapping: wait for its search, apply the winner (W_balance *= s, model dtype), then the plain
    observer path."""
    losses, best = p.resolve()
    s = p.scales[best].contiguous()
    return [rtn_finalize(ops.scale_columns(w, s), p.qargs, s, best, losses) for w in p.weights]


def awq_quantize_group(weights: Sequence[torch.Tensor], batches: Iterable[torch.Tensor], qargs: QuantArgs, *,
                       n_grid: int = 20, duo_scaling: bool = True, device=None) -> List[AWQResult]:
    return awq_finalize_group(awq_search_enqueue(weights, batches, qargs, n_grid=n_grid, duo_scaling=duo_scaling,
                                                 device=device))


def awq_quantize_groups(groups: Sequence[tuple], qargs: QuantArgs, *, n_grid: int = 20, duo_scaling: bool = True,
                        device=None) -> List[List[AWQResult]]:
    """Several independent mappings ((weights, batches) each): every search is enqueued before the host waits for
    the first one, so the device works through all of them while the host does its one look per mapping at the 20
    losses (one mapping at a time, the host's round trip after every search left the GPU idle: 78 -> 6x ms per
    Llama-3-8B layer); apply / round / pack follow per mapping.  Results as ``awq_quantize_group``
    would give them, in the order of ``groups``."""
    pending = [awq_search_enqueue(w, b, qargs, n_grid=n_grid, duo_scaling=duo_scaling, device=device) for w, b in groups]
    return [awq_finalize_group(p) for p in pending]
