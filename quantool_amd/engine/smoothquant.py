"""SmoothQuant scales and their application (upstream SmoothQuantModifier,
reached through ``src/quantool/methods/llm_compressor/smoothquant/smoothquant.py:77-84``)."""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch

from ..hip import ops


class ChannelMinMax:
    """Running per-channel min / max of a smooth layer's output over all calibration tokens."""

    def __init__(self, K: int, device):
        self.cmin = torch.full((K,), float("inf"), dtype=torch.float32, device=device)
        self.cmax = torch.full((K,), float("-inf"), dtype=torch.float32, device=device)

    def add(self, X: torch.Tensor) -> None:
        X = ops.as_act16(X)
        ops.act_stats_accumulate(X, cmin=self.cmin, cmax=self.cmax)


def smoothquant_scales(stats: ChannelMinMax, balance_weights: Sequence[torch.Tensor], alpha: float = 0.5):
    K = stats.cmin.numel()
    wmax = torch.zeros(K, dtype=torch.float32, device=stats.cmin.device)
    for w in balance_weights:
        ops.col_absmax_accumulate(w, wmax)
    return ops.smoothquant_scales(stats.cmin, stats.cmax, wmax, alpha)


def apply_smoothing(s: torch.Tensor, balance_weights: Sequence[torch.Tensor],
                    smooth_vectors: Sequence[torch.Tensor]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """W_balance *= s[None, :];  norm weight (and bias) /= s."""
    new_w = [ops.scale_columns(w, s) for w in balance_weights]
    new_v = [ops.scale_columns(v.reshape(1, -1), s, divide=True).reshape(v.shape) for v in smooth_vectors]
    return new_w, new_v
