"""AWQ on ``torch.nn.Module`` decoder layers: the mapping resolver and the per-layer search / apply /
quantise cycle (SURVEY.md 8f row N1 for ``method=awq``; upstream AWQModifier as recalled in SURVEY
Appendix A.3, reached from ``src/quantool/methods/llm_compressor/awq/awq.py:81`` / ``base.py:161``).

A *mapping* ties one smooth layer (a norm, or a Linear whose output feeds the balance layers
unchanged) to the balance Linears that read its output.  Per mapping, on the device:

1. ``x_mean`` = mean |x| per input channel of the balance layers  (``qt_act_stats_accumulate``)
2. ``w_mean`` = mean group-normalised |W| over all balance rows   (``qt_awq_weight_mean_accumulate``)
3. the 20 candidate scale vectors                                   (``qt_awq_scales``)
4. search loss of each candidate:
     * one balance Linear (v->o, up->down): the parent module *is* that Linear, so the loss is the
       Frobenius product <X^T X, D^T D> -- ``qt_awq_loss`` on the Gram sum taken in step 1's pass;
     * several balance Linears (norm->q/k/v, norm->gate/up): the parent is their common ancestor
       (self_attn, mlp); trial weights come from ``qt_awq_pseudo_quantize`` and the parent is run by
       torch on the cached inputs -- the model's own forward is the caller's code, not this path's.
5. apply the winner: ``W_balance *= s`` (``qt_scale_columns``), smooth layer ``/= s``.

After every mapping of the layer has been applied, each targeted Linear goes through the standard
observer + round-to-nearest + pack (``awq_linear.rtn_finalize``).
"""
from __future__ import annotations

import logging
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..hip import ops
from .awq_linear import AWQResult, rtn_finalize
from .schemes import QuantArgs

logger = logging.getLogger(__name__)

#: Llama-family defaults: smooth-layer pattern -> balance-layer patterns
DEFAULT_MAPPINGS: Tuple[Tuple[str, Tuple[str, ...]], ...] = (
    ("re:.*input_layernorm$", ("re:.*q_proj$", "re:.*k_proj$", "re:.*v_proj$")),
    ("re:.*v_proj$", ("re:.*o_proj$",)),
    ("re:.*post_attention_layernorm$", ("re:.*gate_proj$", "re:.*up_proj$")),
    ("re:.*up_proj$", ("re:.*down_proj$",)),
)


@dataclass
class ResolvedMapping:
    smooth_name: str
    smooth: nn.Module
    balance_names: List[str]
    balance: List[nn.Linear]
    parent_name: str
    parent: nn.Module

    @property
    def single(self) -> bool:
        return self.parent is self.balance[0]


def _pattern_hits(pattern: str, name: str) -> bool:
    if pattern.startswith("re:"):
        return re.match(pattern[3:], name) is not None
    return name == pattern or name.endswith("." + pattern)


def normalise_mappings(mappings: Optional[Sequence[Any]]) -> List[Tuple[str, List[str]]]:
    """Accepts upstream's ``AWQMapping``-like objects / dicts (``smooth_layer``, ``balance_layers``)
    and plain ``[smooth, [balance, ...]]`` pairs as a quantool YAML would carry them."""
    if not mappings:
        return [(s, list(b)) for s, b in DEFAULT_MAPPINGS]
    out = []
    for m in mappings:
        if isinstance(m, dict):
            smooth, balance = m["smooth_layer"], m["balance_layers"]
        elif hasattr(m, "smooth_layer"):
            smooth, balance = m.smooth_layer, m.balance_layers
        else:
            smooth, balance = m[0], m[1]
        out.append((str(smooth), [balance] if isinstance(balance, str) else [str(b) for b in balance]))
    return out


def resolve_mappings(layer: nn.Module, layer_name: str, mappings: Sequence[Tuple[str, List[str]]],
                     wanted) -> List[ResolvedMapping]:
    """Match the patterns inside one decoder layer.  A mapping is dropped when its smooth layer or
    all balance layers are absent, when a balance layer is not targeted, or when a Linear smooth
    layer's output width differs from the balance input width (v_proj -> o_proj under GQA)."""
    named = {(f"{layer_name}.{n}" if n else layer_name): m for n, m in layer.named_modules()}
    resolved: List[ResolvedMapping] = []
    for smooth_pat, balance_pats in mappings:
        smooth_hits = [n for n in named if _pattern_hits(smooth_pat, n)]
        bal = [n for n in named if isinstance(named[n], nn.Linear) and any(_pattern_hits(p, n) for p in balance_pats)]
        if len(smooth_hits) != 1 or not bal:
            continue
        if not all(wanted(n, named[n]) for n in bal):
            continue
        smooth = named[smooth_hits[0]]
        K = named[bal[0]].in_features
        if any(named[n].in_features != K for n in bal):
            continue
        if isinstance(smooth, nn.Linear) and smooth.out_features != K:
            logger.info(f"AWQ mapping {smooth_hits[0]} -> {bal} skipped: widths {smooth.out_features} != {K}")
            continue
        if not isinstance(smooth, nn.Linear) and (getattr(smooth, "weight", None) is None
                                                  or smooth.weight.numel() != K):
            continue
        if len(bal) == 1:
            parent_name = bal[0]
        else:
            parts = [n.split(".") for n in bal]
            common = []
            for segs in zip(*parts):
                if len(set(segs)) != 1:
                    break
                common.append(segs[0])
            parent_name = ".".join(common)
            if parent_name not in named:
                continue
        resolved.append(ResolvedMapping(smooth_hits[0], smooth, bal, [named[n] for n in bal], parent_name,
                                        named[parent_name]))
    return resolved


def _first(out):
    return out[0] if isinstance(out, (tuple, list)) else out


def _as_rows16(x: torch.Tensor) -> torch.Tensor:
    x = x.reshape(-1, x.shape[-1])
    return ops.as_act16(x)


class _Capture:
    """What one mapping needs from the calibration pass over the layer."""

    def __init__(self, mp: ResolvedMapping, dev):
        K = mp.balance[0].in_features
        self.x_abs_sum = torch.zeros(K, dtype=torch.float32, device=dev)
        self.n_tokens = 0
        self.gram = torch.zeros((K, K), dtype=torch.float32, device=dev) if mp.single else None
        self.parent_calls: List[tuple] = []      # (args, kwargs) per batch, multi-balance mappings only

    def on_balance_input(self, _mod, args):
        x = _as_rows16(args[0])
        ops.act_stats_accumulate(x, abs_sum=self.x_abs_sum)
        if self.gram is not None:
            ops.xtx_accumulate(x, self.gram)
        self.n_tokens += x.shape[0]

    def on_parent_call(self, _mod, args, kwargs):
        self.parent_calls.append((args, kwargs))


def _search(mp: ResolvedMapping, cap: _Capture, qargs: QuantArgs, n_grid: int, duo_scaling: bool):
    """Returns (scales[n_grid, K], losses[n_grid], best index tensor)."""
    dev = cap.x_abs_sum.device
    gs = qargs.kernel_group_size
    K = mp.balance[0].in_features
    w_sum = torch.zeros(K, dtype=torch.float32, device=dev)
    n_rows = 0
    for lin in mp.balance:
        ops.awq_weight_mean_accumulate(lin.weight.data, gs, w_sum)
        n_rows += lin.out_features
    scales = ops.awq_scales(cap.x_abs_sum, cap.n_tokens, w_sum, n_rows, n_grid, duo_scaling)
    losses = torch.zeros(n_grid, dtype=torch.float32, device=dev)
    if mp.single:
        from .awq_linear import search_losses

        ops.symmetrize_lower(cap.gram)
        losses, best = search_losses([mp.balance[0].weight.data], scales, cap.gram, cap.n_tokens, qargs)
        return scales, losses, best
    else:
        originals = [lin.weight.data.clone() for lin in mp.balance]
        fp_out = [_first(mp.parent(*a, **kw)).float() for a, kw in cap.parent_calls]
        n_elem = float(sum(o.numel() for o in fp_out))
        try:
            for gi in range(n_grid):
                for lin, w0 in zip(mp.balance, originals):
                    ops.awq_pseudo_quantize(w0, scales[gi], gs, qargs.symmetric, qargs.num_bits, out=lin.weight.data)
                sq = torch.zeros((), dtype=torch.float32, device=dev)
                for (a, kw), ref in zip(cap.parent_calls, fp_out):
                    sq += (ref - _first(mp.parent(*a, **kw)).float()).pow(2).sum()
                losses[gi] = sq / n_elem
        finally:
            for lin, w0 in zip(mp.balance, originals):
                lin.weight.data.copy_(w0)
    # multi-consumer mapping: the parent module's own forward (the caller's torch code) produced the
    # losses above; the arg-min of the 20 numbers is the device kernel the Gram form uses
    return scales, losses, ops.argmin_first(losses).to(torch.int64).reshape(())


def _apply(mp: ResolvedMapping, s: torch.Tensor) -> None:
    for lin in mp.balance:
        lin.weight.data.copy_(ops.scale_columns(lin.weight.data, s))
    K = s.numel()
    sm = mp.smooth
    if isinstance(sm, nn.Linear):
        # the last K output rows (and bias entries) of the producing Linear
        rows = sm.weight.data[-K:]
        rows.copy_((rows.float() / s[:, None]).to(rows.dtype))
        if sm.bias is not None:
            sm.bias.data[-K:].copy_((sm.bias.data[-K:].float() / s).to(sm.bias.dtype))
    else:
        sm.weight.data.copy_((sm.weight.data.float() / s).to(sm.weight.dtype))
        if getattr(sm, "bias", None) is not None:
            sm.bias.data.copy_((sm.bias.data.float() / s).to(sm.bias.dtype))


def awq_layer(layer: nn.Module, layer_name: str, cache: Sequence[tuple], modifier, dev) -> Dict[str, AWQResult]:
    """Search, smooth and quantise one decoder layer in place; returns the per-Linear results."""
    qargs = modifier.weight_args()
    named = {(f"{layer_name}.{n}" if n else layer_name): m for n, m in layer.named_modules()}
    linears = {n: m for n, m in named.items() if isinstance(m, nn.Linear) and modifier.wants(n, m)}
    mappings = resolve_mappings(layer, layer_name, normalise_mappings(modifier.mappings), modifier.wants)

    caps = [_Capture(mp, dev) for mp in mappings]
    hooks = []
    for mp, cap in zip(mappings, caps):
        hooks.append(mp.balance[0].register_forward_pre_hook(cap.on_balance_input))
        if not mp.single:
            hooks.append(mp.parent.register_forward_pre_hook(cap.on_parent_call, with_kwargs=True))
    try:
        for args, kwargs in cache:
            layer(*args, **kwargs)
    finally:
        for hk in hooks:
            hk.remove()

    info: Dict[str, tuple] = {}
    for mp, cap in zip(mappings, caps):
        scales, losses, best = _search(mp, cap, qargs, modifier.n_grid, modifier.duo_scaling)
        s = scales[best].contiguous()
        _apply(mp, s)
        for n in mp.balance_names:
            info[n] = (s, best, losses)
        cap.parent_calls.clear()
        cap.gram = None
    results: Dict[str, AWQResult] = {}
    for n, lin in linears.items():
        s, best, losses = info.get(n, (None, None, None))
        r = rtn_finalize(lin.weight.data, qargs, s, best, losses)
        lin.weight.data.copy_(r.dequantized(lin.weight.dtype))
        r.scaled_weight = None
        results[n] = r
    logger.info(f"AWQ {layer_name}: {len(mappings)} mappings, {len(linears)} Linears quantised")
    return results
