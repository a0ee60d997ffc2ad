"""Decoder-layer Linear shapes of the models BASELINE.json names (public configs).

Each entry lists the Linears of one decoder layer grouped by the activation they read:
Linears in one group see the same input, so upstream's per-Linear Hessians are identical and the
backend computes one (``gptq_linear.HessianAccumulator``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class LayerShape:
    name: str
    n_layers: int
    # (group name, in_features K, [(linear name, out_features R), ...])
    groups: Tuple[Tuple[str, int, Tuple[Tuple[str, int], ...]], ...]

    @property
    def weights_per_layer(self) -> int:
        return sum(K * R for _, K, lins in self.groups for _, R in lins)

    @property
    def total_weights(self) -> int:
        return self.weights_per_layer * self.n_layers


def _llama(name, hidden, inter, kv_out, n_layers) -> LayerShape:
    return LayerShape(name, n_layers, (
        ("attn_in", hidden, (("q_proj", hidden), ("k_proj", kv_out), ("v_proj", kv_out))),
        ("attn_out", hidden, (("o_proj", hidden),)),
        ("mlp_in", hidden, (("gate_proj", inter), ("up_proj", inter))),
        ("mlp_down", inter, (("down_proj", hidden),)),
    ))


def _mixtral(name, hidden, inter, kv_out, n_layers, n_experts) -> LayerShape:
    groups = [
        ("attn_in", hidden, (("q_proj", hidden), ("k_proj", kv_out), ("v_proj", kv_out))),
        ("attn_out", hidden, (("o_proj", hidden),)),
    ]
    for e in range(n_experts):
        groups.append((f"expert{e}_in", hidden, ((f"experts.{e}.w1", inter), (f"experts.{e}.w3", inter))))
        groups.append((f"expert{e}_down", inter, ((f"experts.{e}.w2", hidden),)))
    return LayerShape(name, n_layers, tuple(groups))


MODEL_SHAPES: Dict[str, LayerShape] = {
    "llama-3-8b": _llama("llama-3-8b", 4096, 14336, 1024, 32),
    "llama-3-70b": _llama("llama-3-70b", 8192, 28672, 1024, 80),
    "mixtral-8x7b": _mixtral("mixtral-8x7b", 4096, 14336, 1024, 32, 8),
    "opt-125m": LayerShape("opt-125m", 12, (
        ("attn_in", 768, (("q_proj", 768), ("k_proj", 768), ("v_proj", 768))),
        ("attn_out", 768, (("out_proj", 768),)),
        ("fc1_in", 768, (("fc1", 3072),)),
        ("fc2_in", 3072, (("fc2", 768),)),
    )),
}


def scaled(shape: LayerShape, divisor: int) -> LayerShape:
    """The same layer with every in / out width divided by ``divisor`` (kept a multiple of 128, at least 128): the
    N-rank control flow of a big configuration (which groups exist, which are split over ranks, how many Linears the
    final gather carries, ragged expert routing) at a size a one-GPU rehearsal finishes in seconds.  Not a benchmark
    workload -- ``bench.py`` labels such a line."""
    if divisor <= 1:
        return shape

    def dim(x: int) -> int:
        return max(128, x // divisor // 128 * 128)

    return LayerShape(f"{shape.name}/{divisor}", shape.n_layers,
                      tuple((g, dim(K), tuple((n, dim(R)) for n, R in lins)) for g, K, lins in shape.groups))
