"""Recipe objects: the three modifiers quantool's plugins construct.

Reference call sites: ``GPTQModifier(**modifier_kwargs)`` (``gptq/gptq.py:86``),
``AWQModifier(**modifier_kwargs)`` (``awq/awq.py:81``), ``[SmoothQuantModifier(smoothing_strength=...),
GPTQModifier(scheme, targets, ignore)]`` (``smoothquant/smoothquant.py:77-84``).  Keyword names
and defaults follow the upstream modifiers as recalled in SURVEY.md Appendix A.
"""
from __future__ import annotations

import fnmatch
import re
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Union

from .schemes import QuantArgs, QuantScheme, preset_name_to_scheme


def _as_list(x) -> List[str]:
    if x is None:
        return []
    if isinstance(x, str):
        return [x]
    return list(x)


def match_target(name: str, module, targets: Sequence[str]) -> bool:
    """compressed-tensors style matching: a target is a class name ("Linear"), an exact module
    name, or a regex prefixed with ``re:``."""
    cls = type(module).__name__
    for t in targets:
        if t.startswith("re:"):
            if re.match(t[3:], name):
                return True
        elif t == cls or t == name or fnmatch.fnmatch(name, t):
            return True
    return False


@dataclass
class GPTQModifier:
    scheme: str = "W4A16"
    targets: Union[str, List[str]] = "Linear"
    ignore: List[str] = field(default_factory=lambda: ["lm_head"])
    block_size: int = 128
    dampening_frac: float = 0.01
    sequential_targets: Optional[Union[str, List[str]]] = None
    actorder: Optional[str] = "static"   # upstream default since 0.8 (recalled)
    offload_hessians: bool = False

    def __post_init__(self):
        self.targets = _as_list(self.targets)
        self.ignore = _as_list(self.ignore)
        self.resolved_scheme: QuantScheme = preset_name_to_scheme(self.scheme)

    def weight_args(self) -> QuantArgs:
        w = self.resolved_scheme.weights
        if w is None:
            raise ValueError(f"scheme {self.scheme} does not quantize weights")
        from dataclasses import replace

        # activation ordering only applies to group-wise weights (upstream ignores it otherwise)
        ao = self.actorder if w.strategy == "group" else None
        return replace(w, actorder=ao)

    def wants(self, name: str, module) -> bool:
        if match_target(name, module, self.ignore) or name.split(".")[-1] in self.ignore:
            return False
        return match_target(name, module, self.targets)


@dataclass
class AWQModifier:
    scheme: str = "W4A16"
    targets: Union[str, List[str]] = "Linear"
    ignore: List[str] = field(default_factory=lambda: ["lm_head"])
    mappings: Optional[List[Any]] = None
    smoothing_strength: Optional[float] = None   # accepted for quantool's pass-through (awq.py:77-79)
    duo_scaling: bool = True
    n_grid: int = 20

    def __post_init__(self):
        self.targets = _as_list(self.targets)
        self.ignore = _as_list(self.ignore)
        self.resolved_scheme: QuantScheme = preset_name_to_scheme(self.scheme)

    def weight_args(self) -> QuantArgs:
        w = self.resolved_scheme.weights
        if w is None:
            raise ValueError(f"scheme {self.scheme} does not quantize weights")
        return w

    def wants(self, name: str, module) -> bool:
        if match_target(name, module, self.ignore) or name.split(".")[-1] in self.ignore:
            return False
        return match_target(name, module, self.targets)


@dataclass
class SmoothQuantModifier:
    smoothing_strength: float = 0.5
    mappings: Optional[List[Any]] = None
    ignore: Optional[List[str]] = None
