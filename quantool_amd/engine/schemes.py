"""Preset quantization schemes, by the names quantool's plugins accept.

The reference validates ``level`` with ``compressed_tensors.quantization.is_preset_scheme``
(``src/quantool/methods/llm_compressor/gptq/gptq.py:62-67``) and lists the valid names in its
error message (``gptq.py:64-66``).  The weight arguments below follow the published
compressed-tensors presets as recalled in SURVEY.md Appendix A.2 (unverifiable offline).
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict, Optional


@dataclass(frozen=True)
class QuantArgs:
    num_bits: int = 4
    type: str = "int"
    symmetric: bool = True
    strategy: str = "group"          # "group" | "channel" | "token" | "tensor"
    group_size: Optional[int] = 128  # None for channel / token
    dynamic: bool = False
    actorder: Optional[str] = None   # None | "static" ("weight") | "group"
    observer: str = "minmax"

    @property
    def kernel_group_size(self) -> int:
        """Group size as the C ABI takes it (<= 0: one group per row)."""
        return int(self.group_size) if (self.strategy == "group" and self.group_size) else -1

    def to_config(self) -> dict:
        return {
            "num_bits": self.num_bits, "type": self.type, "symmetric": self.symmetric,
            "strategy": self.strategy, "group_size": self.group_size, "dynamic": self.dynamic,
            "actorder": self.actorder, "observer": self.observer, "block_structure": None,
        }


@dataclass(frozen=True)
class QuantScheme:
    name: str
    weights: Optional[QuantArgs]
    input_activations: Optional[QuantArgs] = None
    format: str = "pack-quantized"
    supported: bool = True  # False: a valid preset name this backend does not implement


_W8_CH = QuantArgs(num_bits=8, strategy="channel", group_size=None)
_A8_TOKEN_DYN = QuantArgs(num_bits=8, strategy="token", group_size=None, dynamic=True, observer="")

PRESET_SCHEMES: Dict[str, QuantScheme] = {
    "UNQUANTIZED": QuantScheme("UNQUANTIZED", None, None, "dense"),
    "W8A16": QuantScheme("W8A16", _W8_CH, None, "pack-quantized"),
    "W4A16": QuantScheme("W4A16", QuantArgs(num_bits=4, symmetric=True, group_size=128), None, "pack-quantized"),
    "W4A16_ASYM": QuantScheme("W4A16_ASYM", QuantArgs(num_bits=4, symmetric=False, group_size=128), None,
                              "pack-quantized"),
    "W8A8": QuantScheme("W8A8", _W8_CH, _A8_TOKEN_DYN, "int-quantized"),
    "INT8": QuantScheme("INT8", _W8_CH, _A8_TOKEN_DYN, "int-quantized"),
    "W4A8": QuantScheme("W4A8", QuantArgs(num_bits=4, symmetric=True, group_size=128),
                        replace(_A8_TOKEN_DYN, symmetric=False), "int-quantized"),
    # float presets: valid names upstream, not implemented by this integer backend
    "FP8": QuantScheme("FP8", None, None, "float-quantized", supported=False),
    "FP8_DYNAMIC": QuantScheme("FP8_DYNAMIC", None, None, "float-quantized", supported=False),
    "FP8_BLOCK": QuantScheme("FP8_BLOCK", None, None, "float-quantized", supported=False),
    "NVFP4A16": QuantScheme("NVFP4A16", None, None, "nvfp4-pack-quantized", supported=False),
    "NVFP4": QuantScheme("NVFP4", None, None, "nvfp4-pack-quantized", supported=False),
}


def is_preset_scheme(name) -> bool:
    """Counterpart of ``compressed_tensors.quantization.is_preset_scheme`` (gptq.py:51,62)."""
    return isinstance(name, str) and name.upper() in PRESET_SCHEMES


def preset_name_to_scheme(name: str) -> QuantScheme:
    if not is_preset_scheme(name):
        raise KeyError(f"Unknown preset scheme name {name}, available names: {list(PRESET_SCHEMES)}")
    scheme = PRESET_SCHEMES[name.upper()]
    if not scheme.supported:
        raise NotImplementedError(
            f"Scheme '{name}' is a valid preset name but is not implemented by the MI355X integer "
            "backend (supported: W4A16, W4A16_ASYM, W8A16, W8A8, INT8, W4A8)")
    return scheme
