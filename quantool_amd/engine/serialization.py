"""compressed-tensors "pack-quantized" on-disk layout (produced upstream by
``last_model.save_pretrained(dest, save_compressed=True)``, ``base.py:188``).

Tensor names and the ``quantization_config`` block follow SURVEY Appendix A.5 (field names
corroborated by the loader shipped in ``transformers/integrations/compressed_tensors.py``; the
writer itself is unverifiable offline).
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Dict

import torch


def result_tensors(r) -> Dict[str, torch.Tensor]:
    """state_dict entries of one quantized Linear (GPTQResult / AWQ result)."""
    out: Dict[str, torch.Tensor] = {}
    if r.weight_packed is not None:
        out["weight_packed"] = r.weight_packed
    else:
        out["weight"] = r.weight_q
    out["weight_scale"] = r.weight_scale
    if r.weight_zero_point is not None:
        out["weight_zero_point"] = r.weight_zero_point
    if r.weight_g_idx is not None:
        out["weight_g_idx"] = r.weight_g_idx
    out["weight_shape"] = r.weight_shape
    return out


def quantization_config(weight_config: dict, fmt: str, ignore, input_activations: dict = None) -> dict:
    """``input_activations`` is the scheme's activation block (W8A8 / INT8 / W4A8: 8-bit dynamic
    per-token): dynamic observers hold no calibration state, so the block is
    configuration only -- the runtime that loads the checkpoint quantises activations on the fly."""
    return {
        "quant_method": "compressed-tensors",
        "format": fmt,
        "quantization_status": "compressed",
        "global_compression_ratio": None,
        "config_groups": {
            "group_0": {"targets": ["Linear"], "weights": weight_config, "input_activations": input_activations,
                        "output_activations": None}
        },
        "ignore": list(ignore),
        "kv_cache_scheme": None,
        "sparsity_config": {},
    }


def save_state(state: Dict[str, torch.Tensor], qconfig: dict, save_directory, base_config: dict = None) -> None:
    from safetensors.torch import save_file

    dest = Path(save_directory)
    dest.mkdir(parents=True, exist_ok=True)
    cpu_state = {k: v.detach().to("cpu").contiguous() for k, v in state.items()}
    save_file(cpu_state, str(dest / "model.safetensors"), metadata={"format": "pt"})
    cfg = dict(base_config or {})
    cfg["quantization_config"] = qconfig
    with open(dest / "config.json", "w", encoding="utf-8") as fh:
        json.dump(cfg, fh, indent=2, default=str)
