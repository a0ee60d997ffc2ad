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
from typing import Dict, List

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


def _parse_size(x) -> int:
    if isinstance(x, (int, float)):
        return int(x)
    t = str(x).strip().upper()
    for suf, mul in (("GIB", 2 ** 30), ("MIB", 2 ** 20), ("KIB", 2 ** 10), ("GB", 10 ** 9), ("MB", 10 ** 6), ("KB", 10 ** 3)):
        if t.endswith(suf):
            return int(float(t[: -len(suf)]) * mul)
    return int(t)


def plan_shards(sizes: Dict[str, int], max_shard_size) -> List[List[str]]:
    """Greedy split in key order, as ``save_pretrained`` shards a state_dict: a tensor never straddles
    two files; a single tensor larger than the limit gets a file of its own."""
    limit = _parse_size(max_shard_size)
    shards: List[List[str]] = [[]]
    used = 0
    for name, nbytes in sizes.items():
        if shards[-1] and used + nbytes > limit:
            shards.append([])
            used = 0
        shards[-1].append(name)
        used += nbytes
    return shards


def save_state(state: Dict[str, torch.Tensor], qconfig: dict, save_directory, base_config: dict = None,
               max_shard_size="5GB") -> None:
    """One ``model.safetensors`` when the state fits ``max_shard_size`` (``save_pretrained``'s default
    5GB), else ``model-0000i-of-0000n.safetensors`` files plus ``model.safetensors.index.json``
    (``{"metadata": {"total_size": bytes}, "weight_map": {tensor name: file}}``) -- a Llama-3-70B
    W4A16 state is ~35 GB.  Tensors are moved to the host one shard at a time."""
    from safetensors.torch import save_file

    dest = Path(save_directory)
    dest.mkdir(parents=True, exist_ok=True)
    sizes = {k: v.numel() * v.element_size() for k, v in state.items()}
    shards = plan_shards(sizes, max_shard_size)
    for old in list(dest.glob("model*.safetensors")) + list(dest.glob("model.safetensors.index.json")):
        old.unlink()
    if len(shards) == 1:
        save_file({k: state[k].detach().to("cpu").contiguous() for k in shards[0]}, str(dest / "model.safetensors"),
                  metadata={"format": "pt"})
    else:
        weight_map = {}
        for i, names in enumerate(shards, 1):
            fname = f"model-{i:05d}-of-{len(shards):05d}.safetensors"
            save_file({k: state[k].detach().to("cpu").contiguous() for k in names}, str(dest / fname),
                      metadata={"format": "pt"})
            weight_map.update({k: fname for k in names})
        index = {"metadata": {"total_size": int(sum(sizes.values()))}, "weight_map": weight_map}
        with open(dest / "model.safetensors.index.json", "w", encoding="utf-8") as fh:
            json.dump(index, fh, indent=2, sort_keys=True)
    cfg = dict(base_config or {})
    cfg["quantization_config"] = qconfig
    with open(dest / "config.json", "w", encoding="utf-8") as fh:
        json.dump(cfg, fh, indent=2, default=str)


def load_state(save_directory) -> Dict[str, torch.Tensor]:
    """Read back what ``save_state`` wrote (single file or sharded), for tests and tools."""
    from safetensors.torch import load_file

    dest = Path(save_directory)
    idx = dest / "model.safetensors.index.json"
    if not idx.exists():
        return load_file(str(dest / "model.safetensors"))
    out: Dict[str, torch.Tensor] = {}
    for fname in sorted(set(json.loads(idx.read_text())["weight_map"].values())):
        out.update(load_file(str(dest / fname)))
    return out
