"""N2: the compressed-tensors on-disk writer (produced upstream by
``save_pretrained(dest, save_compressed=True)``, reference base.py:188).  FORMAT UNPINNED: neither
``compressed-tensors`` nor any checkpoint written by it is available here, so these tests pin the
writer against (i) its own reader, (ii) the sharding contract of ``save_pretrained`` (index file with
``metadata.total_size`` and ``weight_map``) and (iii) the argument names of the loader shipped in
``transformers`` (``CompressedTensorsConfig.__init__``), the only locally available view of the block."""
import inspect
import json

import pytest
import torch

from quantool_amd.engine.schemes import preset_name_to_scheme
from quantool_amd.engine.serialization import load_state, plan_shards, quantization_config, save_state


def _state():
    g = torch.Generator().manual_seed(0)
    sd = {}
    for i in range(6):
        sd[f"model.layers.{i}.mlp.down_proj.weight_packed"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (64, 32), generator=g,
                                                                            dtype=torch.int32)
        sd[f"model.layers.{i}.mlp.down_proj.weight_scale"] = torch.randn(64, 2, generator=g).to(torch.bfloat16)
        sd[f"model.layers.{i}.mlp.down_proj.weight_shape"] = torch.tensor([64, 256], dtype=torch.int64)
    sd["lm_head.weight"] = torch.randn(100, 64, generator=g).to(torch.float16)
    return sd


def test_single_file_when_it_fits(tmp_path):
    sd = _state()
    q = quantization_config(preset_name_to_scheme("W4A16").weights.to_config(), "pack-quantized", ["lm_head"])
    save_state(sd, q, tmp_path, {"hidden_size": 64})
    assert (tmp_path / "model.safetensors").exists() and not (tmp_path / "model.safetensors.index.json").exists()
    back = load_state(tmp_path)
    assert back.keys() == sd.keys() and all(torch.equal(back[k], sd[k]) for k in sd)
    cfg = json.loads((tmp_path / "config.json").read_text())
    assert cfg["hidden_size"] == 64 and cfg["quantization_config"]["format"] == "pack-quantized"


def test_sharded_with_index_above_the_limit(tmp_path):
    sd = _state()
    q = quantization_config(preset_name_to_scheme("W4A16").weights.to_config(), "pack-quantized", ["lm_head"])
    save_state(sd, q, tmp_path, max_shard_size=20_000)          # bytes: forces several files
    idx = json.loads((tmp_path / "model.safetensors.index.json").read_text())
    files = sorted(set(idx["weight_map"].values()))
    assert len(files) > 1 and not (tmp_path / "model.safetensors").exists()
    n = len(files)
    assert files == [f"model-{i:05d}-of-{n:05d}.safetensors" for i in range(1, n + 1)]
    assert all((tmp_path / f).exists() for f in files)
    assert set(idx["weight_map"]) == set(sd)
    assert idx["metadata"]["total_size"] == sum(v.numel() * v.element_size() for v in sd.values())
    back = load_state(tmp_path)
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    # a re-save that fits one file removes the stale shards and index
    save_state(sd, q, tmp_path)
    assert (tmp_path / "model.safetensors").exists() and not list(tmp_path.glob("model-*.safetensors"))
    assert not (tmp_path / "model.safetensors.index.json").exists()


def test_plan_shards_never_splits_a_tensor_and_keeps_order():
    sizes = {"a": 6, "b": 6, "c": 20, "d": 1, "e": 1}
    assert plan_shards(sizes, 10) == [["a"], ["b"], ["c"], ["d", "e"]]
    assert plan_shards(sizes, "1KB") == [list(sizes)]
    assert plan_shards({}, 10) == [[]]


@pytest.mark.parametrize("scheme", ["W4A16", "W4A16_ASYM", "W8A8", "W8A16"])
def test_quantization_config_block_uses_the_loaders_argument_names(scheme):
    """``transformers``' ``CompressedTensorsConfig`` cannot be instantiated without the
    ``compressed-tensors`` package; its ``__init__`` signature is what can be checked offline: every
    top-level key this writer emits must be one of its named arguments (``sparsity_config`` travels
    in ``**kwargs``)."""
    from transformers.utils.quantization_config import CompressedTensorsConfig

    sch = preset_name_to_scheme(scheme)
    acts = sch.input_activations.to_config() if sch.input_activations is not None else None
    q = quantization_config(sch.weights.to_config(), sch.format, ["lm_head"], acts)
    params = set(inspect.signature(CompressedTensorsConfig.__init__).parameters) - {"self", "kwargs"}
    assert set(q) - {"sparsity_config"} <= params, set(q) - params
    assert q["quant_method"] == "compressed-tensors" and q["quantization_status"] == "compressed"
    g0 = q["config_groups"]["group_0"]
    assert g0["targets"] == ["Linear"] and {"num_bits", "type", "symmetric", "strategy"} <= set(g0["weights"])
    json.dumps(q)      # serialisable as it stands
