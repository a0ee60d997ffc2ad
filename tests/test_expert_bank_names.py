"""Saving a model whose fused MoE expert banks were unfused for calibration: the checkpoint
must not carry ``_UnfusedExperts``' internal module names; per-expert gate / up / down entries are written
instead, the fused gate_up rows split in two (exact: rows are independent)."""
import types

import torch
from torch import nn

from quantool_amd.engine import sequential as sq
from quantool_amd.engine.serialization import load_state

E, H, I = 3, 16, 24


class FusedExperts(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.gate_up_proj = nn.Parameter(torch.randn(E, 2 * I, H, generator=g))
        self.down_proj = nn.Parameter(torch.randn(E, H, I, generator=g))
        self.act_fn = nn.SiLU()


class Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.experts = FusedExperts()
        self.norm = nn.LayerNorm(H)


class Res:      # the fields result_tensors() reads
    def __init__(self, R, K, g_idx=False):
        g = torch.Generator().manual_seed(R * 1000 + K)
        self.weight_packed = torch.randint(-2 ** 31, 2 ** 31 - 1, (R, K // 8), generator=g, dtype=torch.int32)
        self.weight_q = None
        self.weight_scale = torch.randn(R, 1, generator=g).to(torch.bfloat16)
        self.weight_zero_point = torch.randint(-8, 8, (R, 1), generator=g, dtype=torch.int8)
        self.weight_g_idx = torch.arange(K, dtype=torch.int32) if g_idx else None
        self.weight_shape = torch.tensor([R, K], dtype=torch.int64)


def _model():
    m = nn.Module()
    m.layers = nn.ModuleList([Block()])
    fused = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert sq.unfuse_expert_banks(m) == 1
    return m, fused


def test_compressed_save_uses_per_expert_names(tmp_path):
    m, _ = _model()
    res = {}
    for e in range(E):
        res[f"layers.0.experts.experts.{e}.gate_up_proj"] = Res(2 * I, H, g_idx=True)
        res[f"layers.0.experts.experts.{e}.down_proj"] = Res(H, I)
    m._qt_results = res
    m._qt_meta = {"weights": {"num_bits": 4}, "format": "pack-quantized", "ignore": []}
    sq._save_compressed(m, str(tmp_path))
    sd = load_state(str(tmp_path))
    assert not any(".experts.experts." in k or "gate_up_proj" in k for k in sd), sorted(sd)
    for e in range(E):
        r = res[f"layers.0.experts.experts.{e}.gate_up_proj"]
        p = f"layers.0.experts.{e}"
        assert torch.equal(sd[f"{p}.gate_proj.weight_packed"], r.weight_packed[:I])
        assert torch.equal(sd[f"{p}.up_proj.weight_packed"], r.weight_packed[I:])
        assert torch.equal(sd[f"{p}.gate_proj.weight_scale"], r.weight_scale[:I])
        assert torch.equal(sd[f"{p}.up_proj.weight_zero_point"], r.weight_zero_point[I:])
        assert sd[f"{p}.gate_proj.weight_shape"].tolist() == [I, H] == sd[f"{p}.up_proj.weight_shape"].tolist()
        assert torch.equal(sd[f"{p}.gate_proj.weight_g_idx"], r.weight_g_idx)
        assert torch.equal(sd[f"{p}.up_proj.weight_g_idx"], r.weight_g_idx)
        d = res[f"layers.0.experts.experts.{e}.down_proj"]
        assert torch.equal(sd[f"{p}.down_proj.weight_packed"], d.weight_packed)
        assert sd[f"{p}.down_proj.weight_shape"].tolist() == [H, I]
        assert f"{p}.down_proj.weight_g_idx" not in sd
    assert "layers.0.norm.weight" in sd


def test_dense_save_splits_the_views_of_the_fused_storage(tmp_path):
    m, fused = _model()
    m._qt_results, m._qt_meta = {}, {}
    sq._save_compressed(m, str(tmp_path), save_compressed=False)
    sd = load_state(str(tmp_path))
    gu, dn = fused["layers.0.experts.gate_up_proj"], fused["layers.0.experts.down_proj"]
    for e in range(E):
        assert torch.equal(sd[f"layers.0.experts.{e}.gate_proj.weight"], gu[e, :I])
        assert torch.equal(sd[f"layers.0.experts.{e}.up_proj.weight"], gu[e, I:])
        assert torch.equal(sd[f"layers.0.experts.{e}.down_proj.weight"], dn[e])
    assert not any("gate_up_proj" in k for k in sd)


def test_result_store_releases_detail_beyond_its_budget(monkeypatch):
    """The sequential driver keeps a result's integer levels (and AWQ's rescaled weight) only while the total stays
    under RESULT_DETAIL_BYTES; later results keep their checkpoint tensors and `dequantized()` says why it cannot."""
    import types

    import torch

    from quantool_amd.engine import sequential
    from quantool_amd.engine.gptq_linear import GPTQResult

    monkeypatch.setattr(sequential, "RESULT_DETAIL_BYTES", 3000)
    store = sequential._ResultStore()

    def result(n):
        z = torch.zeros(1)
        return GPTQResult(weight_packed=torch.zeros(4, dtype=torch.int32), weight_q=None, weight_scale=z, weight_zero_point=None,
                          weight_g_idx=None, weight_shape=torch.tensor([1, 8]), loss=z, info=z, scale_f32=z, zp_f32=z,
                          Qt=torch.zeros(n, dtype=torch.int8), col_src=None, g_of_col=z)

    store["a"] = result(2000)
    store.update({"b": result(900)})
    store["c"] = result(200)                         # 2000 + 900 + 200 > 3000
    store["d"] = types.SimpleNamespace(Qt=torch.zeros(50, dtype=torch.int8), scaled_weight=torch.zeros(10))   # fits: 2900 + 90 <= 3000
    assert store["a"].Qt is not None and store["b"].Qt is not None and store["c"].Qt is None
    assert store["d"].Qt is not None and store.detail_bytes == 2990
    assert store["c"].weight_packed is not None
    import pytest

    with pytest.raises(RuntimeError, match="released"):
        store["c"].dequantized()


def test_mixtral_vocabulary_matches_what_transformers_converts_from(tmp_path):
    """``config.model_type == "mixtral"``: the per-expert names are the legacy Mixtral checkpoint's
    (``block_sparse_moe.experts.{e}.w1 / w3 / w2``), checked against the source patterns of the installed transformers'
    own conversion table for that architecture; ``ignore`` entries of the saved quantization_config are renamed in the
    same pass."""
    import json
    import re

    m = nn.Module()
    m.model = nn.Module()
    blk = nn.Module()
    blk.mlp = nn.Module()
    blk.mlp.experts = FusedExperts()
    blk.mlp.gate = nn.Linear(H, E, bias=False)
    blk.self_attn = nn.Linear(H, H, bias=False)
    m.model.layers = nn.ModuleList([blk])
    m.config = types.SimpleNamespace(to_dict=lambda: {"model_type": "mixtral", "architectures": ["MixtralForCausalLM"]})
    assert sq.unfuse_expert_banks(m) == 1
    res = {}
    for e in range(E):
        res[f"model.layers.0.mlp.experts.experts.{e}.gate_up_proj"] = Res(2 * I, H)
        res[f"model.layers.0.mlp.experts.experts.{e}.down_proj"] = Res(H, I)
    m._qt_results = res
    m._qt_meta = {"weights": {"num_bits": 4}, "format": "pack-quantized", "ignore": ["lm_head", "model.layers.0.mlp.gate"]}
    sq._save_compressed(m, str(tmp_path))
    sd = load_state(str(tmp_path))
    p = "model.layers.0.block_sparse_moe"
    for e in range(E):
        r = res[f"model.layers.0.mlp.experts.experts.{e}.gate_up_proj"]
        assert torch.equal(sd[f"{p}.experts.{e}.w1.weight_packed"], r.weight_packed[:I])
        assert torch.equal(sd[f"{p}.experts.{e}.w3.weight_packed"], r.weight_packed[I:])
        assert torch.equal(sd[f"{p}.experts.{e}.w2.weight_packed"],
                           res[f"model.layers.0.mlp.experts.experts.{e}.down_proj"].weight_packed)
    assert f"{p}.gate.weight" in sd and "model.layers.0.self_attn.weight" in sd
    assert not any(".mlp." in k or "gate_proj" in k or "up_proj" in k for k in sd), sorted(sd)
    cfg = json.loads((tmp_path / "config.json").read_text())
    assert cfg["quantization_config"]["ignore"] == ["lm_head", "model.layers.0.block_sparse_moe.gate"]
    # the installed transformers converts FROM exactly these names (source patterns of its "mixtral" entry)
    conv = __import__("pytest").importorskip("transformers.conversion_mapping")
    table = getattr(conv, "_build_checkpoint_conversion_mapping", None)
    if table is None:
        return
    pats = []
    for entry in table().get("mixtral", []):
        src = getattr(entry, "source_patterns", None)
        pats += [src] if isinstance(src, str) else list(src or [])
    assert any("block_sparse_moe" in q for q in pats)
    for leaf in ("w1", "w2", "w3"):
        assert f".experts.*.{leaf}.weight" in pats                                   # what the loader collects per expert ...
        rx = re.compile(rf"\.block_sparse_moe\.experts\.\d+\.{leaf}\.weight_packed$")        # ... and what was written
        assert sum(bool(rx.search(k)) for k in sd) == E, sorted(sd)[:6]
