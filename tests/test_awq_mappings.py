"""AWQ mapping resolver (host logic, no GPU): which (smooth layer -> balance Linears, parent) triples
the default Llama-family patterns give inside one decoder layer."""
import pytest
import torch

from quantool_amd.engine.awq_module import DEFAULT_MAPPINGS, normalise_mappings, resolve_mappings
from quantool_amd.engine.modifiers import AWQModifier


def _layer(kv_heads):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4,
                      num_key_value_heads=kv_heads, vocab_size=64, max_position_embeddings=32)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).model.layers[0]


def test_default_mappings_on_mha_layer():
    layer = _layer(kv_heads=4)
    mod = AWQModifier()
    got = resolve_mappings(layer, "model.layers.0", normalise_mappings(None), mod.wants)
    assert [(m.smooth_name.split(".")[-1], [b.split(".")[-1] for b in m.balance_names], m.parent_name) for m in got] == [
        ("input_layernorm", ["q_proj", "k_proj", "v_proj"], "model.layers.0.self_attn"),
        ("v_proj", ["o_proj"], "model.layers.0.self_attn.o_proj"),
        ("post_attention_layernorm", ["gate_proj", "up_proj"], "model.layers.0.mlp"),
        ("up_proj", ["down_proj"], "model.layers.0.mlp.down_proj"),
    ]
    assert [m.single for m in got] == [False, True, False, True]
    assert got[0].parent is layer.self_attn and got[3].parent is layer.mlp.down_proj


def test_v_to_o_is_skipped_under_gqa():
    layer = _layer(kv_heads=2)          # v_proj: 64 outputs, o_proj: 128 inputs
    got = resolve_mappings(layer, "model.layers.0", normalise_mappings(None), AWQModifier().wants)
    assert [m.smooth_name.split(".")[-1] for m in got] == ["input_layernorm", "post_attention_layernorm", "up_proj"]


def test_ignored_balance_layer_drops_its_mapping():
    layer = _layer(kv_heads=4)
    mod = AWQModifier(ignore=["lm_head", "re:.*down_proj$"])
    got = resolve_mappings(layer, "model.layers.0", normalise_mappings(None), mod.wants)
    assert "up_proj" not in [m.smooth_name.split(".")[-1] for m in got] and len(got) == 3


@pytest.mark.parametrize("spec", [
    [["re:.*post_attention_layernorm$", ["re:.*gate_proj$", "re:.*up_proj$"]]],
    [{"smooth_layer": "re:.*post_attention_layernorm$", "balance_layers": ["re:.*gate_proj$", "re:.*up_proj$"]}],
])
def test_user_mappings_forms(spec):
    layer = _layer(kv_heads=4)
    norm = normalise_mappings(spec)
    assert norm == [("re:.*post_attention_layernorm$", ["re:.*gate_proj$", "re:.*up_proj$"])]
    got = resolve_mappings(layer, "model.layers.0", norm, AWQModifier(mappings=spec).wants)
    assert len(got) == 1 and got[0].parent is layer.mlp


def test_default_table_is_the_llama_family_one():
    assert [s for s, _ in DEFAULT_MAPPINGS] == ["re:.*input_layernorm$", "re:.*v_proj$",
                                                "re:.*post_attention_layernorm$", "re:.*up_proj$"]
