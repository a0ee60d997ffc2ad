"""Config 5 through the nn.Module path: a tiny random-init Mixtral (transformers >= 5 keeps the experts
as fused 3-d parameters ``gate_up_proj [E, 2I, H]`` / ``down_proj [E, H, I]``, which a Linear-only walk
would silently leave dense).  The driver unfuses every expert bank into per-expert Linears that
view the fused storage, so each expert is calibrated on its own routed tokens."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny_mixtral(dev):
    from transformers import MixtralConfig, MixtralForCausalLM

    cfg = MixtralConfig(hidden_size=256, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2, vocab_size=512,
                        max_position_embeddings=128, tie_word_embeddings=False)
    torch.manual_seed(0)
    return MixtralForCausalLM(cfg).to(torch.bfloat16).to(dev).eval()


def test_unfused_experts_keep_the_function_and_the_storage(dev):
    from quantool_amd.engine.sequential import find_decoder_layers, uncovered_weight_fraction, unfuse_expert_banks

    m = _tiny_mixtral(dev)
    x = torch.randint(0, 512, (1, 24), device=dev)
    with torch.no_grad():
        y0 = m(input_ids=x).logits.clone()
    fused = m.model.layers[0].mlp.experts.gate_up_proj
    assert unfuse_expert_banks(m) == 2
    with torch.no_grad():
        y1 = m(input_ids=x).logits
    # same routing, same weights; the fused module may run its experts through another GEMM kernel
    # (grouped / batched), so bf16 logits agree to rounding, not bit for bit (they do on the CPU in fp32)
    assert torch.allclose(y0.float(), y1.float(), rtol=3e-2, atol=3e-2)
    lin = m.model.layers[0].mlp.experts.experts[1].gate_up_proj
    assert lin.weight.untyped_storage().data_ptr() == fused.untyped_storage().data_ptr()     # a view, not a copy
    layers = find_decoder_layers(m)
    assert len(layers) == 2 and type(layers[0]).__name__ == "MixtralDecoderLayer"        # not the 4-expert list
    linears = {n: mod for n, mod in layers[0].named_modules() if isinstance(mod, torch.nn.Linear)}
    assert len(linears) == 4 + 2 * 4 and uncovered_weight_fraction(layers[0], linears) < 0.02   # the router only


@pytest.mark.parametrize("calib_mode", ["merged", "per-sample"])
def test_gptq_plugin_quantises_every_expert_on_its_routed_tokens(dev, oracle, tmp_path, monkeypatch, calib_mode):
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry
    from quantool_amd.engine import sequential
    from quantool_amd.engine.sequential import unfuse_expert_banks

    from tests.util import hook_inputs, oracle_group

    monkeypatch.chdir(tmp_path)
    model, ref = _tiny_mixtral(dev), _tiny_mixtral(dev)
    unfuse_expert_banks(ref)
    g = torch.Generator().manual_seed(3)
    data = [{"input_ids": torch.randint(0, 512, (64,), generator=g)} for _ in range(8)]
    monkeypatch.setattr(sequential, "DEBUG_KEEP", {})
    # "merged" = the default mode (the 8 equal-length rows share one forward per layer; the plain hook sees the same
    # stacked forward, hence the same routing); "per-sample" = one sample per forward, the reference's calling pattern
    batched = calib_mode == "merged"
    if batched:
        monkeypatch.delenv("QT_CALIB_BATCH_TOKENS", raising=False)
    else:
        monkeypatch.setenv("QT_CALIB_BATCH_TOKENS", "0")
    q = QuantizerRegistry.create("gptq", model_id="synthetic/tiny-mixtral")
    q.quantize(model=model, level="W4A16", dataset=data, num_calibration_samples=8, max_seq_length=64,
               shuffle_calibration_samples=False)
    torch.cuda.synchronize()
    res, keep = model._qt_results, sequential.DEBUG_KEEP
    assert len(res) == 2 * (4 + 2 * 4)
    pre = "model.layers.0.mlp.experts.experts."
    counts = []
    for e in range(4):
        for which in ("gate_up_proj", "down_proj"):
            name = f"{pre}{e}.{which}"
            k = keep[name]
            mod = ref.model.layers[0].mlp.experts.experts[e].get_submodule(which)
            acts = hook_inputs(ref, mod, data, dev, batched=batched)
            assert sum(a.shape[0] for a in acts) > 0
            counts.append(sum(a.shape[0] for a in acts))
            # the Hessian's sample count is the number of SAMPLES that routed a token to this expert -- what one
            # forward per sample counts (upstream's num_added), also when the samples shared a forward
            per_sample = hook_inputs(ref, mod, data, dev) if batched else acts
            assert k["n"] == sum(1 for a in per_sample if a.shape[0] > 0), name
            (o,) = oracle_group(oracle, acts, [mod.weight.data], k)
            np.testing.assert_array_equal(res[name].weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]), err_msg=name)
    assert len(set(counts)) > 1                      # ragged: the experts saw different token counts
    # the fused parameter now holds the dequantised experts
    fused = model.model.layers[0].mlp.experts.experts[2].gate_up_proj.weight
    assert torch.equal(fused.data, res[f"{pre}2.gate_up_proj"].dequantized(torch.bfloat16))


def test_smoothquant_plugin_on_tiny_mixtral(dev, tmp_path, monkeypatch):
    """BASELINE config 5's recipe (SmoothQuant + GPTQ, W8A8) on the nn.Module path of a sparse-MoE model:
    the attention inputs are smoothed against the input norm (norm / s, W * s), every expert Linear is
    quantised to int8 channel-wise on its routed tokens, and the function stays close."""
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry

    monkeypatch.chdir(tmp_path)
    model = _tiny_mixtral(dev)
    x = torch.randint(0, 512, (1, 32), device=dev)
    with torch.no_grad():
        before = model(input_ids=x).logits.float()
    norm_before = model.model.layers[0].input_layernorm.weight.data.clone()
    g = torch.Generator().manual_seed(4)
    data = [{"input_ids": torch.randint(0, 512, (64,), generator=g)} for _ in range(8)]
    q = QuantizerRegistry.create("smoothquant", model_id="synthetic/tiny-mixtral")
    q.quantize(model=model, level="W8A8", dataset=data, num_calibration_samples=8, max_seq_length=64)
    torch.cuda.synchronize()
    assert not torch.equal(model.model.layers[0].input_layernorm.weight.data, norm_before)      # norm /= s
    res = model._qt_results
    assert len(res) == 2 * (4 + 2 * 4)
    r = res["model.layers.1.mlp.experts.experts.3.down_proj"]
    assert r.weight_packed is None and r.weight_q.dtype == torch.int8 and r.weight_scale.shape == (256, 1)
    with torch.no_grad():
        after = model(input_ids=x).logits.float()
    assert float((after - before).norm() / before.norm()) < 0.1
