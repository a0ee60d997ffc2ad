"""``method=awq`` on an nn.Module (tiny random-init Llama, no download): the pseudo-quantise kernel
against the oracle, smoothing as a function-preserving rewrite, the multi-balance search against an
independent restatement, and the plugin end to end."""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.util import bf16_tensor_to_bits, synth_weight

pytestmark = pytest.mark.gpu


def _tiny_llama(dev, kv_heads=4):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=kv_heads, vocab_size=512, max_position_embeddings=128,
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev)


def _data(n=6, length=40, seed=3):
    g = torch.Generator().manual_seed(seed)
    return [{"input_ids": torch.randint(0, 512, (length,), generator=g)} for _ in range(n)]


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pseudo_quantize_kernel_matches_oracle(dev, oracle, symmetric, dtype):
    from quantool_amd.hip import ops

    R, K, gs = 48, 512, 128
    W = synth_weight(R, K, seed=11)
    s = np.random.default_rng(5).uniform(0.3, 3.0, K).astype(np.float32)
    if dtype == "bf16":
        W = oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(W))
    want = (oracle.awq_pseudo_quantize((W * s[None, :]).astype(np.float32), gs, symmetric, 4) / s[None, :]).astype(np.float32)
    Wt = torch.from_numpy(W).to(dev)
    if dtype == "bf16":
        Wt = Wt.to(torch.bfloat16)
    got = ops.awq_pseudo_quantize(Wt, torch.from_numpy(s).to(dev), gs, symmetric, 4)
    torch.cuda.synchronize()
    if dtype == "bf16":
        np.testing.assert_array_equal(bf16_tensor_to_bits(got), oracle.f32_to_bf16_bits(want))   # bit-exact
    else:
        np.testing.assert_array_equal(got.cpu().numpy(), want)                                   # bit-exact
    # in place (out aliases W) gives the same bytes
    W2 = Wt.clone()
    ops.awq_pseudo_quantize(W2, torch.from_numpy(s).to(dev), gs, symmetric, 4, out=W2)
    torch.cuda.synchronize()
    assert torch.equal(W2, got)


def _first(o):
    return o[0] if isinstance(o, (tuple, list)) else o


def _layer_inputs(model, data, dev):
    """(args, kwargs) of decoder layer 0 for every calibration row."""
    calls = []

    class Stop(Exception):
        pass

    def grab(_m, a, kw):
        calls.append((a, kw))
        raise Stop

    h = model.model.layers[0].register_forward_pre_hook(grab, with_kwargs=True)
    with torch.no_grad():
        for row in data:
            try:
                model(input_ids=row["input_ids"].reshape(1, -1).to(dev), use_cache=False)
            except Stop:
                pass
    h.remove()
    return calls


def test_smoothing_preserves_the_layer_function(dev):
    from quantool_amd.engine.awq_module import _apply, normalise_mappings, resolve_mappings
    from quantool_amd.engine.modifiers import AWQModifier

    model = _tiny_llama(dev)
    calls = _layer_inputs(model, _data(2), dev)
    layer = model.model.layers[0]
    with torch.no_grad():
        before = [_first(layer(*a, **kw)).float() for a, kw in calls]
        maps = resolve_mappings(layer, "model.layers.0", normalise_mappings(None), AWQModifier().wants)
        assert len(maps) == 4
        g = torch.Generator().manual_seed(9)
        for mp in maps:
            K = mp.balance[0].in_features
            s = (0.5 + 1.5 * torch.rand(K, generator=g)).to(dev)
            w_before = mp.balance[0].weight.data.float().clone()
            _apply(mp, s)
            torch.testing.assert_close(mp.balance[0].weight.data.float(), (w_before * s).to(torch.bfloat16).float())
        after = [_first(layer(*a, **kw)).float() for a, kw in calls]
    for b, a in zip(before, after):
        rel = (a - b).norm() / b.norm()
        assert rel < 2e-2, rel              # only bf16 re-rounding of the rescaled weights


def test_multi_balance_search_matches_independent_restatement(dev, oracle):
    """input_layernorm -> {q,k,v}: parent = self_attn.  Restated with the oracle's scale / pseudo-quant
    formulas and torch's own forward of the parent; the device path must pick the same grid point
    with the same losses."""
    from quantool_amd.engine.awq_module import awq_layer
    from quantool_amd.engine.modifiers import AWQModifier

    data = _data(4)
    model, ref = _tiny_llama(dev), _tiny_llama(dev)
    calls = _layer_inputs(ref, data, dev)
    layer = ref.model.layers[0]
    attn = layer.self_attn
    xs, parent_calls = [], []
    h1 = attn.q_proj.register_forward_pre_hook(lambda m, a: xs.append(a[0].reshape(-1, a[0].shape[-1]).clone()))
    h2 = attn.register_forward_pre_hook(lambda m, a, kw: parent_calls.append((a, kw)), with_kwargs=True)
    with torch.no_grad():
        for a, kw in calls:
            layer(*a, **kw)
    h1.remove(), h2.remove()
    X = torch.cat(xs).float().cpu().numpy()
    x_mean = np.abs(X).astype(np.float64).mean(axis=0).astype(np.float32)
    lins = [attn.q_proj, attn.k_proj, attn.v_proj]
    Ws = [l.weight.data.float().cpu().numpy() for l in lins]
    w_mean = oracle.awq_weight_mean(Ws, 128)
    with torch.no_grad():
        fp = [_first(attn(*a, **kw)).float() for a, kw in parent_calls]
        want_losses = []
        for gi in range(20):
            s = oracle.awq_scales_for_ratio(x_mean, w_mean, gi / 20)
            for l, W in zip(lins, Ws):
                trial = (oracle.awq_pseudo_quantize((W * s[None, :]).astype(np.float32), 128, True, 4) / s[None, :])
                l.weight.data.copy_(torch.from_numpy(trial.astype(np.float32)).to(dev).to(torch.bfloat16))
            sq = sum(float((r - _first(attn(*a, **kw)).float()).pow(2).sum()) for (a, kw), r in zip(parent_calls, fp))
            want_losses.append(sq / sum(r.numel() for r in fp))
    want_best = int(np.argmin(want_losses))

    with torch.no_grad():
        res = awq_layer(model.model.layers[0], "model.layers.0", _layer_inputs(model, data, dev), AWQModifier(), dev)
    torch.cuda.synchronize()
    r = res["model.layers.0.self_attn.q_proj"]
    got_losses = r.losses.cpu().numpy()
    # scales differ by powf rounding (1e-5), which can flip single roundings in the trial weights:
    # losses agree to a fraction of a percent, the argmin exactly unless two grid points tie that closely
    np.testing.assert_allclose(got_losses, np.array(want_losses, np.float32), rtol=2e-2)
    assert int(r.best_ratio_idx.item()) == want_best or \
        abs(want_losses[int(r.best_ratio_idx.item())] - want_losses[want_best]) < 2e-2 * want_losses[want_best]
    np.testing.assert_allclose(r.smoothing_scales.cpu().numpy(),
                               oracle.awq_scales_for_ratio(x_mean, w_mean, int(r.best_ratio_idx.item()) / 20), rtol=1e-4)
    # q, k and v share one mapping: same scale vector object
    assert res["model.layers.0.self_attn.k_proj"].smoothing_scales is r.smoothing_scales
    # single-balance mappings went through the Gram loss and also carry 20 losses
    assert res["model.layers.0.mlp.down_proj"].losses.shape == (20,)
    assert res["model.layers.0.self_attn.o_proj"].smoothing_scales is not None       # MHA: v -> o applies


@pytest.mark.parametrize("kv_heads", [4, 2])
def test_awq_plugin_on_tiny_llama(dev, oracle, tmp_path, monkeypatch, kv_heads):
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry
    from safetensors.torch import load_file

    monkeypatch.chdir(tmp_path)
    model = _tiny_llama(dev, kv_heads)
    x = torch.randint(0, 512, (1, 32), device=dev)
    with torch.no_grad():
        before = model(input_ids=x).logits.float()
    q = QuantizerRegistry.create("awq", model_id="synthetic/tiny-llama")
    out = q.quantize(model=model, level="W4A16", dataset=_data(6), num_calibration_samples=6, max_seq_length=64)
    torch.cuda.synchronize()
    assert q.last_model is model
    q.save_pretrained(str(tmp_path / "saved"))
    sd = load_file(str(tmp_path / "saved" / "model.safetensors"))
    assert sum(1 for k in sd if k.endswith("weight_packed")) == 2 * 7
    assert "lm_head.weight" in sd and Path(out).is_dir()
    res = model._qt_results
    o = res["model.layers.0.self_attn.o_proj"]
    assert (o.smoothing_scales is None) == (kv_heads == 2)          # v -> o skipped under GQA, o_proj still RTN'd
    for name, r in res.items():
        lin = model.get_submodule(name)
        # module weight == dequantised levels, levels == unpacked words
        np.testing.assert_array_equal(
            lin.weight.data.float().cpu().numpy(),
            oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(r.dequantized().cpu().numpy())))
        np.testing.assert_array_equal(oracle.unpack_int4(r.weight_packed.cpu().numpy(), lin.in_features),
                                      r.Qt.t().cpu().numpy())
        if r.smoothing_scales is not None:
            assert int(torch.argmin(r.losses)) == int(r.best_ratio_idx)
            s = r.smoothing_scales
            assert torch.isfinite(s).all() and (s > 0).all()
    with torch.no_grad():
        after = model(input_ids=x).logits.float()
    rel = (after - before).norm() / before.norm()
    assert rel < 0.35, rel                                          # int4 g128 on a random-init model
