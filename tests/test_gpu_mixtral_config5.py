"""BASELINE config 5 at reduced size: Mixtral-shaped layer, SmoothQuant + GPTQ int4 with the GPTQ stage
covering the expert Linears.  Tokens reach an expert through a seeded top-2 router, so every expert
group has its own (ragged) token count.  Attention q/k/v are a SmoothQuant mapping (their producer is
the input norm); experts are not smoothed.  Through the ``smoothquant`` plugin."""
import numpy as np
import pytest
import torch

from tests.util import bf16_tensor_to_bits, bits_to_bf16_tensor, synth_activations, synth_weight

pytestmark = pytest.mark.gpu

HID, INTER, KV, E, N_TOK = 256, 512, 128, 4, 1536


def _layer(oracle, dev):
    from quantool_amd.engine.oneshot import LinearCalibrationSet, LinearGroup

    bf = lambda a: bits_to_bf16_tensor(oracle.f32_to_bf16_bits(a), dev)      # noqa: E731
    x_attn = synth_activations(N_TOK, HID, seed=21)                           # bf16 bits [N, HID]
    x_moe = synth_activations(N_TOK, HID, seed=22)
    rng = np.random.default_rng(7)
    logits = rng.standard_normal((N_TOK, E))
    top2 = np.argsort(-logits, axis=1)[:, :2]
    route = [np.nonzero((top2 == e).any(axis=1))[0] for e in range(E)]        # token ids per expert
    assert sum(len(r) for r in route) == 2 * N_TOK and len({len(r) for r in route}) > 1    # ragged
    norm_w = (1.0 + 0.1 * rng.standard_normal(HID)).astype(np.float32)
    raw = {"attn_in": (x_attn, {"self_attn.q_proj": synth_weight(HID, HID, 31), "self_attn.k_proj": synth_weight(KV, HID, 32),
                                "self_attn.v_proj": synth_weight(KV, HID, 33)})}
    groups = [LinearGroup("attn_in", bits_to_bf16_tensor(x_attn, dev), {n: bf(w) for n, w in raw["attn_in"][1].items()},
                          smooth_vectors={"input_layernorm.weight": bf(norm_w)})]
    for e in range(E):
        xe = x_moe[route[e]]
        he = synth_activations(len(route[e]), INTER, seed=40 + e)
        w_in = {f"block_sparse_moe.experts.{e}.w1": synth_weight(INTER, HID, 50 + e),
                f"block_sparse_moe.experts.{e}.w3": synth_weight(INTER, HID, 60 + e)}
        w_dn = {f"block_sparse_moe.experts.{e}.w2": synth_weight(HID, INTER, 70 + e)}
        raw[f"expert{e}_in"] = (xe, w_in)
        raw[f"expert{e}_down"] = (he, w_dn)
        groups.append(LinearGroup(f"expert{e}_in", bits_to_bf16_tensor(xe, dev), {n: bf(w) for n, w in w_in.items()}))
        groups.append(LinearGroup(f"expert{e}_down", bits_to_bf16_tensor(he, dev), {n: bf(w) for n, w in w_dn.items()}))
    return LinearCalibrationSet(groups, model_name="mixtral-shaped"), raw, norm_w, route


def test_smoothquant_gptq_on_mixtral_shaped_layer(dev, oracle, tmp_path, monkeypatch):
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    monkeypatch.chdir(tmp_path)
    cal, raw, norm_w, route = _layer(oracle, dev)
    q = QuantizerRegistry.create("smoothquant", model_id="synthetic/mixtral-shaped")
    q.quantize(model=cal, level="W4A8", dataset=cal, method_kwargs__smoothing_strength=0.5)
    torch.cuda.synchronize()
    ql = q.last_model
    assert len(ql.results) == 3 + 3 * E and set(ql.smoothing_scales) == {"attn_in"}       # experts not smoothed

    # ---- smoothing stage vs the oracle: s, W * s, norm / s ----
    xb, ws = raw["attn_in"]
    wf = {n: oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(w)) for n, w in ws.items()}
    amin, amax = oracle.channel_minmax(xb)
    s_want = oracle.smoothquant_scales(amin, amax, list(wf.values()), 0.5)
    s = ql.smoothing_scales["attn_in"].cpu().numpy()
    np.testing.assert_allclose(s, s_want, rtol=1e-6)                                        # powf rounding only
    nw = oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(norm_w))
    np.testing.assert_array_equal(bf16_tensor_to_bits(ql.smoothed["input_layernorm.weight"]),
                                  oracle.f32_to_bf16_bits((nw / s).astype(np.float32)))
    assert "input_layernorm.weight" in ql.state_dict()

    # ---- GPTQ stage: every group equals a stand-alone run on its own (routed, smoothed) inputs ----
    qa = QuantArgs(num_bits=4, symmetric=True, group_size=128, actorder="static")
    for gname, (xg, wg) in raw.items():
        X = oracle.bf16_bits_to_f32(xg)
        Wl = [oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(w)) for w in wg.values()]
        if gname == "attn_in":
            X = oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits((X / s[None, :]).astype(np.float32)))
            Wl = [oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits((w * s[None, :]).astype(np.float32))) for w in Wl]
        acc = HessianAccumulator(X.shape[1], dev)
        acc.add(torch.from_numpy(X).to(dev).to(torch.bfloat16))
        want = gptq_quantize_shared([torch.from_numpy(w).to(dev).to(torch.bfloat16) for w in Wl], acc, qa)
        for name, w_res in zip(wg, want):
            got = ql.results[name]
            assert torch.equal(got.weight_packed, w_res.weight_packed), name
            assert torch.equal(got.weight_scale, w_res.weight_scale), name
    # ragged expert token counts really differed
    assert len({raw[f"expert{e}_in"][0].shape[0] for e in range(E)}) > 1
    # one expert against the oracle proper (given the device's factor), as the per-Linear tests do
    xg, wg = raw["expert1_down"]
    keep = {}
    acc = HessianAccumulator(INTER, dev)
    acc.add(bits_to_bf16_tensor(xg, dev))
    w = oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(next(iter(wg.values()))))
    gptq_quantize_shared([torch.from_numpy(w).to(dev).to(torch.bfloat16)], acc, qa, keep=keep)
    Gl = np.tril(acc.G.cpu().numpy())
    H = oracle.hessian_from_gram_f32(Gl + np.tril(Gl, -1).T, acc.n)
    o = oracle.quantize_weight(w, H, group_size=128, symmetric=True, num_bits=4, actorder="static",
                               U_override=keep["U"].cpu().numpy())
    np.testing.assert_array_equal(ql.results["block_sparse_moe.experts.1.w2"].weight_packed.cpu().numpy(),
                                  oracle.pack_int4(o["q"]))
    # W4A8: the activation block is configuration only (dynamic per-token)
    a = ql.quantization_config()["config_groups"]["group_0"]["input_activations"]
    assert a["num_bits"] == 8 and a["dynamic"] is True
