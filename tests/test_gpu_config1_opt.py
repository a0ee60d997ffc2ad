"""BASELINE config 1 on the REAL engine: an OPT-125M-shaped fp16 model (hidden 768, ffn 3072, 12
layers, biased Linears and LayerNorms, blocks under ``model.decoder.layers`` -- random init, no
download) driven with the field set of the reference's ``test_gptq_config.yaml`` (:1-37, restated
in ``tests/golden/plumbing_gptq_config.yaml``) through ``QuantizerRegistry.create("gptq")`` ->
``quantize`` -> ``save_pretrained`` on the GPU, and checked against the ORACLE on the activations a
plain torch hook sees -- not against the HIP path itself.

What the oracle comparison covers beyond the per-Linear tests: hook order and input grouping of the
sequential driver, the fp16 path end to end (the reference injects no dtype, base.py:222-241), the
un-permutation, the write-back dtype, and that layer l+1 is calibrated on the outputs of the
QUANTISED layer l.
"""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

from tests.util import hook_inputs as _hook_inputs, oracle_group as _oracle_group

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).parent / "golden"


def _opt(dev, layers=12):
    from transformers import OPTConfig, OPTForCausalLM

    cfg = OPTConfig(hidden_size=768, ffn_dim=3072, num_hidden_layers=layers, num_attention_heads=12, vocab_size=2048,
                    max_position_embeddings=256, word_embed_proj_dim=768, dtype="float16")
    torch.manual_seed(0)
    return OPTForCausalLM(cfg).to(torch.float16).to(dev).eval()


@pytest.mark.parametrize("calib_mode", ["merged", "per-sample"])
def test_opt125m_shaped_fp16_through_the_gptq_plugin(dev, oracle, tmp_path, monkeypatch, calib_mode):
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry
    from quantool_amd.engine import sequential
    from quantool_amd.engine.serialization import load_state

    monkeypatch.chdir(tmp_path)
    cfg = yaml.safe_load((GOLD / "plumbing_gptq_config.yaml").read_text())
    qcfg = cfg["quantization_config"]
    model, ref = _opt(dev), _opt(dev)                              # same seed -> same weights
    assert model.model.decoder.layers[0].fc1.weight.dtype == torch.float16
    g = torch.Generator().manual_seed(cfg["dataset_seed"])
    data = [{"input_ids": torch.randint(0, 2048, (96,), generator=g)} for _ in range(cfg["sample_size"])]

    monkeypatch.setattr(sequential, "DEBUG_KEEP", {})
    # "merged" = the default mode (the 32 equal-length rows share one forward per layer; the plain hook the oracle is
    # fed from sees the same stacked forward); "per-sample" = one sample per forward, the reference's calling pattern
    batched = calib_mode == "merged"
    if batched:
        monkeypatch.delenv("QT_CALIB_BATCH_TOKENS", raising=False)
    else:
        monkeypatch.setenv("QT_CALIB_BATCH_TOKENS", "0")
    q = QuantizerRegistry.create(cfg["method"], model_id=cfg["model_id"], **qcfg)
    out = q.quantize(model=model, level=cfg["quant_level"], dataset=data, num_calibration_samples=cfg["sample_size"],
                     max_seq_length=128, shuffle_calibration_samples=False, **qcfg)
    torch.cuda.synchronize()
    keep = sequential.DEBUG_KEEP
    assert out.endswith("gptq_synthetic_opt-125m-shaped_W4A16") and q.last_model is model
    res = model._qt_results
    assert len(res) == 12 * 6 and "lm_head" not in " ".join(res)

    # ---- layer 0: q/k/v share one input (one Hessian), out_proj, fc1, fc2 (K = 3072) have their own ----
    l0, r0 = model.model.decoder.layers[0], ref.model.decoder.layers[0]
    pre = "model.decoder.layers.0."
    by_members = {frozenset(v["names"]): v for key, v in keep.items() if key.startswith(pre)}
    assert sorted(map(len, by_members)) == [1, 1, 1, 3]            # 4 input groups, q/k/v together
    for members in (["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], ["self_attn.out_proj"], ["fc1"], ["fc2"]):
        k = by_members[frozenset(pre + s for s in members)]
        names = k["names"]                                          # the driver's own order inside the group
        sub = [n[len(pre):] for n in names]
        acts = _hook_inputs(ref, r0.get_submodule(sub[0]), data, dev, batched=batched)
        assert acts[0].dtype == torch.float16 and k["n"] == len(data)
        outs = _oracle_group(oracle, acts, [r0.get_submodule(s).weight.data for s in sub], k)
        assert np.array_equal(k["perm"].cpu().numpy(), outs[0]["perm"].astype(np.int32))
        for name, s, o in zip(names, sub, outs):
            r = res[name]
            np.testing.assert_array_equal(r.weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]), err_msg=name)
            np.testing.assert_array_equal(r.scale_f32.cpu().numpy(), o["scale"], err_msg=name)
            assert r.weight_scale.dtype == torch.float16
            # the module now holds the dequantised weight, rounded once to the checkpoint dtype
            want = torch.from_numpy(o["w_dq"]).to(torch.float16)
            assert torch.equal(l0.get_submodule(s).weight.data.cpu(), want), name
            assert torch.equal(l0.get_submodule(s).bias.data, r0.get_submodule(s).bias.data)      # biases untouched

    # ---- layer 1 is calibrated on the outputs of the quantised layer 0 ----
    r0.load_state_dict(l0.state_dict())
    r1 = ref.model.decoder.layers[1]
    pre1 = "model.decoder.layers.1."
    k1 = next(v for key, v in keep.items() if pre1 + "self_attn.q_proj" in v["names"])
    sub1 = [n[len(pre1):] for n in k1["names"]]
    acts1 = _hook_inputs(ref, r1.get_submodule(sub1[0]), data, dev, batched=batched)
    outs1 = _oracle_group(oracle, acts1, [r1.get_submodule(s).weight.data for s in sub1], k1)
    for n, o in zip(k1["names"], outs1):
        np.testing.assert_array_equal(res[n].weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]), err_msg=n)

    # ---- what save_pretrained leaves on disk ----
    sd = load_state(out)
    assert f"{pre}fc2.weight_packed" in sd and f"{pre}fc2.weight" not in sd and f"{pre}fc2.bias" in sd
    assert sd[f"{pre}fc2.weight_packed"].shape == (768, 3072 // 8) and sd[f"{pre}fc2.weight_scale"].dtype == torch.float16
    assert "lm_head.weight" in sd and "model.decoder.embed_tokens.weight" in sd
    assert torch.equal(sd[f"{pre}fc1.weight_packed"], res[pre + "fc1"].weight_packed.cpu())
    conf = json.loads((Path(out) / "config.json").read_text())
    assert conf["quantization_config"]["format"] == "pack-quantized" and conf["hidden_size"] == 768
    q.save_pretrained(str(tmp_path / "again"))
    assert (tmp_path / "again" / "model.safetensors").exists()


def test_sequential_targets_and_block_size_are_honoured_or_refused(dev, tmp_path, monkeypatch):
    """The plugin forwards block_size / dampening_frac / sequential_targets (reference gptq.py:82-84)."""
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry

    monkeypatch.chdir(tmp_path)
    g = torch.Generator().manual_seed(0)
    data = [{"input_ids": torch.randint(0, 2048, (64,), generator=g)} for _ in range(4)]
    model = _opt(dev, layers=2)
    q = QuantizerRegistry.create("gptq", model_id="synthetic/opt-tiny")
    q.quantize(model=model, level="W4A16", dataset=data, num_calibration_samples=4, max_seq_length=64,
               method_kwargs={"sequential_targets": ["OPTDecoderLayer"], "dampening_frac": 0.05})
    assert len(model._qt_results) == 12
    with pytest.raises(ValueError, match="matches no module"):
        QuantizerRegistry.create("gptq", model_id="x").quantize(
            model=_opt(dev, layers=1), level="W4A16", dataset=data, num_calibration_samples=4, max_seq_length=64,
            method_kwargs={"sequential_targets": ["LlamaDecoderLayer"]})
    with pytest.raises(ValueError, match="block_size=64"):
        QuantizerRegistry.create("gptq", model_id="x").quantize(
            model=_opt(dev, layers=1), level="W4A16", dataset=data, num_calibration_samples=4, max_seq_length=64,
            method_kwargs={"block_size": 64})
