"""The plugin -> oneshot -> HIP path end to end on the GPU, plus the committed golden fixtures
replayed through the C ABI."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.util import bits_to_bf16_tensor

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _load(name):
    with np.load(GOLD / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", sorted(p.stem for p in GOLD.glob("gptq_*.npz")))
def test_golden_fixture_through_hip_path(dev, oracle, name):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    g = _load(name)
    ao = str(g["actorder"])
    ao = None if ao == "None" else ao
    S = int(g["n_samples"])
    K = g["W"].shape[1]
    X = bits_to_bf16_tensor(g["X_bf16"], dev).reshape(S, -1, K)
    acc = HessianAccumulator(K, dev)
    acc.add(X)
    W = bits_to_bf16_tensor(oracle.f32_to_bf16_bits(g["W"]), dev)   # fixture weights are bf16-exact
    r = gptq_quantize_linear(W, acc, QuantArgs(symmetric=bool(g["symmetric"]), actorder=ao))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r.scale_f32.cpu().numpy(), g["scale"])      # bit-exact
    np.testing.assert_array_equal(r.zp_f32.cpu().numpy(), g["zp"])
    q = oracle.unpack_int4(r.weight_packed.cpu().numpy(), K)
    rate = float((q != g["q"]).mean())
    assert rate < 2e-2, f"nibble mismatch rate {rate} vs fixture (independent LAPACK factor)"
    if ao == "group":
        # same permutation unless two diagonal entries tie to the last bit
        assert (r.weight_g_idx.cpu().numpy() != g["g_idx"]).mean() < 0.02


def test_plugin_quantize_on_linear_calibration_set(dev, oracle, tmp_path, monkeypatch):
    """method=gptq through the registry, on explicit (activation, weight) groups; writes the
    compressed-tensors layout and returns the output directory like the reference."""
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry
    from quantool_amd.engine.oneshot import LinearCalibrationSet, LinearGroup
    from safetensors.torch import load_file

    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    K, S, T = 256, 6, 64
    X = torch.randn(S, T, K, device=dev).to(torch.bfloat16)
    Wq = (torch.randn(96, K, device=dev) * 0.02).to(torch.bfloat16)
    Wk = (torch.randn(32, K, device=dev) * 0.02).to(torch.bfloat16)
    Wh = (torch.randn(16, K, device=dev) * 0.02).to(torch.bfloat16)
    cal = LinearCalibrationSet([LinearGroup("attn_in", X, {"layers.0.q_proj": Wq, "layers.0.k_proj": Wk,
                                                            "lm_head": Wh})])
    quantizer = QuantizerRegistry.create("gptq", model_id="synthetic/tiny")
    out = quantizer.quantize(model=cal, level="W4A16", dataset=cal, method_kwargs__dampening_frac=0.02)
    assert Path(out).is_dir()
    sd = load_file(str(Path(out) / "model.safetensors"))
    assert set(sd) == {f"layers.0.{n}.{k}" for n in ("q_proj", "k_proj")
                       for k in ("weight_packed", "weight_scale", "weight_shape")}   # lm_head ignored
    assert sd["layers.0.q_proj.weight_packed"].shape == (96, K // 8)
    assert sd["layers.0.q_proj.weight_packed"].dtype == torch.int32
    assert sd["layers.0.q_proj.weight_scale"].dtype == torch.bfloat16
    assert sd["layers.0.k_proj.weight_shape"].tolist() == [32, K]
    cfg = json.loads((Path(out) / "config.json").read_text())["quantization_config"]
    assert cfg["format"] == "pack-quantized" and cfg["ignore"] == ["lm_head"]
    w = cfg["config_groups"]["group_0"]["weights"]
    assert (w["num_bits"], w["symmetric"], w["strategy"], w["group_size"]) == (4, True, "group", 128)
    assert cfg["config_groups"]["group_0"]["input_activations"] is None        # W4A16: weight-only
    # parity of the stacked sweep: q_proj rows equal a stand-alone run's rows (rows are independent)
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    acc = HessianAccumulator(K, dev)
    acc.add(X)
    solo = gptq_quantize_linear(Wq, acc, QuantArgs(actorder="static"), dampening_frac=0.02)
    assert torch.equal(solo.weight_packed.cpu(), sd["layers.0.q_proj.weight_packed"])
    quantizer.save_pretrained(str(tmp_path / "export"))
    assert (tmp_path / "export" / "model.safetensors").exists()
