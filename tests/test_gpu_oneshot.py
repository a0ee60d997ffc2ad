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


def test_gram_stream_and_group_stream_routes_give_the_single_stream_result(dev):
    """`_oneshot_linears` puts the Gram sum of a group whose activations arrive in long batches (>= 16 384 rows) on
    the shared Gram stream and its chain on a group stream behind an event; groups fed by short batches keep both on
    their group stream (engine/streams.py).  Whatever the route, the results equal a plain one-stream run bit for
    bit -- including a SmoothQuant stage, whose rescaled weights cross from the Gram stream to the chain's."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.modifiers import GPTQModifier, SmoothQuantModifier
    from quantool_amd.engine.oneshot import LinearCalibrationSet, LinearGroup, oneshot
    from quantool_amd.engine.schemes import QuantArgs

    torch.manual_seed(3)
    n_long = HessianAccumulator.DIRECT_TOKENS
    shapes = {"a": (512, [("a.q", 96), ("a.k", 32)], [n_long]),              # one long batch: Gram stream
              "b": (256, [("b.up", 160)], [n_long, n_long]),                 # two long batches: Gram stream
              "c": (384, [("c.o", 64)], [64] * 6),                           # short batches: staged, group stream
              "d": (1024, [("d.down", 48)], [n_long])}
    groups, ref = [], {}
    for gname, (K, lins, batches) in shapes.items():
        Xs = [torch.randn(n, K, device=dev).to(torch.bfloat16) for n in batches]
        Ws = {n: (torch.randn(r, K, device=dev) * 0.02).to(torch.bfloat16) for n, r in lins}
        groups.append(LinearGroup(gname, Xs if len(Xs) > 1 else Xs[0], Ws, num_samples=len(batches)))
        acc = HessianAccumulator(K, dev)
        for x in Xs:
            acc.add(x, num_samples=1)
        res = gptq_quantize_shared([Ws[n] for n, _ in lins], acc, QuantArgs(actorder="static"))
        for (n, _), r in zip(lins, res):
            ref[n] = (r.weight_packed.clone(), r.weight_scale.clone())
    torch.cuda.synchronize()
    out = oneshot(model=LinearCalibrationSet(groups), recipe=GPTQModifier(scheme="W4A16"))
    torch.cuda.synchronize()
    for n, (packed, scale) in ref.items():
        r = out.results[n]
        assert torch.equal(r.weight_packed, packed) and torch.equal(r.weight_scale, scale), n
    # twice in a row (streams and workspaces are cached per process) and with a smoothing stage on a long-batch group
    K = 512
    X = torch.randn(n_long, K, device=dev).to(torch.bfloat16)
    W = {"s.q": (torch.randn(64, K, device=dev) * 0.02).to(torch.bfloat16)}
    v = torch.rand(K, device=dev).to(torch.bfloat16) + 0.5
    runs = []
    for _ in range(2):
        g = LinearGroup("s", X, {k: w.clone() for k, w in W.items()}, smooth_vectors={"s.norm": v.clone()}, num_samples=4)
        o = oneshot(model=LinearCalibrationSet([g] + groups[:1]), recipe=[SmoothQuantModifier(smoothing_strength=0.5),
                                                                        GPTQModifier(scheme="W4A16")])
        torch.cuda.synchronize()
        runs.append((o.results["s.q"].weight_packed.clone(), o.results["s.q"].weight_scale.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
