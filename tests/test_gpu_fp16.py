"""Checkpoint-dtype fidelity: the reference passes the model through with no dtype override
(base.py:222-241) and config 1's model (OPT-125M) is an fp16 checkpoint, so every entry point that
takes activations or weights must take IEEE half as it is -- fp16 -> fp32 is exact, fp16 x fp16
products are exact in the fp32 MFMA accumulator, and outputs are rounded once (RNE) to fp16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from quantool_amd.hip import ops as _ops

    return _ops


def _x16(n, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n, K), generator=g, device=dev)
    x[:, : max(1, K // 100)] *= 10
    return x.to(torch.float16)


@pytest.mark.parametrize("n_tokens,K", [(200, 64), (1000, 384), (4096 + 33, 768), (2048, 4352)])
def test_xtx_fp16_matches_f64_gram(ops, dev, n_tokens, K):
    X = _x16(n_tokens, K, dev, n_tokens + K)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    torch.cuda.synchronize()
    x = X.double()
    want = torch.tril(x.t() @ x)
    d = torch.sqrt(torch.diag(want))
    assert bool(((torch.tril(G).double() - want).abs() <= 1e-5 * torch.tril(torch.outer(d, d)) + 1e-30).all())
    # the same VALUES as bf16 would have lost mantissa bits: the fp16 kernel must not round through bf16
    Gb = torch.zeros_like(G)
    ops.xtx_accumulate(X.to(torch.bfloat16), Gb)
    assert not torch.equal(torch.tril(G), torch.tril(Gb))


def test_act_stats_fp16_exact_minmax(ops, dev):
    X = _x16(777, 264, dev, 3)
    s = torch.zeros(264, dtype=torch.float32, device=dev)
    mn = torch.full((264,), float("inf"), device=dev)
    mx = torch.full((264,), float("-inf"), device=dev)
    ops.act_stats_accumulate(X, s, mn, mx)
    assert torch.equal(mn, X.float().min(0).values) and torch.equal(mx, X.float().max(0).values)
    torch.testing.assert_close(s.double(), X.double().abs().sum(0), rtol=1e-5, atol=0)


@pytest.mark.parametrize("actorder,sym", [("static", True), ("group", False), (None, True)])
def test_linear_fp16_bit_exact_given_gpu_factor(dev, oracle, actorder, sym):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    K, S, T = 512, 8, 160
    X = _x16(S * T, K, dev, 5).reshape(S, T, K)
    g = torch.Generator(device=dev).manual_seed(6)
    Ws = [(torch.randn((r, K), generator=g, device=dev) * 0.02).to(torch.float16) for r in (96, 40)]
    acc = HessianAccumulator(K, dev)
    for b in range(S):
        acc.add(X[b:b + 1])
    assert acc.dtype == torch.float16
    keep = {}
    res = gptq_quantize_shared(Ws, acc, QuantArgs(num_bits=4, symmetric=sym, group_size=128, actorder=actorder), keep=keep)
    torch.cuda.synchronize()
    Gl = np.tril(acc.G.cpu().numpy())
    H = oracle.hessian_from_gram_f32(Gl + np.tril(Gl, -1).T, acc.n)
    for w, r in zip(Ws, res):
        o = oracle.quantize_weight(w.float().cpu().numpy(), H, group_size=128, symmetric=sym, num_bits=4,
                                   actorder=actorder, U_override=keep["U"].cpu().numpy())
        np.testing.assert_array_equal(r.scale_f32.cpu().numpy(), o["scale"])
        np.testing.assert_array_equal(r.zp_f32.cpu().numpy(), o["zp"])
        np.testing.assert_array_equal(r.weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]))
        assert r.weight_scale.dtype == torch.float16
        # dequantised write-back in the checkpoint dtype = one RNE rounding of the fp32 value
        assert torch.equal(r.dequantized(torch.float16).cpu(), torch.from_numpy(o["w_dq"]).to(torch.float16))


def test_awq_and_smoothquant_helpers_take_fp16(ops, oracle, dev):
    g = torch.Generator(device=dev).manual_seed(1)
    W = (torch.randn((48, 256), generator=g, device=dev) * 0.05).to(torch.float16)
    s = (torch.rand(256, generator=g, device=dev) + 0.5).float()
    out = ops.scale_columns(W, s)
    assert out.dtype == torch.float16 and torch.equal(out, (W.float() * s).to(torch.float16))
    pq = ops.awq_pseudo_quantize(W, s, 128, True, 4)
    want = oracle.awq_pseudo_quantize((W.float() * s).cpu().numpy(), 128, True, 4) / s.cpu().numpy()
    assert pq.dtype == torch.float16 and torch.equal(pq.cpu(), torch.from_numpy(want.astype(np.float32)).to(torch.float16))
    wmax = torch.zeros(256, device=dev)
    ops.col_absmax_accumulate(W, wmax)
    assert torch.equal(wmax, W.float().abs().max(0).values)
    scale, zp, _, _ = ops.group_minmax_qparams(W, 128, True, 4)
    o_s, o_z = oracle.minmax_qparams(W.float().cpu().numpy(), 128, True, 4)
    np.testing.assert_array_equal(scale.cpu().numpy(), o_s)


def test_fp32_activations_keep_their_precision_by_default(dev, monkeypatch):
    """fp32 batches are staged and accumulated in fp32 (the three-plane Gram product, tests/test_gpu_fp32_activations.py);
    QT_FP32_ACTIVATIONS=bf16 is the explicit downgrade."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator

    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    acc = HessianAccumulator(64, dev)
    acc.add(torch.randn(2, 16, 64, device=dev))
    assert acc.dtype == torch.float32 and acc.n == 2
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "bf16")
    acc = HessianAccumulator(64, dev)
    acc.add(torch.randn(2, 16, 64, device=dev))
    assert acc.dtype == torch.bfloat16 and acc.n == 2
