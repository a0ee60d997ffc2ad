"""a7 for an fp32 checkpoint: upstream accumulates ``inp.float()``, i.e. an fp32 Gram product.
``qt_xtx_accumulate_f32`` runs it fp32-accurately on the bf16 MFMA (three bf16 planes per activation, six plane
products per tile in one fp32 accumulator); ``HessianAccumulator`` routes fp32 batches there by default."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _x(n, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n, K), generator=g, device=dev)
    x[:, 3] *= 30.0                      # an outlier channel
    x[:, K // 2] *= 1e-3
    return x


@pytest.mark.parametrize("n,K", [(128, 256), (300, 512), (1000, 776), (9000, 1024), (20000, 260)])
def test_f32_gram_is_fp32_accurate(ops, dev, n, K):
    """Against the fp64 Gram of the SAME fp32 activations: the bar of the 16-bit path (1e-5 of sqrt(G_ii G_jj)) --
    and more than 20 times closer than what rounding the activations to bf16 first would give.  Shapes: one
    token chunk, ragged tokens, K not a multiple of 256 (edge tiles, padded plane pitch), two chunks of 8192."""
    X = _x(n, K, dev, n + K)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate_f32(X, G)
    ops.xtx_accumulate_f32(X[: n // 2], G)                       # accumulates
    torch.cuda.synchronize()
    x64 = X.double()
    want = x64.t() @ x64 + x64[: n // 2].t() @ x64[: n // 2]
    got = torch.tril(G).double()
    d = torch.sqrt(torch.diagonal(want))
    bound = 1e-5 * torch.outer(d, d)
    err = (got - torch.tril(want)).abs()
    assert bool((err <= bound + 1e-30).all()), float((err / bound).max())
    if K % 8:
        return                        # the 16-bit kernel wants K % 8 == 0; the fp32 path only K % 4 == 0
    # the bf16-rounded path for comparison
    Gb = torch.zeros_like(G)
    ops.xtx_accumulate(X.to(torch.bfloat16), Gb)
    ops.xtx_accumulate(X[: n // 2].to(torch.bfloat16), Gb)
    torch.cuda.synchronize()
    err_b = (torch.tril(Gb).double() - torch.tril(want)).abs()
    assert float((err / torch.outer(d, d)).max()) * 20 < float((err_b / torch.outer(d, d)).max())


def test_accumulator_routes_fp32_batches_by_policy(dev, monkeypatch):
    from quantool_amd.engine.gptq_linear import HessianAccumulator
    from quantool_amd.hip import ops

    K = 384
    X = _x(6 * 100, K, dev, 5).reshape(6, 100, K)
    x64 = X.reshape(-1, K).double()
    want = torch.tril(x64.t() @ x64)
    d = torch.sqrt(torch.diagonal(want))

    def run():
        acc = HessianAccumulator(K, dev, stage_tokens=256)          # forces staging + several flushes
        for b in range(6):
            acc.add(X[b:b + 1])
        return acc

    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    acc = run()
    assert acc.dtype == torch.float32 and acc.n == 6
    err = (torch.tril(acc.G).double() - want).abs() / torch.outer(d, d)
    assert float(err.max()) <= 1e-5
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "bf16")
    monkeypatch.setattr(ops, "_FP32_ACT_WARNED", False)
    acc_b = run()
    assert acc_b.dtype == torch.bfloat16
    err_b = (torch.tril(acc_b.G).double() - want).abs() / torch.outer(d, d)
    assert float(err_b.max()) > 20 * float(err.max())                # the rounding the exact path avoids
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "error")
    with pytest.raises(ValueError):
        run()


def test_fp32_linear_end_to_end_against_the_oracle(dev, oracle, monkeypatch):
    """An fp32 Linear with fp32 activations through the per-Linear path: Hessian from the fp32 Gram, fp32 weights;
    packed words and scales bit-exact against the oracle given the device's factor."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    K, R = 512, 96
    X = _x(4 * 200, K, dev, 9).reshape(4, 200, K)
    g = torch.Generator(device=dev).manual_seed(10)
    W = torch.randn((R, K), generator=g, device=dev) * 0.02
    acc = HessianAccumulator(K, dev)
    acc.add(X)
    keep = {}
    res = gptq_quantize_linear(W, acc, QuantArgs(num_bits=4, symmetric=True, group_size=128, actorder="static"), keep=keep)
    torch.cuda.synchronize()
    Gl = torch.tril(acc.G).cpu().numpy()
    H = oracle.hessian_from_gram_f32(Gl + np.tril(Gl, -1).T, acc.n)
    o = oracle.quantize_weight(W.cpu().numpy(), H, actorder="static", U_override=keep["U"].cpu().numpy())
    np.testing.assert_array_equal(res.weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]))
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o["scale"])
    assert res.weight_scale.dtype == torch.float32
