"""End-to-end per-Linear GPTQ on the GPU vs the CPU oracle (stage-wise contract).

(a) scales / zero-points: bit-exact (they come from the original W);
(b) given the SAME factor U (taken from the GPU run), integer levels, packed words and g_idx
    are bit-exact against the oracle's quantize_weight;
(c) against the fully independent oracle (fp32 LAPACK three-step inverse on its own fp64-summed
    Hessian) the nibble mismatch rate is reported and bounded.
"""
import numpy as np
import pytest
import torch

from tests.util import bits_to_bf16_tensor, synth_activations, synth_weight

pytestmark = pytest.mark.gpu


def _run(dev, oracle, R_list, K, n_samples, T, scheme_kw, actorder, seed):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    xb = synth_activations(n_samples * T, K, seed=seed)
    Ws = [synth_weight(R, K, seed=seed + 10 + i) for i, R in enumerate(R_list)]
    Wb = [oracle.f32_to_bf16_bits(w) for w in Ws]
    Wf = [oracle.bf16_bits_to_f32(b) for b in Wb]
    acc = HessianAccumulator(K, dev)
    X = bits_to_bf16_tensor(xb, dev).reshape(n_samples, T, K)
    for b in range(n_samples):
        acc.add(X[b:b + 1])
    qa = QuantArgs(actorder=actorder, **scheme_kw)
    keep = {}
    res = gptq_quantize_shared([bits_to_bf16_tensor(b, dev) for b in Wb], acc, qa, keep=keep)
    torch.cuda.synchronize()
    Gl = np.tril(acc.G.cpu().numpy())
    Gfull = Gl + np.tril(Gl, -1).T
    return xb, Wf, res, keep, Gfull, acc.n


@pytest.mark.parametrize("actorder", [None, "static", "group"])
@pytest.mark.parametrize("sym", [True, False])
def test_linear_bit_exact_given_gpu_factor(dev, oracle, actorder, sym):
    K, n_samples, T = 512, 8, 160
    kw = dict(num_bits=4, symmetric=sym, group_size=128)
    xb, Wf, res, keep, Gfull, n = _run(dev, oracle, [96, 40], K, n_samples, T, kw, actorder, seed=3)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    U_gpu = keep["U"].cpu().numpy()
    for w, r in zip(Wf, res):
        o = oracle.quantize_weight(w, H, group_size=128, symmetric=sym, num_bits=4, actorder=actorder,
                                   U_override=U_gpu)
        if actorder is not None:
            assert np.array_equal(keep["perm"].cpu().numpy(), o["perm"].astype(np.int32))
        np.testing.assert_array_equal(r.scale_f32.cpu().numpy(), o["scale"])
        np.testing.assert_array_equal(r.zp_f32.cpu().numpy(), o["zp"])
        np.testing.assert_array_equal(r.weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]))
        if actorder == "group":
            np.testing.assert_array_equal(r.weight_g_idx.cpu().numpy(), o["g_idx"])
        else:
            assert r.weight_g_idx is None
        np.testing.assert_array_equal(r.dequantized().cpu().numpy(), o["w_dq"])
        # scales travel in model dtype (bf16): within 1e-5 is the north_star bar for fp32 scales;
        # the bf16 cast itself is exact rounding of the fp32 value
        assert r.weight_scale.dtype == torch.bfloat16
        assert int(r.info.item()) == 0


def test_linear_vs_independent_oracle_mismatch_rate(dev, oracle):
    K, n_samples, T = 512, 8, 160
    kw = dict(num_bits=4, symmetric=True, group_size=128)
    xb, Wf, res, keep, Gfull, n = _run(dev, oracle, [128], K, n_samples, T, kw, "static", seed=5)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), n)      # independent Hessian
    np.testing.assert_allclose(oracle.hessian_from_gram_f32(Gfull, n), H, rtol=0,
                               atol=1e-5 * np.abs(np.diag(H)).max())
    o = oracle.quantize_weight(Wf[0], H, actorder="static", inverse="lapack")
    q_gpu = oracle.unpack_int4(res[0].weight_packed.cpu().numpy(), K)
    rate = float((q_gpu != o["q"]).mean())
    print(f"nibble mismatch rate vs independent LAPACK oracle: {rate:.3e}")
    # error feedback amplifies last-bit differences in U into flipped roundings; upstream itself
    # is not reproducible across BLAS thread counts at this level
    assert rate <= 1e-4      # 0 observed (65 536 nibbles); a regression of the factor or the sweep shows far above this
    np.testing.assert_array_equal(res[0].scale_f32.cpu().numpy(), o["scale"])


def test_w8a16_channelwise(dev, oracle):
    K, n_samples, T = 256, 4, 128
    kw = dict(num_bits=8, symmetric=True, group_size=None, strategy="channel")
    xb, Wf, res, keep, Gfull, n = _run(dev, oracle, [64], K, n_samples, T, kw, None, seed=9)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    o = oracle.quantize_weight(Wf[0], H, group_size=-1, symmetric=True, num_bits=8, actorder=None,
                               U_override=keep["U"].cpu().numpy())
    assert res[0].weight_packed is None
    np.testing.assert_array_equal(res[0].weight_q.cpu().numpy(), o["q"])


def test_non_pd_hessian_falls_back_to_rtn(dev, oracle):
    """Upstream: LinAlgError -> Hinv = I.  Force it with dampening_frac < 0."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    K, R = 256, 32
    acc = HessianAccumulator(K, dev)
    X = torch.randn(1, 64, K, device=dev).to(torch.bfloat16)   # rank 64 < K
    acc.add(X)
    Wn = synth_weight(R, K, seed=1)
    W = torch.from_numpy(Wn).to(dev)
    r = gptq_quantize_linear(W, acc, QuantArgs(actorder=None), dampening_frac=-1.0)
    torch.cuda.synchronize()
    assert int(r.info.item()) != 0
    scale, zp = oracle.minmax_qparams(Wn, 128, True, 4)
    g = np.arange(K) // 128
    q_rtn, _ = oracle.fake_quantize(Wn, scale[:, g], zp[:, g], 4)
    np.testing.assert_array_equal(oracle.unpack_int4(r.weight_packed.cpu().numpy(), K), q_rtn.astype(np.int8))
