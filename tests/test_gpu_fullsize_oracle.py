"""Config 2 (Llama-3-8B shapes) against the ORACLE at the real in_features, not only through
size-independent properties: the oracle's C sweep handles 4096 x 4096 in seconds, and rows are
independent given the factor U, so a 256-row slice of down_proj (K = 14336) is affordable too.

Bars (north_star: "bit-exact int4 packed weights and group indices, scales within 1e-5"):
  * given the GPU's factor U: scales, zero-points, packed words and dequantised weights bit-exact;
  * against the independent LAPACK three-step inverse on the same Hessian: the nibble mismatch rate
    is measured, printed and bounded (error feedback turns last-bit differences of U into flipped
    roundings; upstream itself is not reproducible across BLAS thread counts at that level).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _acts(n, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.empty((n, K), dtype=torch.bfloat16, device=dev)
    gain = torch.ones(K, device=dev)
    gain[torch.randperm(K, generator=g, device=dev)[: K // 100]] = 10.0     # outlier channels
    for t0 in range(0, n, 16384):
        t1 = min(n, t0 + 16384)
        X[t0:t1] = (torch.randn((t1 - t0, K), generator=g, device=dev) * gain).to(torch.bfloat16)
    return X


def _gpu_run(dev, R, K, n_samples, T, seed, actorder="static", symmetric=True):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    X = _acts(n_samples * T, K, dev, seed)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    W = (torch.randn((R, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    acc = HessianAccumulator(K, dev)
    for s in range(n_samples):                       # the plugin path's calling pattern: one sample per call
        acc.add(X[s * T:(s + 1) * T])
    keep = {}
    res = gptq_quantize_shared([W], acc, QuantArgs(num_bits=4, symmetric=symmetric, group_size=128, actorder=actorder),
                               keep=keep)[0]
    torch.cuda.synchronize()
    Gl = torch.tril(acc.G).cpu().numpy()
    Gfull = Gl + np.tril(Gl, -1).T
    return W.float().cpu().numpy(), res, keep, Gfull, acc.n


def _assert_bit_exact(oracle, res, o, K, actorder):
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o["scale"])
    np.testing.assert_array_equal(res.zp_f32.cpu().numpy(), o["zp"])
    np.testing.assert_array_equal(res.weight_packed.cpu().numpy(), oracle.pack_int4(o["q"]))
    np.testing.assert_array_equal(res.dequantized().cpu().numpy(), o["w_dq"])
    if actorder == "group":
        np.testing.assert_array_equal(res.weight_g_idx.cpu().numpy(), o["g_idx"])
    assert int(res.info.item()) == 0


def test_q_proj_size_bit_exact_given_gpu_factor_and_lapack_rate(dev, oracle):
    """4096 x 4096 (q_proj / o_proj), W4A16 g128, actorder = static (upstream's default)."""
    R = K = 4096
    Wf, res, keep, Gfull, n = _gpu_run(dev, R, K, n_samples=32, T=384, seed=11)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    o = oracle.quantize_weight(Wf, H, actorder="static", U_override=keep["U"].cpu().numpy())
    assert np.array_equal(keep["perm"].cpu().numpy(), o["perm"].astype(np.int32))
    _assert_bit_exact(oracle, res, o, K, "static")
    # independent factor: fp32 LAPACK potrf / potri / potrf on the same Hessian
    o2 = oracle.quantize_weight(Wf, H, actorder="static", inverse="lapack")
    assert o2["ok"]
    q_gpu = oracle.unpack_int4(res.weight_packed.cpu().numpy(), K)
    rate = float((q_gpu != o2["q"]).mean())
    print(f"\n[fullsize] q_proj 4096x4096: nibble mismatch rate vs LAPACK-factor oracle = {rate:.3e}")
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o2["scale"])
    # yardstick: two CPU factorisations of the same matrix (fp64 three-step rounded to fp32 vs fp32 LAPACK)
    Hp = H[o2["perm"]][:, o2["perm"]]
    U64 = oracle.cholesky_inverse_upper_f64_lapack(oracle.hessian_dead_and_damp(Hp, 0.01)[0]).astype(np.float32)
    o3 = oracle.quantize_weight(Wf, H, actorder="static", U_override=U64)
    rate64 = float((o3["q"] != o2["q"]).mean())
    print(f"[fullsize] q_proj 4096x4096: fp64-factor oracle vs LAPACK-factor oracle = {rate64:.3e}")
    # observed 2.5e-5 on the round-2/3 code (421 of 16.8 M nibbles; 9.0e-6 before the K = 4096
    # Gram passes moved to the 16x16x32 kernel -- the count moves with any change of a summation order);
    # the bound is 10x the observation, and a few times the distance between the two CPU factorisations
    assert rate <= 2.5e-4
    assert rate <= max(5 * rate64, 2e-5), (rate, rate64)


def test_q_proj_size_group_actorder_asymmetric(dev, oracle):
    """Same size, the other branch set: asymmetric zero-points and actorder = group (g_idx saved)."""
    R, K = 1024, 4096        # k_proj / v_proj rows
    Wf, res, keep, Gfull, n = _gpu_run(dev, R, K, n_samples=32, T=384, seed=13, actorder="group", symmetric=False)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    o = oracle.quantize_weight(Wf, H, actorder="group", symmetric=False, U_override=keep["U"].cpu().numpy())
    _assert_bit_exact(oracle, res, o, K, "group")


def test_down_proj_row_slice_bit_exact_given_gpu_factor(dev, oracle):
    """down_proj's K = 14336: rows are independent given U, so a 256-row slice swept by the GPU and by
    the oracle with the GPU's 14336 x 14336 factor must agree bit for bit."""
    R, K = 256, 14336
    Wf, res, keep, Gfull, n = _gpu_run(dev, R, K, n_samples=80, T=384, seed=17)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    del Gfull
    U = keep["U"].cpu().numpy()
    o = oracle.quantize_weight(Wf, H, actorder="static", U_override=U)
    assert np.array_equal(keep["perm"].cpu().numpy(), o["perm"].astype(np.int32))
    _assert_bit_exact(oracle, res, o, K, "static")


def test_llama3_70b_k8192_row_slice_bit_exact_given_gpu_factor(dev, oracle):
    """Config 4's K = 8192 groups (Llama-3-70B hidden size: q/k/v/o and gate/up inputs): a 192-row slice,
    asymmetric + actorder = group, against the oracle with the GPU's 8192 x 8192 factor.  (The K = 28672
    group runs in test_gpu_edges.py; the 8-rank split of both is planned in test_sharding_gloo.py.)"""
    R, K = 192, 8192
    Wf, res, keep, Gfull, n = _gpu_run(dev, R, K, n_samples=48, T=384, seed=19, actorder="group", symmetric=False)
    H = oracle.hessian_from_gram_f32(Gfull, n)
    del Gfull
    o = oracle.quantize_weight(Wf, H, actorder="group", symmetric=False, U_override=keep["U"].cpu().numpy())
    assert np.array_equal(keep["perm"].cpu().numpy(), o["perm"].astype(np.int32))
    _assert_bit_exact(oracle, res, o, K, "group")
