"""Known-answer tests for the CPU oracle, derived by hand from the published algorithms
(GPTQ arXiv 2210.17323, AWQ 2306.00978, SmoothQuant 2211.10438) and SURVEY.md Appendix A.

The reference holds no golden vector for this path (parity unpinned), so these KATs
plus the self-generated fixtures in tests/golden/ are what pins the oracle.
"""
import numpy as np
import pytest


def test_pack_nibble_order_known_answer(oracle):
    # q = [-8..-1] -> u = [0..7] -> element j at bits 4j: 0x76543210
    q = np.arange(-8, 0, dtype=np.int8)[None, :]
    assert oracle.pack_int4(q).view(np.uint32)[0, 0] == 0x76543210
    assert oracle.pack_int4_c(q).view(np.uint32)[0, 0] == 0x76543210
    q2 = np.array([[7, 0, -8, 1]], dtype=np.int8)  # padded with zero NIBBLES (not level 0)
    assert oracle.pack_int4(q2).view(np.uint32)[0, 0] == 0x0000908F


def test_pack_unpack_roundtrip_ragged(oracle):
    rng = np.random.default_rng(0)
    for K in (8, 13, 64, 131):
        q = rng.integers(-8, 8, size=(5, K)).astype(np.int8)
        p = oracle.pack_int4(q)
        assert p.shape == (5, (K + 7) // 8)
        assert np.array_equal(oracle.unpack_int4(p, K), q)
        assert np.array_equal(p, oracle.pack_int4_c(q))


def test_qparams_symmetric_known_answer(oracle):
    # one group of 4: min -3, max 1.5 -> absmax 3 -> scale 3/7.5 = 0.4, zp 0
    W = np.array([[-3.0, 1.5, 0.25, 0.0]], dtype=np.float32)
    s, z = oracle.minmax_qparams(W, 4, True, 4)
    assert s[0, 0] == np.float32(3.0) / np.float32(7.5) and z[0, 0] == 0
    # all-positive group: min is clamped to include 0
    W = np.array([[1.0, 2.0, 3.0, 6.0]], dtype=np.float32)
    s, z = oracle.minmax_qparams(W, 4, True, 4)
    assert s[0, 0] == np.float32(6.0) / np.float32(7.5)
    # all-zero group: scale floors at fp32 eps
    s, z = oracle.minmax_qparams(np.zeros((1, 4), np.float32), 4, True, 4)
    assert s[0, 0] == np.finfo(np.float32).eps


def test_qparams_asymmetric_known_answer(oracle):
    # min -1, max 2 -> scale 3/15 = 0.2; zp = clamp(round(-8 - (-1/0.2)), -8, 7) = round(-3) = -3
    W = np.array([[-1.0, 2.0, 0.5, 0.0]], dtype=np.float32)
    s, z = oracle.minmax_qparams(W, 4, False, 4)
    assert s[0, 0] == np.float32(3.0) / np.float32(15.0)
    assert z[0, 0] == -3.0
    sc, zc = oracle.minmax_qparams_c(W, 4, False, 4)
    assert np.array_equal(s, sc) and np.array_equal(z, zc)


def test_fake_quantize_ties_round_half_even_and_clamp(oracle):
    s = np.float32(1.0)
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 7.5, 8.4, -8.5, -9.0], dtype=np.float32)
    q, dq = oracle.fake_quantize(x, s, np.float32(0), 4)
    assert q.tolist() == [0, 2, 2, -0.0, -2, 7, 7, -8, -8]
    assert np.array_equal(dq, q)


def test_identity_hessian_is_round_to_nearest(oracle):
    rng = np.random.default_rng(1)
    W = (rng.standard_normal((8, 256)) * 0.05).astype(np.float32)
    o = oracle.quantize_weight(W, np.eye(256, dtype=np.float32), actorder=None, percdamp=0.0,
                               U_override=np.eye(256, dtype=np.float32))
    s, z = oracle.minmax_qparams(W, 128, True, 4)
    g = np.arange(256) // 128
    q_rtn, _ = oracle.fake_quantize(W, s[:, g], z[:, g], 4)
    assert np.array_equal(o["q"], q_rtn.astype(np.int8))


def test_diagonal_hessian_is_round_to_nearest(oracle):
    # diagonal H => U diagonal => no cross-column error feedback
    rng = np.random.default_rng(2)
    W = (rng.standard_normal((4, 128)) * 0.05).astype(np.float32)
    H = np.diag(rng.random(128).astype(np.float32) + 0.5)
    o = oracle.quantize_weight(W, H, actorder=None)
    s, z = oracle.minmax_qparams(W, 128, True, 4)
    q_rtn, _ = oracle.fake_quantize(W, s[:, [0] * 128], z[:, [0] * 128], 4)
    assert np.array_equal(o["q"], q_rtn.astype(np.int8))


def test_two_column_sweep_by_hand(oracle):
    """2 columns, scale 1, U = [[2, 1], [0, 4]]: w = (0.6, 0.3).
    col 0: q=round(0.6)=1, err=(0.6-1)/2=-0.2; w1 <- 0.3 - (-0.2*1) = 0.5 -> q=round(0.5)=0 (half-even)."""
    W = np.array([[0.6, 0.3]], dtype=np.float32)
    U = np.array([[2.0, 1.0], [0.0, 4.0]], dtype=np.float32)
    scale = np.ones((1, 1), np.float32)
    zp = np.zeros((1, 1), np.float32)
    g_idx = np.zeros(2, np.int32)
    for sweep in (oracle.gptq_sweep_c, oracle.gptq_sweep_numpy):
        Q, Wdq, loss = sweep(W, U, scale, zp, g_idx, 128, 4)
        assert Q.tolist() == [[1, 0]]
        # loss = ((0.6-1)^2/4 + (0.5-0)^2/16)/2
        e0 = (np.float32(0.6) - np.float32(1.0)) ** 2 / np.float32(4.0)
        w1 = np.float32(0.3) - (np.float32(0.6) - np.float32(1.0)) / np.float32(2.0) * np.float32(1.0)
        e1 = w1 ** 2 / np.float32(16.0)
        assert loss[0] == np.float32((e0 + e1) / np.float32(2.0))


def test_sweep_invariant_to_hessian_scaling(oracle):
    """err/d * U[i,:] is homogeneous of degree 0 in H (SURVEY A.2 property): scaling H by a power
    of two changes nothing, bit for bit."""
    rng = np.random.default_rng(3)
    K = 128
    X = rng.standard_normal((512, K)).astype(np.float32)
    H = (X.T @ X / 256).astype(np.float32)
    W = (rng.standard_normal((8, K)) * 0.03).astype(np.float32)
    a = oracle.quantize_weight(W, H, actorder="static")
    b = oracle.quantize_weight(W, H * np.float32(4.0), actorder="static")
    assert np.array_equal(a["q"], b["q"])


def test_c_and_numpy_sweeps_agree_across_blocks(oracle):
    rng = np.random.default_rng(4)
    R, K = 12, 384
    X = rng.standard_normal((1024, K)).astype(np.float32)
    X[:, :4] *= 8
    H = (X.T @ X * (2.0 / 8)).astype(np.float32)
    W = (rng.standard_normal((R, K)) * 0.02).astype(np.float32)
    for ao in (None, "static", "group"):
        for sym in (True, False):
            a = oracle.quantize_weight(W, H, symmetric=sym, actorder=ao, sweep=oracle.gptq_sweep_c)
            b = oracle.quantize_weight(W, H, symmetric=sym, actorder=ao, sweep=oracle.gptq_sweep_numpy)
            assert np.array_equal(a["q"], b["q"])
            assert np.array_equal(a["w_dq"], b["w_dq"])


def test_actorder_semantics(oracle):
    rng = np.random.default_rng(5)
    K = 256
    X = rng.standard_normal((600, K)).astype(np.float32)
    X[:, 200:210] *= 6
    H = (X.T @ X / 300).astype(np.float32)
    W = (rng.standard_normal((4, K)) * 0.02).astype(np.float32)
    st = oracle.quantize_weight(W, H, actorder="static")
    gr = oracle.quantize_weight(W, H, actorder="group")
    no = oracle.quantize_weight(W, H, actorder=None)
    assert st["g_idx"] is None and no["g_idx"] is None           # static: g_idx dropped after un-permute
    assert gr["g_idx"] is not None and sorted(np.bincount(gr["g_idx"]).tolist()) == [128, 128]
    assert np.array_equal(st["scale"], no["scale"])               # static: observer runs on the original W
    assert st["perm"][0] in range(200, 210)                        # most salient channel swept first
    assert np.array_equal(np.sort(st["perm"]), np.arange(K))


def test_dead_columns(oracle):
    K = 128
    H = np.eye(K, dtype=np.float32)
    H[5, 5] = 0
    W = np.ones((2, K), np.float32) * 0.3
    o = oracle.quantize_weight(W, H, actorder=None)
    assert o["dead"][5] and o["dead"].sum() == 1
    assert np.all(o["q"][:, 5] == 0)   # W[:, dead] = 0 before the sweep


def test_lapack_path_matches_fp64_and_ul_shortcut(oracle):
    rng = np.random.default_rng(6)
    K = 96
    X = rng.standard_normal((400, K))
    H = (X.T @ X / 200).astype(np.float32)
    Hd, _, damp = oracle.hessian_dead_and_damp(H)
    assert damp == np.float32(0.01) * np.float32(np.diag(H).astype(np.float64).mean())
    U1, ok = oracle.cholesky_inverse_upper_lapack(Hd)
    assert ok
    U2 = oracle.cholesky_inverse_upper_f64(Hd)
    U3 = oracle.cholesky_inverse_upper_ul(Hd)
    assert np.abs(U1 - U2).max() < 1e-5 * np.abs(U2).max()
    assert np.abs(U3 - U2).max() < 1e-12 * np.abs(U2).max()      # the one-factorisation identity
    assert np.allclose(U2.T @ U2 @ Hd.astype(np.float64), np.eye(K), atol=1e-6)
    bad, ok = oracle.cholesky_inverse_upper_lapack(-np.eye(4, dtype=np.float32))
    assert not ok and np.array_equal(bad, np.eye(4, dtype=np.float32))   # LinAlgError -> identity


def test_running_hessian_equals_scaled_gram(oracle):
    rng = np.random.default_rng(7)
    K, S, T = 64, 6, 40
    xb = oracle.f32_to_bf16_bits(rng.standard_normal((S * T, K)).astype(np.float32))
    Href = oracle.accumulate_hessian_reference([xb[i * T:(i + 1) * T] for i in range(S)], K)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), S)
    assert np.abs(H - Href).max() <= 2e-6 * np.abs(H).max()
    assert np.array_equal(np.tril(oracle.gram_f64(xb)), oracle.gram_f64_c(xb))


def test_save_time_requantization_recovers_levels(oracle):
    """upstream re-derives int4 from the bf16 dequantised weight and bf16 scale
    at save time; that reproduces the sweep's levels, so emitting q directly is equivalent."""
    rng = np.random.default_rng(8)
    R, K = 16, 256
    X = rng.standard_normal((700, K)).astype(np.float32)
    H = (X.T @ X / 350).astype(np.float32)
    for sym in (True, False):
        W = oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits((rng.standard_normal((R, K)) * 0.02).astype(np.float32)))
        o = oracle.quantize_weight(W, H, symmetric=sym, actorder="static")
        g_cols = (np.arange(K) // 128).astype(np.int64)
        q2 = oracle.requantize_at_save(oracle.f32_to_bf16_bits(o["w_dq"]), oracle.f32_to_bf16_bits(o["scale"]),
                                       o["zp"], g_cols, 4)
        assert np.array_equal(q2, o["q"])


def test_awq_pseudo_quant_uses_max_int_7(oracle):
    W = np.array([[0.7, -0.35, 0.1, 0.0]], dtype=np.float32)
    out = oracle.awq_pseudo_quantize(W, 4, True, 4)
    sc = np.float32(0.7) / np.float32(7.0)
    want = np.clip(np.rint(W / sc), -8, 7) * sc
    assert np.array_equal(out, want.astype(np.float32))


def test_awq_scales_normalisation(oracle):
    x_mean = np.array([1.0, 4.0, 0.25, 1.0], np.float32)
    w_mean = np.array([0.5, 0.5, 0.5, 0.5], np.float32)
    s0 = oracle.awq_scales_for_ratio(x_mean, w_mean, 0.0)
    assert np.allclose(s0, 1.0)  # ratio 0 with equal w_mean: all channels equal -> normalised to 1
    s = oracle.awq_scales_for_ratio(x_mean, w_mean, 0.5)
    assert np.isclose(np.sqrt(s.max() * s.min()), 1.0, rtol=1e-6)
    assert s[1] > s[0] > s[2]


def test_awq_search_prefers_scaling_salient_channels(oracle):
    rng = np.random.default_rng(9)
    K, R, N = 128, 32, 512
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, :4] *= 30
    W = (rng.standard_normal((R, K)) * 0.05).astype(np.float32)
    r = oracle.awq_best_scale(oracle.f32_to_bf16_bits(X), [W], 128)
    assert r["best_ratio_idx"] > 0
    assert r["losses"][r["best_ratio_idx"]] < r["losses"][0]


def test_smoothquant_scales_known_answer(oracle):
    amin = np.array([-1.0, 0.0, -2.0], np.float32)
    amax = np.array([3.0, 0.0, 2.0], np.float32)
    W = np.array([[0.5, 0.0, -4.0], [0.25, 0.0, 1.0]], np.float32)
    s = oracle.smoothquant_scales(amin, amax, [W], 0.5)
    # a = (4, 0, 4); w = (0.5, 0, 4): s = sqrt(a)/sqrt(w) = (2.828.., a (w==0) = 0, 1)
    assert np.isclose(s[0], np.sqrt(4.0) / np.sqrt(0.5)) and s[1] == 0.0 and np.isclose(s[2], 1.0)
