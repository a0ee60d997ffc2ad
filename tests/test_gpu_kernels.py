"""Parity of every HIP kernel against the CPU oracle, called through the C ABI (ctypes).

Bars (DESIGN.md "Parity contract"): integer / index outputs bit-exact; fp32 outputs that
restate an elementwise upstream op bit-exact; Gram matrix and the Cholesky factor within the
tolerances written next to each assertion (their upstream counterparts are BLAS/LAPACK calls
whose summation order torch does not fix).
"""
import numpy as np
import pytest
import torch

from tests.util import bf16_tensor_to_bits, bits_to_bf16_tensor, synth_activations, synth_weight

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from quantool_amd.hip import ops as _ops

    return _ops


# ------------------------------------------------------------------------------------- a7
@pytest.mark.parametrize("n_tokens,K", [(64, 64), (200, 64), (1000, 384), (4096, 512), (777, 264), (130, 1032)])
def test_xtx_matches_f64_gram(ops, oracle, dev, n_tokens, K):
    xb = synth_activations(n_tokens, K, seed=n_tokens + K)
    X = bits_to_bf16_tensor(xb, dev)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    torch.cuda.synchronize()
    Gt = oracle.gram_f64(xb)
    got = np.tril(G.cpu().numpy().astype(np.float64))
    want = np.tril(Gt)
    # fp32 accumulation of n exact products: |err| <~ sqrt(n)*eps*sum|x_i x_j|; 1e-5 relative to
    # the geometric mean of the diagonals is the north_star's "H within 1e-5" bar.
    dscale = np.sqrt(np.outer(np.diag(Gt), np.diag(Gt)))
    assert np.all(np.abs(got - want) <= 1e-5 * np.tril(dscale) + 1e-30)
    # accumulate semantics: a second call adds
    ops.xtx_accumulate(X, G)
    torch.cuda.synchronize()
    got2 = np.tril(G.cpu().numpy().astype(np.float64))
    assert np.all(np.abs(got2 - 2 * want) <= 2e-5 * np.tril(dscale) + 1e-30)


@pytest.mark.parametrize("K,order", [(5888, None), (6400, None), (8192, None), (5888, "1"), (5888, "0"), (4096, "2")])
def test_xtx_tile_tables_at_and_above_one_round_of_tiles(ops, dev, monkeypatch, K, order):
    """From one full round of tiles up (K >= 5632) the tile table is the 32-aligned walk; K = 5888 (23 panels: a
    ragged last block row and 20 left-over tiles at the table's end) and 6400 (25 panels) are its awkward shapes,
    8192 the first with whole rounds AND a remainder split over token chunks.  Each lower tile, diagonal ones
    included, against an fp32 torch product; the other orders (QT_XTX_ORDER) on the same input must agree to the
    bit (a tile's value does not depend on where the table lists it; tables are cached per (K, order))."""
    n = 1536
    torch.manual_seed(K)
    X = torch.randn(n, K, device=dev).to(torch.bfloat16)
    Gd = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, Gd)                      # the default order of this K
    if order is not None:
        monkeypatch.setenv("QT_XTX_ORDER", order)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    torch.cuda.synchronize()
    ref = X.float().t() @ X.float()
    d = torch.sqrt(torch.outer(torch.diagonal(ref), torch.diagonal(ref)))
    err = ((torch.tril(G) - torch.tril(ref)).abs() / d).max().item()
    assert err <= 1e-5, err
    assert torch.equal(torch.tril(G), torch.tril(Gd))


def test_xtx_is_deterministic_and_handles_3d_input(ops, dev):
    torch.manual_seed(0)
    X = torch.randn(4, 96, 256, device=dev).to(torch.bfloat16)
    G1 = torch.zeros((256, 256), dtype=torch.float32, device=dev)
    G2 = torch.zeros_like(G1)
    ops.xtx_accumulate(X, G1)
    ops.xtx_accumulate(X, G2)
    torch.cuda.synchronize()
    assert torch.equal(torch.tril(G1), torch.tril(G2))


@pytest.mark.parametrize("n_tokens,K,pad", [(4096, 2816, 0), (1000 + 13, 768, 64), (2048, 4352, 0)])
def test_xtx_split_tiles_strided_and_ragged(ops, oracle, dev, n_tokens, K, pad):
    """Paths the small cases do not reach: more tiles than one round can take plus a remainder that is
    split over token chunks into slabs (K = 4352: 153 tiles; K = 2816: 66 tiles x S chunks), a row
    pitch larger than K, and a ragged token tail together with that pitch (two launches)."""
    xb = synth_activations(n_tokens, K, seed=n_tokens + K)
    X = bits_to_bf16_tensor(xb, dev)
    if pad:
        Xw = torch.zeros((n_tokens, K + pad), dtype=torch.bfloat16, device=dev)
        Xw[:, :K] = X
        X = Xw[:, :K]
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    ops.xtx_accumulate(X, G)
    torch.cuda.synchronize()
    Gt = oracle.gram_f64(xb)
    d = np.sqrt(np.diag(Gt))
    got = np.tril(G.cpu().numpy().astype(np.float64))
    assert np.all(np.abs(got - 2 * np.tril(Gt)) <= 2e-5 * np.tril(np.outer(d, d)) + 1e-30)


@pytest.mark.parametrize("shape", ["16", "32"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_xtx_both_mfma_shapes(ops, dev, monkeypatch, shape, dtype):
    """The Gram kernel exists on two MFMA shapes (16x16x32 by default, the 32x32x16 ring with the progress
    throttle for many-round launches); QT_XTX_SHAPE forces one, so both are covered whatever the default."""
    monkeypatch.setenv("QT_XTX_SHAPE", shape)
    g = torch.Generator(device=dev).manual_seed(int(shape))
    for n, K in ((777, 264), (4096 + 5, 1280), (300, 4352)):
        X = torch.randn((n, K), generator=g, device=dev).to(dtype)
        G = torch.zeros((K, K), dtype=torch.float32, device=dev)
        ops.xtx_accumulate(X, G)
        ops.xtx_accumulate(X, G)
        want = 2 * torch.tril(X.double().t() @ X.double())
        d = torch.sqrt(torch.diag(want) / 2)
        assert bool(((torch.tril(G).double() - want).abs() <= 2e-5 * torch.tril(torch.outer(d, d)) + 1e-30).all())


def test_per_sample_staging_matches_single_launch(ops, oracle, dev):
    """The plugin path's calling pattern (reference base.py:161: one sample per batch) goes through the
    accumulator's device token buffer; the result must be the Gram sum a single launch gives (the
    fp32 partial sums are grouped differently: 1e-5 on G, north_star's bar) and n the sample count."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator

    S, T, K = 40, 96, 520
    xb = synth_activations(S * T, K, seed=5)
    X = bits_to_bf16_tensor(xb, dev).reshape(S, T, K)
    one = HessianAccumulator(K, dev, stage_tokens=0)
    one.add(X)
    staged = HessianAccumulator(K, dev, stage_tokens=1024)       # forces several flushes
    for i in range(S):
        staged.add(X[i:i + 1])
    assert staged._fill > 0                                        # something is still staged ...
    Gs = staged.G                                                  # ... and reading G folds it in
    assert staged._fill == 0 and staged.n == one.n == S
    Gt = oracle.gram_f64(xb)
    d = np.sqrt(np.diag(Gt))
    for G in (one.G, Gs):
        got = np.tril(G.cpu().numpy().astype(np.float64))
        assert np.all(np.abs(got - np.tril(Gt)) <= 1e-5 * np.tril(np.outer(d, d)) + 1e-30)
    # a batch at least DIRECT_TOKENS long skips the buffer; staged rows still come first or later, never lost
    big = HessianAccumulator(K, dev, stage_tokens=512)
    big.add(X[:2].reshape(-1, K), num_samples=2)
    big.add(X[2:].reshape(-1, K), num_samples=S - 2)               # 3648 rows >= stage_tokens -> direct launch
    got = np.tril(big.G.cpu().numpy().astype(np.float64))
    assert np.all(np.abs(got - np.tril(Gt)) <= 1e-5 * np.tril(np.outer(d, d)) + 1e-30)


def test_xtx_rejects_bad_shapes(ops, dev):
    from quantool_amd.hip._lib import HipBackendError

    X = torch.zeros((16, 12), dtype=torch.bfloat16, device=dev)
    G = torch.zeros((12, 12), dtype=torch.float32, device=dev)
    with pytest.raises(HipBackendError):
        ops.xtx_accumulate(X, G)  # K % 8 != 0


# ------------------------------------------------------------------------------------ a12/a13
@pytest.mark.parametrize("n_tokens,K", [(50, 64), (1000, 264), (5000, 4096)])
def test_act_stats(ops, oracle, dev, n_tokens, K):
    xb = synth_activations(n_tokens, K, seed=7)
    X = bits_to_bf16_tensor(xb, dev)
    s = torch.zeros(K, dtype=torch.float32, device=dev)
    mn = torch.full((K,), float("inf"), dtype=torch.float32, device=dev)
    mx = torch.full((K,), float("-inf"), dtype=torch.float32, device=dev)
    ops.act_stats_accumulate(X, s, mn, mx)
    torch.cuda.synchronize()
    xf = oracle.bf16_bits_to_f32(xb)
    omn, omx = oracle.channel_minmax(xb)
    assert np.array_equal(mn.cpu().numpy(), omn)  # min / max are exact
    assert np.array_equal(mx.cpu().numpy(), omx)
    want = np.abs(xf).astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(s.cpu().numpy(), want, rtol=2e-6 * np.sqrt(n_tokens))


# ------------------------------------------------------------------------------------- a8/a9
@pytest.mark.parametrize("K,with_perm,with_dead", [(64, False, False), (264, True, True), (512, True, False)])
def test_hessian_prepare(ops, oracle, dev, K, with_perm, with_dead):
    rng = np.random.default_rng(K)
    n_tok, n_samples = 3 * K, 6
    xb = synth_activations(n_tok, K, seed=K)
    if with_dead:
        xb[:, [3, K - 2]] = 0
    X = bits_to_bf16_tensor(xb, dev)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    perm_np = rng.permutation(K).astype(np.int32) if with_perm else None
    perm = torch.from_numpy(perm_np).to(dev) if with_perm else None
    A, dead, diag = ops.hessian_prepare(G, n_samples, 0.01, perm)
    torch.cuda.synchronize()
    Gl = np.tril(G.cpu().numpy())
    Gfull = Gl + np.tril(Gl, -1).T
    c = np.float32(2.0 / n_samples)
    H = (Gfull * c).astype(np.float32)  # one fp32 multiply per element, as the kernel
    np.testing.assert_array_equal(diag.cpu().numpy(), np.diag(H))
    Hp = H[perm_np][:, perm_np] if with_perm else H
    Hd, odead, damp = oracle.hessian_dead_and_damp(Hp, 0.01)
    np.testing.assert_array_equal(dead.cpu().numpy().astype(bool), odead)
    want = Hd[::-1, ::-1]
    got = A.cpu().numpy()
    iu = np.triu_indices(K, 1)
    np.testing.assert_array_equal(got[iu], want[iu])  # off-diagonal: exact
    # diagonal: +damp where the mean is an fp64 sum in a different order -> allow 1 ulp
    np.testing.assert_allclose(np.diag(got), np.diag(want), rtol=2.4e-7)


@pytest.mark.parametrize("K,with_perm,with_dead", [(2048, True, True), (2056, True, False), (4096, False, False),
                                                   (5120, True, True)])
def test_hessian_prepare_two_pass_equals_one_pass(ops, dev, monkeypatch, K, with_perm, with_dead):
    """From K = 2048 the prepared matrix is built in two coalesced passes (symmetric copy, then one row of it in LDS per
    output row) instead of one kernel that walks down columns of the lower triangle -- pure data movement, so the
    upper triangle of A (all the factorisation reads), the dead flags and the diagonal are the one-pass kernel's bit
    for bit; the strict upper part of G (never valid) is poisoned to show that nothing reads it."""
    torch.manual_seed(K)
    X = torch.randn(3 * K, K, device=dev)
    if with_dead:
        X[:, [5, K - 3, K // 2]] = 0
    X = X.to(torch.bfloat16)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    tri = torch.triu(torch.ones(K, K, dtype=torch.bool, device=dev), 1)
    blk = (torch.arange(K, device=dev) // 256)
    G[tri & (blk[:, None] != blk[None, :])] = float("nan")      # outside the lower 256-tiles: garbage in real use
    perm = torch.randperm(K, device=dev).to(torch.int32) if with_perm else None
    monkeypatch.setenv("QT_PREPARE_TWO_PASS", "0")
    A0, dead0, diag0 = ops.hessian_prepare(G, 8, 0.01, perm)
    monkeypatch.delenv("QT_PREPARE_TWO_PASS")
    A1, dead1, diag1 = ops.hessian_prepare(G, 8, 0.01, perm)
    torch.cuda.synchronize()
    assert torch.equal(dead0, dead1) and torch.equal(diag0, diag1)
    assert torch.equal(torch.triu(A0), torch.triu(A1))
    assert bool(torch.isfinite(torch.triu(A1)).all())


@pytest.mark.parametrize("K", [1, 7, 1000, 4096, 14336])
def test_argsort_desc_stable(ops, dev, K):
    g = torch.Generator(device=dev).manual_seed(K)
    v = torch.randn(K, generator=g, device=dev)
    v[::5] = float(v[0])                              # plenty of exact ties
    perm, inv = ops.argsort_desc(v)
    torch.cuda.synchronize()
    want = np.argsort(-v.cpu().numpy(), kind="stable").astype(np.int32)
    np.testing.assert_array_equal(perm.cpu().numpy(), want)
    np.testing.assert_array_equal(inv.cpu().numpy()[want], np.arange(K, dtype=np.int32))


@pytest.mark.parametrize("K", [300, 5000])
def test_argsort_desc_with_nan_and_inf_is_a_permutation(ops, dev, K):
    """A non-finite activation can reach diag(H).  NaN orders as the largest value (as torch.argsort(descending=True)
    places it), ties and NaNs keep ascending index: perm is a permutation for ANY input, on the one-pass kernel
    (K <= 1024) and on the 2-d counting kernels (K > 1024) alike."""
    g = torch.Generator(device=dev).manual_seed(K)
    v = torch.randn(K, generator=g, device=dev)
    v[3] = float("nan")
    v[K - 2] = float("nan")
    v[10] = float("inf")
    v[11] = float("-inf")
    v[20:30] = 0.5
    perm, inv = ops.argsort_desc(v)
    torch.cuda.synchronize()
    p, iv = perm.cpu().numpy(), inv.cpu().numpy()
    assert sorted(p.tolist()) == list(range(K))
    np.testing.assert_array_equal(iv[p], np.arange(K, dtype=np.int32))
    assert p[0] == 3 and p[1] == K - 2 and p[2] == 10 and p[-1] == 11
    want = torch.argsort(v, descending=True, stable=True).cpu().numpy()
    np.testing.assert_array_equal(p, want.astype(np.int32))


def _check_factor(ops, oracle, dev, K):
    xb = synth_activations(4 * K, K, seed=K + 1)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), 8)
    Hd, _, _ = oracle.hessian_dead_and_damp(H, 0.01)
    A = torch.from_numpy(np.ascontiguousarray(Hd[::-1, ::-1])).to(dev)
    U, info = ops.cholesky_inverse_upper(A)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    got = U.cpu().numpy()
    assert np.all(np.tril(got, -1) == 0), "strict lower triangle must be zero-filled"
    truth = oracle.cholesky_inverse_upper_f64(Hd)
    U_lapack, ok = oracle.cholesky_inverse_upper_lapack(Hd)
    assert ok
    scale = np.abs(truth).max()
    err_gpu = np.abs(got - truth).max() / scale
    err_lapack = np.abs(U_lapack - truth).max() / scale
    # The HIP path (one factorisation + one triangular inverse) must be at least as close to the
    # fp64 factor as upstream's own fp32 LAPACK three-step sequence, within a factor of 4.
    assert err_gpu <= max(4 * err_lapack, 5e-6), (err_gpu, err_lapack)
    # and it must actually factor H^-1: U^T U Hd ~= I
    resid = got.astype(np.float64).T @ got.astype(np.float64) @ Hd.astype(np.float64) - np.eye(K)
    assert np.abs(resid).max() < 5e-3
    return got, err_gpu, err_lapack


@pytest.mark.parametrize("K", [64, 128, 200, 384, 1024, 2048])
def test_cholesky_inverse_upper(ops, oracle, dev, K):
    _check_factor(ops, oracle, dev, K)


@pytest.mark.parametrize("K", [768, 1000, 2048])
def test_cholesky_inverse_upper_bf16x3_products(ops, oracle, dev, K, monkeypatch):
    """The long-k block-row products on the bf16 MFMA (three bf16 planes per operand, csrc/gemm3_tn.hip) are
    only chosen for large K; QT_CHOL_G3_MIN_CHUNKS=1 makes every step with a product take them (left-looking
    outer level), at sizes the oracle still factors in seconds.  Same accuracy bar as the f32-MFMA chain."""
    monkeypatch.setenv("QT_CHOL_G3", "0")
    base, e0, _ = _check_factor(ops, oracle, dev, K)
    monkeypatch.setenv("QT_CHOL_G3", "1")
    monkeypatch.setenv("QT_CHOL_G3_MIN_CHUNKS", "1")
    got, e1, el = _check_factor(ops, oracle, dev, K)
    print(f"K={K}: max error / max|U| vs fp64: f32 chain {e0:.2e}, bf16x3 products {e1:.2e}, fp32 LAPACK {el:.2e}")
    assert not np.array_equal(got, base), "the bf16x3 path was not taken"


def test_cholesky_bf16x3_products_wide_dynamic_range(ops, oracle, dev, monkeypatch):
    """Massive-activation channels (x1000 on 2 % of the channels, x0.01 on another 2 %: diag(H) spans ten
    decades) through the bf16x3 chain: a bf16 plane keeps 8 significant bits of EVERY element whatever its
    magnitude, so the factor stays as close to the fp64 factor as the f32-MFMA chain's."""
    K = 1536
    rng = np.random.default_rng(5)
    X = rng.standard_normal((4 * K, K)).astype(np.float32)
    big = rng.choice(K, size=K // 50, replace=False)
    small = rng.choice(np.setdiff1d(np.arange(K), big), size=K // 50, replace=False)
    X[:, big] *= 1000.0
    X[:, small] *= 0.01
    xb = oracle.f32_to_bf16_bits(X)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), 8)
    Hd, _, _ = oracle.hessian_dead_and_damp(H, 0.01)
    truth = oracle.cholesky_inverse_upper_f64(Hd)
    scale = np.abs(truth).max()
    errs = {}
    for name, env in (("f32", {"QT_CHOL_G3": "0"}), ("bf16x3", {"QT_CHOL_G3": "1", "QT_CHOL_G3_MIN_CHUNKS": "1"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        A = torch.from_numpy(np.ascontiguousarray(Hd[::-1, ::-1])).to(dev)
        U, info = ops.cholesky_inverse_upper(A)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        got = U.cpu().numpy().astype(np.float64)
        errs[name] = np.abs(got - truth).max() / scale
        # element-wise too: every entry within 1e-4 of its own magnitude or 1e-6 of the largest one
        assert np.all(np.abs(got - truth) <= 1e-4 * np.abs(truth) + 1e-6 * scale), name
    print(f"max error / max|U| vs fp64: f32 chain {errs['f32']:.2e}, bf16x3 products {errs['bf16x3']:.2e}")
    assert errs["bf16x3"] <= max(4 * errs["f32"], 5e-6)


@pytest.mark.parametrize("K", [200, 1000, 2048])
def test_cholesky_diagonal_block_inverses_on_the_mfma(ops, oracle, dev, K, monkeypatch):
    """The 128x128 diagonal-block inverses come from `trinv_mfma_kernel` (blocked substitution with potf2's 32x32
    inverses); `QT_CHOL_TRINV=div` selects the divide-and-update kernel of rounds 1-3.  Both must meet the accuracy
    bar of `_check_factor` (ragged last block included: K = 200, 1000) and agree with each other to fp32 rounding."""
    monkeypatch.setenv("QT_CHOL_TRINV", "div")
    base, e0, _ = _check_factor(ops, oracle, dev, K)
    monkeypatch.delenv("QT_CHOL_TRINV")
    got, e1, el = _check_factor(ops, oracle, dev, K)
    print(f"K={K}: max error / max|U| vs fp64: divide-and-update {e0:.2e}, MFMA blocks {e1:.2e}, fp32 LAPACK {el:.2e}")
    assert not np.array_equal(got, base), "the MFMA kernel was not taken"
    assert np.abs(got - base).max() <= 2e-5 * np.abs(base).max()


@pytest.mark.parametrize("K", [128, 384, 1024])
def test_cholesky_ignores_the_strict_lower_triangle(ops, oracle, dev, K):
    """`qt_hessian_prepare` writes the upper triangle of the flipped matrix only; whatever the allocation held
    below it -- NaN and Inf included -- must not reach the factor (a masked MFMA lane multiplies by zero, and
    0 * NaN is NaN: the rank-1 steps of potf2 mask both operands)."""
    xb = synth_activations(4 * K, K, seed=K + 7)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), 8)
    Hd, _, _ = oracle.hessian_dead_and_damp(H, 0.01)
    A0 = torch.from_numpy(np.ascontiguousarray(Hd[::-1, ::-1])).to(dev)
    U0, info0 = ops.cholesky_inverse_upper(A0.clone())
    poison = torch.full_like(A0, float("nan"))
    poison[::3] = float("inf")
    A1 = torch.triu(A0) + torch.tril(poison, -1)
    U1, info1 = ops.cholesky_inverse_upper(A1)
    torch.cuda.synchronize()
    assert int(info0.item()) == 0 and int(info1.item()) == 0
    assert torch.equal(U0, U1)


def test_cholesky_reports_non_pd(ops, dev):
    K = 256
    A = torch.eye(K, dtype=torch.float32, device=dev)
    A[200, 200] = -1.0
    U, info = ops.cholesky_inverse_upper(A)
    torch.cuda.synchronize()
    assert int(info.item()) == 201  # 1-based position in the flipped matrix


# ------------------------------------------------------------------------------------- a10
@pytest.mark.parametrize("R,K,gs,sym,dtype", [(16, 256, 128, True, "f32"), (33, 384, 128, False, "f32"),
                                             (8, 512, -1, True, "f32"), (16, 256, 64, True, "bf16")])
def test_qparams_exact(ops, oracle, dev, R, K, gs, sym, dtype):
    Wn = synth_weight(R, K, seed=R)
    if dtype == "bf16":
        bits = oracle.f32_to_bf16_bits(Wn)
        Wn = oracle.bf16_bits_to_f32(bits)
        W = bits_to_bf16_tensor(bits, dev)
    else:
        W = torch.from_numpy(Wn).to(dev)
    scale, zp, scale_t, zp_t = ops.group_minmax_qparams(W, gs, sym, 4)
    torch.cuda.synchronize()
    os_, oz = oracle.minmax_qparams(Wn, gs, sym, 4)
    np.testing.assert_array_equal(scale.cpu().numpy(), os_)
    np.testing.assert_array_equal(zp.cpu().numpy(), oz)
    np.testing.assert_array_equal(scale_t.cpu().numpy(), os_.T)
    np.testing.assert_array_equal(zp_t.cpu().numpy(), oz.T)


def test_weight_gather(ops, dev):
    R, K = 20, 136
    W = torch.randn(R, K, device=dev).to(torch.bfloat16)
    perm = torch.randperm(K, device=dev).to(torch.int32)
    dead = torch.zeros(K, dtype=torch.uint8, device=dev)
    dead[5] = 1
    out = ops.weight_gather_f32(W, perm, dead)
    torch.cuda.synchronize()
    want = W.float()[:, perm.long()]
    want[:, 5] = 0
    assert torch.equal(out, want)


# ------------------------------------------------------------------------------------- a11
def _sweep_case(oracle, R, K, gs, sym, actorder, seed):
    Wn = synth_weight(R, K, seed=seed)
    xb = synth_activations(2 * K, K, seed=seed + 1)
    H = oracle.hessian_from_gram(oracle.gram_f64(xb), 4)
    return Wn, H


@pytest.mark.parametrize("R,K,gs,sym", [(16, 128, 128, True), (64, 256, 128, True), (130, 384, 128, True),
                                        (200, 640, 128, False), (96, 200, -1, True), (257, 512, 64, True),
                                        # 9 blocks / 8 blocks + a ragged one: several far updates that carry
                                        # more than one block's chain in a single pass over W (sweep.hip)
                                        (40, 1152, 128, True), (48, 1100, -1, True)])
def test_sweep_bit_exact_given_same_U(ops, oracle, dev, R, K, gs, sym):
    Wn, H = _sweep_case(oracle, R, K, gs, sym, None, seed=R + K)
    Hd, dead, _ = oracle.hessian_dead_and_damp(H)
    Un = oracle.cholesky_inverse_upper_f64(Hd).astype(np.float32)
    scale, zp = oracle.minmax_qparams(Wn, gs, sym, 4)
    rng = np.random.default_rng(0)
    G = scale.shape[1]
    g_idx = (np.arange(K) // (K if gs <= 0 else gs)).astype(np.int32)
    g_idx = g_idx[rng.permutation(K)]  # as under activation ordering
    Qo, Wo, lo = oracle.gptq_sweep_c(Wn, Un, scale, zp, g_idx, 128, 4)

    W = torch.from_numpy(Wn.copy()).to(dev)
    U = torch.from_numpy(Un).to(dev)
    Qt, loss = ops.gptq_sweep(W, U, torch.from_numpy(np.ascontiguousarray(scale.T)).to(dev),
                              torch.from_numpy(np.ascontiguousarray(zp.T)).to(dev),
                              torch.from_numpy(g_idx).to(dev), 128, 4)
    torch.cuda.synchronize()
    q = Qt.cpu().numpy().T
    assert np.array_equal(q, Qo), f"{(q != Qo).sum()} of {q.size} levels differ"
    np.testing.assert_array_equal(W.cpu().numpy(), Wo)  # dequantised weights, bit-exact
    np.testing.assert_allclose(loss.cpu().numpy(), lo, rtol=1e-6)


# ------------------------------------------------------------------------------------- a14
@pytest.mark.parametrize("R,K,perm", [(16, 64, False), (70, 264, True), (64, 4096, True), (5, 20, False), (300, 520, True), (516, 36, False)])
def test_pack_and_dequant_exact(ops, oracle, dev, R, K, perm):
    rng = np.random.default_rng(R * K)
    Q = rng.integers(-8, 8, size=(R, K)).astype(np.int8)  # levels in ORIGINAL column order
    col_src_np = rng.permutation(K).astype(np.int32) if perm else None
    # sweep-position-major storage: position col_src[c] holds original column c
    Qpos = np.empty((K, R), np.int8)
    src = col_src_np if perm else np.arange(K)
    Qpos[src] = Q.T
    Qt = torch.from_numpy(Qpos).to(dev)
    cs = torch.from_numpy(col_src_np).to(dev) if perm else None
    packed = ops.pack_int4(Qt, cs)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(packed.cpu().numpy(), oracle.pack_int4(Q))
    assert np.array_equal(oracle.unpack_int4(packed.cpu().numpy(), K), Q)
    gs = 4 if K % 4 == 0 else K
    G = K // gs
    scale = (rng.random((R, G)).astype(np.float32) + 0.1)
    zp = rng.integers(-8, 8, size=(R, G)).astype(np.float32)
    g_of_col = (np.arange(K) // gs).astype(np.int32)
    out = ops.dequantize(Qt, torch.from_numpy(scale).to(dev), torch.from_numpy(zp).to(dev),
                         torch.from_numpy(g_of_col).to(dev), cs, torch.float32)
    torch.cuda.synchronize()
    want = (Q.astype(np.float32) - zp[:, g_of_col]) * scale[:, g_of_col]
    np.testing.assert_array_equal(out.cpu().numpy(), want)


# ------------------------------------------------------------------------------------- a12 building block
@pytest.mark.parametrize("n,K,dtype", [(256, 256, "bf16"), (1024, 1000, "bf16"), (4096, 2048, "f16"), (640, 4096, "bf16")])
def test_xtx_dot_is_the_frobenius_product_with_the_gram_matrix(ops, dev, n, K, dtype):
    """qt_xtx_dot = scale * <H, X^T X>_F straight from the Gram kernel's accumulators (every work item --
    whole tile or token chunk -- contributes <H tile, its partial tile>): against fp64 on the same 16-bit
    X and fp32 H.  Floating point: relative error bound 2e-6 (fp32 tile accumulators over <= 4096 tokens,
    fp64 everywhere after that); accumulate=True adds to the output; run-to-run identical."""
    g = torch.Generator(device=dev).manual_seed(n + K)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    X = torch.randn((n, K), generator=g, device=dev).to(td)
    A = torch.randn((K, K), generator=g, device=dev)
    H = (A + A.t()).contiguous()
    H_lower_only = torch.tril(H) + torch.triu(torch.full_like(H, float("nan")), 1)   # the upper triangle is never read
    out = ops.xtx_dot(X, H_lower_only, scale=0.5)
    out2 = ops.xtx_dot(X, H_lower_only, scale=0.5)
    torch.cuda.synchronize()
    Xd = X.double()
    want = 0.5 * float(((Xd.t() @ Xd) * H.double()).sum())
    scale_ref = 0.5 * float(((Xd.t() @ Xd).abs() * H.double().abs()).sum())
    assert abs(float(out.item()) - want) <= 2e-6 * scale_ref
    assert torch.equal(out, out2)
    acc = out.clone()
    ops.xtx_dot(X, H_lower_only, scale=0.5, out=acc, accumulate=True)
    torch.cuda.synchronize()
    assert abs(float(acc.item()) - 2 * float(out.item())) <= 1e-6 * abs(float(out.item())) + 1e-30


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("K,gs,sym,bits", [(512, 128, True, 4), (1152, 128, False, 4), (200, -1, True, 8), (4096, 128, True, 4)])
def test_fused_gather_qparams_equals_the_two_passes(ops, dev, dtype, K, gs, sym, bits):
    """``qt_weight_gather_qparams`` (one read of W) against ``qt_group_minmax_qparams`` + ``qt_weight_gather_f32``: the
    same bits everywhere, also when the group-major tables are a column range of wider ones."""
    g = torch.Generator(device=dev).manual_seed(K + bits)
    R = 72
    W = (torch.randn((R, K), generator=g, device=dev) * 0.05).to(dtype)
    W[:, 5] = 0
    perm = torch.randperm(K, generator=g, device=dev).to(torch.int32)
    dead = (torch.rand(K, generator=g, device=dev) < 0.02).to(torch.uint8)
    for pm, dd in ((perm, dead), (None, None)):
        s0, z0, st0, zt0 = ops.group_minmax_qparams(W, gs, sym, bits)
        w0 = ops.weight_gather_f32(W, pm, dd)
        G = s0.shape[1]
        out = torch.empty((R, K), dtype=torch.float32, device=dev)
        s1, z1 = torch.empty_like(s0), torch.empty_like(z0)
        wide_s = torch.full((G, R + 56), -1.0, device=dev)
        wide_z = torch.full((G, R + 56), -1.0, device=dev)
        ops.weight_gather_qparams(W, pm, dd, gs, sym, bits, out=out, scale=s1, zp=z1, scale_t=wide_s[:, 24:24 + R],
                                  zp_t=wide_z[:, 24:24 + R])
        torch.cuda.synchronize()
        assert torch.equal(out, w0) and torch.equal(s1, s0) and torch.equal(z1, z0)
        assert torch.equal(wide_s[:, 24:24 + R], st0) and torch.equal(wide_z[:, 24:24 + R], zt0)
        assert bool((wide_s[:, :24] == -1).all()) and bool((wide_s[:, 24 + R:] == -1).all())
