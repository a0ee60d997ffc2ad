"""fp32-accurate TN product on the bf16 MFMA (csrc/gemm3_tn.hip: three bf16 planes per operand, six plane
products) against an fp64 product of the same fp32 inputs.  Floating point: the bound is stated per test
(a few fp32 ulps of sum_k |a||b|, the same scale an fp32 fmaf chain is bounded by)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24


@pytest.fixture(scope="module")
def ops(dev):
    from quantool_amd.hip import ops as _ops

    return _ops


def _inputs(k, M, N, seed, spread=6.0):
    rng = np.random.default_rng(seed)
    # wide dynamic range (exponents spread over 2^-spread .. 2^spread), signs mixed
    A = (rng.standard_normal((k, M)) * np.exp2(rng.uniform(-spread, spread, (k, M)))).astype(np.float32)
    B = (rng.standard_normal((k, N)) * np.exp2(rng.uniform(-spread, spread, (k, N)))).astype(np.float32)
    return A, B


def _bound(A, B):
    return np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)


@pytest.mark.parametrize("k,M,N", [(256, 512, 512), (128, 264, 520), (384, 256, 1024)])
def test_gemm3_sub_whole_tiles(ops, dev, k, M, N):
    A, B = _inputs(k, M, N, seed=k + M + N)
    C0 = np.random.default_rng(1).standard_normal((M, N)).astype(np.float32)
    C = torch.from_numpy(C0.copy()).to(dev)
    ops.gemm3_tn(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), C, kind=0)
    torch.cuda.synchronize()
    ref = C0.astype(np.float64) - A.astype(np.float64).T @ B.astype(np.float64)
    err = np.abs(C.cpu().numpy().astype(np.float64) - ref)
    # 16 eps of sum|a||b|: a bf16 plane keeps 8 significant bits, so three planes leave <= 2^-24 |x| per
    # operand in the worst case, the dropped products (mid*lo, lo*mid) another 2 * 2^-24, the rest is the
    # MFMA's fp32 accumulation; observed ~10 eps on these wide-range inputs, ~1 eps on normal ones
    tol = 16 * EPS * _bound(A, B) + 2 * EPS * np.abs(ref)
    assert (err <= tol).all(), f"max err / tol = {(err / tol).max():.3f}"


@pytest.mark.parametrize("k,M,N", [(1024, 256, 768), (2048, 256, 512)])
def test_gemm3_set_split_k(ops, dev, k, M, N):
    A, B = _inputs(k, M, N, seed=7 + k, spread=3.0)
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm3_tn(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), C, kind=1)
    C2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm3_tn(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), C2, kind=1)
    torch.cuda.synchronize()
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    err = np.abs(C.cpu().numpy().astype(np.float64) - ref)
    tol = 16 * EPS * _bound(A, B) + 2 * EPS * np.abs(ref)
    assert (err <= tol).all(), f"max err / tol = {(err / tol).max():.3f}"
    assert torch.equal(C, C2)        # slabs are reduced in table order: run-to-run identical


def test_gemm3_error_is_fp32_class(ops, dev):
    """Same inputs through the f32-MFMA fmaf chain (sgemm_tn) and through the split product: the split
    product's error against fp64 is of the same size (it is what replaces sgemm_tn in the Cholesky chain)."""
    k, M, N = 512, 512, 512
    A, B = _inputs(k, M, N, seed=3, spread=2.0)
    At, Bt = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    C = torch.zeros((M, N), dtype=torch.float32, device=dev)
    ops.gemm3_tn(At, Bt, C, kind=0)
    S = ops.sgemm_tn(At, Bt, None, 1)
    torch.cuda.synchronize()
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    e3 = np.abs(-C.cpu().numpy().astype(np.float64) - ref)
    es = np.abs(S.cpu().numpy().astype(np.float64) - ref)
    scale = _bound(A, B)
    r3, rs = float((e3 / scale).max()), float((es / scale).max())
    print(f"max error / sum|a||b|: gemm3 {r3 / EPS:.2f} eps, sgemm {rs / EPS:.2f} eps")
    assert r3 <= max(4 * rs, 2 * EPS)
