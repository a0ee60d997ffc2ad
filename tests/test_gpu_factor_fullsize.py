"""The factor U = chol(Hd^-1, upper) AT THE SIZES THE bf16x3 CHAIN SHIPS AT (K = 8192 and 14336), and
the end-to-end nibble-mismatch rate at those sizes against the oracle's own fp32 LAPACK factor.

Upstream step: ``cholesky -> cholesky_inverse -> cholesky(upper)``, reached through
``/root/reference/src/quantool/methods/llm_compressor/gptq/gptq.py:86`` -> ``base.py:161``.

Bars
  * factor: max |U_gpu - U_f64| / max|U_f64| <= max(4 x the same error of the fp32 LAPACK three-step, 5e-6)
    -- the bar of ``test_gpu_kernels._check_factor`` -- for the DEFAULT path (three-plane bf16 block-row
    products live at these K) and for the f32-MFMA chain (QT_CHOL_G3=0) beside it;
  * end to end: a 256-row slice swept by the GPU with its own factor vs the oracle's C sweep with the
    LAPACK factor of the same damped Hessian: scales bit-exact; the nibble mismatch rate is printed and
    bounded at ~10x the observation (error feedback turns last-bit differences of U into flipped
    roundings; upstream itself is not reproducible across BLAS builds at that level).

One fp64 and one fp32 LAPACK factorisation per K on the host (K = 14336: about half a minute on the
GPU box's 16 cores), shared by the tests of a K through a module-scoped fixture.
"""
import os

import numpy as np
import pytest
import torch

from .test_gpu_fullsize_oracle import _gpu_run

pytestmark = pytest.mark.gpu

# K -> rows of the slice, calibration samples of 384 tokens, seed
CASES = {
    8192: dict(R=256, n_samples=96, seed=23),
    14336: dict(R=256, n_samples=128, seed=29),
}
# observed rates; the asserted bound is 10x the observation, floor 2e-5
OBSERVED = {8192: 0.0, 14336: 1.25e-4}     # round 3, 256-row slices: 0 of 2 097 152 and 459 of 3 670 016 nibbles


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request, dev, oracle):
    K = request.param
    c = CASES[K]
    for v in ("QT_CHOL_G3", "QT_CHOL_G3_MIN_CHUNKS"):
        assert v not in os.environ, f"{v} is set: this test pins the DEFAULT factorisation path"
    Wf, res, keep, Gfull, n = _gpu_run(dev, c["R"], K, n_samples=c["n_samples"], T=384, seed=c["seed"])
    H = oracle.hessian_from_gram_f32(Gfull, n)
    del Gfull
    perm = keep["perm"].cpu().numpy().astype(np.int64)
    Hp = H[perm][:, perm]
    Hd, dead, _ = oracle.hessian_dead_and_damp(Hp, 0.01)
    del Hp
    assert not dead.any()
    truth = oracle.cholesky_inverse_upper_f64_lapack(Hd)
    U_lapack, ok = oracle.cholesky_inverse_upper_lapack(Hd)
    assert ok
    return dict(K=K, Wf=Wf, res=res, keep=keep, H=H, Hd=Hd, truth=truth, U_lapack=U_lapack, n=n)


def _err(U, truth):
    # chunked: a K x K fp64 temporary per operand is 1.6 GB at K = 14336
    m = 0.0
    for r0 in range(0, U.shape[0], 2048):
        m = max(m, float(np.abs(U[r0:r0 + 2048] - truth[r0:r0 + 2048]).max()))
    return m / float(np.abs(truth).max())


def test_default_factor_vs_fp64_and_lapack(case, ops, dev):
    """The factor the sweep actually used (default knobs: bf16x3 block-row products on)."""
    K = case["K"]
    U = case["keep"]["U"].cpu().numpy()
    assert np.all(np.tril(U, -1) == 0)
    e_gpu = _err(U, case["truth"])
    e_lap = _err(case["U_lapack"], case["truth"])
    print(f"\n[factor] K={K} default path: max err / max|U| vs fp64 = {e_gpu:.3e}; fp32 LAPACK three-step = {e_lap:.3e}")
    assert e_gpu <= max(4 * e_lap, 5e-6), (e_gpu, e_lap)


def test_f32_chain_factor_beside_it(case, ops, dev, monkeypatch):
    """QT_CHOL_G3=0 (every product on the f32 MFMA) on the same damped Hessian: same bar, and the default
    path must really have been a different computation (i.e. the bf16x3 products were live)."""
    K = case["K"]
    monkeypatch.setenv("QT_CHOL_G3", "0")
    A = torch.from_numpy(np.ascontiguousarray(case["Hd"][::-1, ::-1])).to(dev)
    U32, info = ops.cholesky_inverse_upper(A)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    U32 = U32.cpu().numpy()
    e32 = _err(U32, case["truth"])
    e_lap = _err(case["U_lapack"], case["truth"])
    print(f"\n[factor] K={K} f32 chain: max err / max|U| vs fp64 = {e32:.3e}; fp32 LAPACK three-step = {e_lap:.3e}")
    assert e32 <= max(4 * e_lap, 5e-6), (e32, e_lap)
    assert not np.array_equal(U32, case["keep"]["U"].cpu().numpy()), "the default path did not take the bf16x3 products"
    # the oracle's damped matrix through the default path (the sweep's factor came from hessian_prepare's
    # own damped matrix: same values up to 1 ulp on the diagonal): same bar
    monkeypatch.delenv("QT_CHOL_G3")
    A = torch.from_numpy(np.ascontiguousarray(case["Hd"][::-1, ::-1])).to(dev)
    U3, _ = ops.cholesky_inverse_upper(A)
    torch.cuda.synchronize()
    e3 = _err(U3.cpu().numpy(), case["truth"])
    print(f"[factor] K={K} default path on the oracle's damped matrix: {e3:.3e}")
    assert e3 <= max(4 * e_lap, 5e-6), (e3, e_lap)


def test_end_to_end_nibble_mismatch_rate_vs_lapack_factor_oracle(case, oracle):
    K = case["K"]
    res = case["res"]
    o = oracle.quantize_weight(case["Wf"], case["H"], actorder="static", U_override=case["U_lapack"])
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o["scale"])
    q_gpu = oracle.unpack_int4(res.weight_packed.cpu().numpy(), K)
    mism = int((q_gpu != o["q"]).sum())
    rate = mism / q_gpu.size
    # how far apart are two CPU factors of the same matrix? (fp64 factor rounded to fp32 vs fp32 LAPACK)
    o64 = oracle.quantize_weight(case["Wf"], case["H"], actorder="static", U_override=case["truth"].astype(np.float32))
    rate64 = float((o64["q"] != o["q"]).mean())
    print(f"\n[fullsize] K={K} R={q_gpu.shape[0]}: nibble mismatch rate GPU vs LAPACK-factor oracle = {rate:.3e} "
          f"({mism} of {q_gpu.size}); fp64-factor oracle vs LAPACK-factor oracle = {rate64:.3e}")
    obs = OBSERVED[K]
    bound = 1e-3 if obs is None else max(10 * obs, 2e-5)
    assert rate <= bound, (rate, bound)
    # and never worse than a few times the distance between two CPU factorisations of the same matrix
    assert rate <= max(5 * rate64, 2e-5), (rate, rate64)
