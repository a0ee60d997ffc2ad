"""The literal north-star comparison, host side: "outputs match the reference CPU path ON THE SAME CALIBRATION
INPUTS" -- the oracle gets X and W and nothing else, forms ITS OWN Hessian with upstream's per-sample fp32
running update (``oracle.accumulate_hessian_reference``; SURVEY A.2 ``accumulate_hessian``, reached through
``/root/reference/src/quantool/methods/llm_compressor/base.py:161``), factorises it with the fp32 LAPACK
three-step and runs the C sweep.

Error feedback turns any last-bit difference of H or U into a flipped level somewhere in a row, so "how many
nibbles differ" only reads against a yardstick.  Two CPU-vs-CPU yardsticks are computed on the same inputs:

  (i)  H-ORDER: the same pipeline fed ``hessian_from_gram(gram_f64(X))`` -- the exactly rounded Hessian -- instead
       of the S-rank-T-updates-in-fp32 one.  Both are legitimate fp32 evaluations of (2/n) sum_b X_b^T X_b;
       upstream's own result moves this much with the BLAS build / thread count that sums ``x.T @ x``.
  (ii) FACTOR: the per-sample Hessian through the fp64 three-step rounded to fp32 instead of fp32 LAPACK.

Under activation ordering (upstream's default, ``actorder="static"``) a third effect sits on top: the sweep order is
``argsort(diag H)``, and channels whose diagonal entries agree to the last bits swap places with the summation order
of H.  ``Side.run(perm=...)`` fixes the order, so that effect can be counted (positions where two orders differ) and
taken out (both sides swept in one order).

Shared by ``tests/test_gpu_own_hessian_parity.py`` (the GPU column), ``tests/test_own_hessian_yardsticks.py``
(CPU: the yardstick code itself at the small size); DESIGN.md section 2's table is the GPU test's printed output
(``profiles/r04_own_hessian_parity.txt``).
"""
from __future__ import annotations

import numpy as np

# name -> R, K, calibration samples S of T tokens, seed.  S * T >= K everywhere (a full-rank Gram sum).
CASES = {
    "128x512": dict(R=128, K=512, S=8, T=96, seed=41),
    "4096x4096": dict(R=4096, K=4096, S=32, T=384, seed=43),
    "256x14336": dict(R=256, K=14336, S=40, T=384, seed=47),
}


def make_inputs(oracle, case: dict):
    """(W fp32 with bf16-representable values [R, K], its bf16 bits, X bf16 bits [S, T, K]).  Activations
    N(0, 1) with 1 % of the channels x10, weights N(0, 0.02^2)."""
    R, K, S, T = case["R"], case["K"], case["S"], case["T"]
    rng = np.random.default_rng(case["seed"])
    wb = oracle.f32_to_bf16_bits((rng.standard_normal((R, K)) * 0.02).astype(np.float32))
    gain = np.ones(K, np.float32)
    gain[rng.choice(K, size=max(1, K // 100), replace=False)] = 10.0
    xb = np.empty((S, T, K), np.uint16)
    for s in range(S):
        xb[s] = oracle.f32_to_bf16_bits(rng.standard_normal((T, K)).astype(np.float32) * gain)
    return oracle.bf16_bits_to_f32(wb), wb, xb


def gram_f64_chunked(oracle, xb: np.ndarray) -> np.ndarray:
    S, T, K = xb.shape
    G = np.zeros((K, K), np.float64)
    for s0 in range(0, S, 8):       # a 40 x 384 x 14336 fp64 temporary would be 1.7 GB
        x = oracle.bf16_bits_to_f32(xb[s0:s0 + 8].reshape(-1, K)).astype(np.float64)
        G += x.T @ x
    return G


def rel_diff(Ha: np.ndarray, Hb: np.ndarray) -> float:
    """max |Ha - Hb|_ij / sqrt(Hb_ii Hb_jj), in row chunks (fp64 temporaries)."""
    d = np.sqrt(np.diag(Hb).astype(np.float64))
    m = 0.0
    for r0 in range(0, Ha.shape[0], 1024):
        sl = slice(r0, r0 + 1024)
        m = max(m, float((np.abs(Ha[sl].astype(np.float64) - Hb[sl]) / np.outer(d[sl], d)).max()))
    return m


class Side:
    """The oracle on (X, W) alone.  ``H_own`` = upstream's per-sample fp32 running update; ``H_g64`` = the exactly
    rounded Hessian (yardstick i), formed on first use."""

    def __init__(self, oracle, Wf: np.ndarray, xb: np.ndarray):
        self.oracle, self.Wf, self.xb = oracle, Wf, xb
        self.S, self.T, self.K = xb.shape
        self.H_own = oracle.accumulate_hessian_reference((xb[s] for s in range(self.S)), self.K)
        self._H_g64 = None

    @property
    def H_g64(self) -> np.ndarray:
        if self._H_g64 is None:
            self._H_g64 = self.oracle.hessian_from_gram(gram_f64_chunked(self.oracle, self.xb), self.S)
        return self._H_g64

    def run(self, H=None, actorder="static", perm=None, factor="lapack") -> dict:
        """quantize_weight on ``H`` (default: the own Hessian).  ``factor``: "lapack" (fp32 potrf / potri / potrf,
        what torch-CPU calls) or "f64" (the fp64 three-step rounded to fp32: yardstick ii)."""
        orc = self.oracle
        H = self.H_own if H is None else H
        if factor == "lapack":
            o = orc.quantize_weight(self.Wf, H, actorder=actorder, inverse="lapack", perm_override=perm)
        else:
            if actorder is None:
                p = np.arange(self.K)
            else:
                p = np.argsort(-np.diag(H), kind="stable") if perm is None else np.asarray(perm, np.int64)
            Hd = orc.hessian_dead_and_damp(H[p][:, p], 0.01)[0]
            U64 = orc.cholesky_inverse_upper_f64_lapack(Hd).astype(np.float32)
            o = orc.quantize_weight(self.Wf, H, actorder=actorder, U_override=U64, perm_override=perm)
        assert o["ok"] and not o["dead"].any()
        return o

    def yardsticks(self, actorder="static", perm=None, o_own=None, with_factor=True) -> dict:
        """Rates (fraction of differing integer levels) between CPU pipelines on the same inputs: ``h_order`` (i) and
        ``factor`` (ii); ``perm_flips`` = sweep positions at which the two Hessians' orders differ (0 with ``perm``)."""
        o = o_own if o_own is not None else self.run(actorder=actorder, perm=perm)
        o_g = self.run(H=self.H_g64, actorder=actorder, perm=perm)
        o_f = self.run(actorder=actorder, perm=perm, factor="f64") if with_factor else None
        flips = 0 if o["perm"] is None else int((np.asarray(o["perm"]) != np.asarray(o_g["perm"])).sum())
        return dict(o=o, h_order=float((o_g["q"] != o["q"]).mean()), factor=float((o_f["q"] != o["q"]).mean()) if with_factor else None,
                    perm_flips=flips, scales_equal=bool(np.array_equal(o_g["scale"], o["scale"])))


def nibble_rate(oracle, packed: np.ndarray, q_ref: np.ndarray) -> tuple:
    """(rate, differing levels, rows with at least one) of packed int4 words against reference levels."""
    q = oracle.unpack_int4(packed, q_ref.shape[1])
    ne = q != q_ref
    return float(ne.mean()), int(ne.sum()), int(ne.any(axis=1).sum())
