"""GPU vs the CPU oracle ON THE SAME CALIBRATION INPUTS, the oracle forming its OWN Hessian (north_star: "Outputs match
the reference CPU path on the same calibration inputs").  Both sides get ``X`` and ``W`` and nothing else:

    oracle : ``accumulate_hessian_reference`` (upstream's per-sample fp32 running update; SURVEY A.2, reached through
             ``/root/reference/src/quantool/methods/llm_compressor/base.py:161``) -> ``quantize_weight(inverse="lapack")``
             -> ``pack_int4``
    GPU    : ``HessianAccumulator.add`` per sample -> ``gptq_quantize_shared`` (qt_xtx_accumulate, qt_hessian_prepare,
             qt_cholesky_inverse_upper, qt_gptq_sweep, qt_pack_int4 through the C ABI)

Every other oracle comparison in this suite hands the oracle the GPU's Gram sum (and mostly its factor) and demands bit
equality; here nothing is shared, so the comparison is a RATE, read next to two CPU-vs-CPU yardsticks on the same inputs
(``tests/own_hessian_cases.py``): (i) the oracle fed the exactly rounded Hessian instead of its per-sample fp32 one,
(ii) the oracle with the fp64 factor instead of fp32 LAPACK.

Bars
  * scales (and the absence of zero points / g_idx for the symmetric static scheme) bit-equal: they depend on W alone;
  * actorder = None: rate(GPU vs oracle) <= RATIO x max(yardstick i, yardstick ii), floor 3e-5 (measured: <= 2.5 x);
  * actorder = "static" (upstream's default): the sweep order is argsort(diag H), which flips for channels whose
    diagonals agree to the last bits -- between the GPU's and the oracle's Hessian exactly as between the oracle's two.
    The literal rate is printed and bounded by RATIO x the literal yardsticks when the GPU picked the oracle's order, else
    by 10 x; and with the order taken out (the oracle swept in the GPU's order, yardsticks likewise) the RATIO bound holds.

Host cost: the K = 14336 case forms one per-sample fp32 and one fp64 Hessian and factorises four times (about two to
three minutes on the GPU box's 16 cores).
"""
import numpy as np
import pytest
import torch

from . import own_hessian_cases as oc
from .util import bits_to_bf16_tensor

pytestmark = pytest.mark.gpu

FLOOR = 3e-5          # a handful of levels at the small sizes
# Asserted ratio to the yardsticks.  Measured (profiles/r04_own_hessian_parity.txt): <= 2.5 everywhere.  The CPU side --
# both the oracle's Hessian and the yardsticks -- depends on the BLAS build and thread count of the box that runs the
# test (that dependence is the very thing yardstick (i) measures), so the assertion leaves room above the 3x DESIGN.md
# quotes instead of failing on another host.
RATIO = 4.0
_SIDES = {}


def _side(oracle, name):
    if name not in _SIDES:
        _SIDES.clear()        # one case's Hessians at a time (K = 14336: 0.8 GB each)
        Wf, wb, xb = oc.make_inputs(oracle, oc.CASES[name])
        _SIDES[name] = (oc.Side(oracle, Wf, xb), wb, xb)
    return _SIDES[name]


def gpu_side(dev, wb, xb, actorder, symmetric=True):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    S, T, K = xb.shape
    acc = HessianAccumulator(K, dev)
    for s in range(S):                                   # the plugin path's calling pattern: one sample per call
        acc.add(bits_to_bf16_tensor(xb[s], dev))
    keep = {}
    res = gptq_quantize_shared([bits_to_bf16_tensor(wb, dev)], acc,
                               QuantArgs(num_bits=4, symmetric=symmetric, group_size=128, actorder=actorder), keep=keep)[0]
    torch.cuda.synchronize()
    assert acc.n == S and int(res.info.item()) == 0
    return res, keep, acc


@pytest.mark.parametrize("name,actorder", [("128x512", None), ("128x512", "static"), ("4096x4096", None),
                                           ("4096x4096", "static"), ("256x14336", "static")])
def test_gpu_vs_oracle_forming_its_own_hessian(dev, oracle, name, actorder):
    side, wb, xb = _side(oracle, name)
    K = side.K
    res, keep, acc = gpu_side(dev, wb, xb, actorder)

    # Hessians: the GPU's (2/n) G against the oracle's own, both against the exactly rounded one
    Gl = torch.tril(acc.G).cpu().numpy()
    H_gpu = oracle.hessian_from_gram_f32(Gl + np.tril(Gl, -1).T, acc.n)
    del Gl
    h_gpu, h_own = oc.rel_diff(H_gpu, side.H_g64), oc.rel_diff(side.H_own, side.H_g64)
    assert h_gpu <= 1e-5 and h_own <= 1e-5, (h_gpu, h_own)

    y = side.yardsticks(actorder=actorder)
    o = y["o"]
    # what depends on W alone is equal to the bit; the static scheme saves no g_idx, the symmetric one no zero points
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o["scale"])
    np.testing.assert_array_equal(res.zp_f32.cpu().numpy(), o["zp"])
    assert res.weight_g_idx is None and o["g_idx"] is None and res.weight_zero_point is None
    packed = res.weight_packed.cpu().numpy()
    rate, mism, rows = oc.nibble_rate(oracle, packed, o["q"])
    yard = max(y["h_order"], y["factor"])
    flips_gpu = 0
    if actorder is not None:
        perm_gpu = keep["perm"].cpu().numpy().astype(np.int64)
        flips_gpu = int((perm_gpu != np.asarray(o["perm"])).sum())
    print(f"\n[own-H] {name} actorder={actorder}: GPU vs oracle(own H, LAPACK) = {rate:.3e} ({mism} of {packed.size * 8}, "
          f"{rows} rows) | H-order yardstick {y['h_order']:.3e} | factor yardstick {y['factor']:.3e} | "
          f"|dH|/sqrt(HiiHjj): GPU {h_gpu:.1e}, oracle {h_own:.1e} | sweep-order flips: GPU {flips_gpu}, yardstick {y['perm_flips']}")
    assert y["scales_equal"]
    if flips_gpu == 0:
        assert rate <= max(RATIO * yard, FLOOR), (rate, y)
        return
    # the argsort of near-equal diagonals came out differently: literal rate bounded loosely, then the order taken out
    assert rate <= max(10 * yard, 0.2 if y["perm_flips"] == 0 else 0.0, FLOOR), (rate, y)
    # (the factor yardstick is not recomputed: one fp64 factorisation per order is the dearest step at K = 14336)
    y2 = side.yardsticks(actorder=actorder, perm=perm_gpu, with_factor=False)
    rate2, mism2, rows2 = oc.nibble_rate(oracle, packed, y2["o"]["q"])
    print(f"[own-H] {name} in the GPU's sweep order on both sides: GPU vs oracle = {rate2:.3e} ({mism2}, {rows2} rows) | "
          f"H-order yardstick {y2['h_order']:.3e}")
    assert rate2 <= max(RATIO * max(y2["h_order"], y["factor"]), FLOOR), (rate2, y2["h_order"], y["factor"])


@pytest.mark.parametrize("actorder,symmetric", [("group", True), ("static", False), ("group", False)])
def test_small_case_other_schemes_with_nothing_shared(dev, oracle, actorder, symmetric):
    """The same comparison for the schemes that save more than packed words and scales: ``actorder="group"`` keeps
    ``weight_g_idx`` (a function of the sweep order) and takes its scales from the PERMUTED matrix, the asymmetric preset
    keeps zero points.  At 128 x 512 the GPU's and the oracle's Hessians order the channels identically, and then every
    saved tensor is equal to the bit."""
    side, wb, xb = _side(oracle, "128x512")
    res, keep, acc = gpu_side(dev, wb, xb, actorder, symmetric)
    o = oracle.quantize_weight(side.Wf, side.H_own, actorder=actorder, symmetric=symmetric, inverse="lapack")
    assert np.array_equal(keep["perm"].cpu().numpy().astype(np.int64), np.asarray(o["perm"], np.int64))
    np.testing.assert_array_equal(res.scale_f32.cpu().numpy(), o["scale"])
    np.testing.assert_array_equal(res.zp_f32.cpu().numpy(), o["zp"])
    if actorder == "group":
        np.testing.assert_array_equal(res.weight_g_idx.cpu().numpy(), o["g_idx"])
    else:
        assert res.weight_g_idx is None and o["g_idx"] is None
    if not symmetric:
        np.testing.assert_array_equal(res.weight_zero_point.cpu().numpy(), o["zp"].astype(np.int8))
    rate, mism, rows = oc.nibble_rate(oracle, res.weight_packed.cpu().numpy(), o["q"])
    print(f"\n[own-H] 128x512 actorder={actorder} symmetric={symmetric}: GPU vs oracle(own H, LAPACK) = {rate:.3e} ({mism} levels, {rows} rows)")
    assert rate <= FLOOR

