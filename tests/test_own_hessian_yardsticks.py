"""CPU: the yardstick machinery of ``tests/own_hessian_cases.py`` at the small size (the GPU test that uses it at
4096 x 4096 and on a K = 14336 slice is ``tests/test_gpu_own_hessian_parity.py``)."""
import numpy as np

from . import own_hessian_cases as oc


def test_small_case_yardsticks_and_forced_order(oracle):
    Wf, wb, xb = oc.make_inputs(oracle, oc.CASES["128x512"])
    side = oc.Side(oracle, Wf, xb)
    # the two fp32 Hessians are the same matrix to fp32 summation accuracy, and not the same bits
    assert oc.rel_diff(side.H_own, side.H_g64) <= 1e-5
    assert not np.array_equal(side.H_own, side.H_g64)
    for actorder in (None, "static"):
        y = side.yardsticks(actorder=actorder)
        assert y["scales_equal"]                         # scales come from W alone
        assert 0.0 <= y["h_order"] <= 5e-3 and 0.0 <= y["factor"] <= 5e-3
        # rate of a result against itself through the packed words
        rate, mism, rows = oc.nibble_rate(oracle, oracle.pack_int4(y["o"]["q"]), y["o"]["q"])
        assert (rate, mism, rows) == (0.0, 0, 0)
    # a forced sweep order is honoured, and gives the un-forced result when it is the argsort itself
    o = side.run(actorder="static")
    o_same = side.run(actorder="static", perm=o["perm"])
    np.testing.assert_array_equal(o_same["q"], o["q"])
    rev = np.asarray(o["perm"])[::-1].copy()
    o_rev = side.run(actorder="static", perm=rev)
    np.testing.assert_array_equal(o_rev["perm"], rev)
    np.testing.assert_array_equal(o_rev["scale"], o["scale"])
    assert (o_rev["q"] != o["q"]).any()
    y2 = side.yardsticks(actorder="static", perm=rev, with_factor=False)
    assert y2["perm_flips"] == 0 and y2["factor"] is None
