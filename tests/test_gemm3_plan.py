"""Item tables of the bf16x3 block-row products (csrc/gemm3_tn.hip: g3_plan_row) for every step of the
Cholesky chain at the K values of BASELINE.json's configs and a few awkward ones -- host logic only, through
the library's self-check entry point (no GPU call)."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def lib():
    from quantool_amd.hip import _lib

    return _lib.load()


@pytest.mark.parametrize("K", [768, 1000, 4096, 5120, 8192, 11008, 13824, 14336, 28672])
@pytest.mark.parametrize("nb", [256, 512])
def test_every_chunk_of_every_tile_is_covered_once(lib, K, nb):
    n_items, n_slabs, longest = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    checked = 0
    for J0 in range(nb, K, nb):
        J1 = min(K, J0 + nb)
        Tm, c_end = (J1 - J0 + 255) // 256, J0 // 64
        for tri, Tn in ((0, (K - J0 + 255) // 256), (1, (J0 + 255) // 256)):   # factor step / inverse block row
            rc = lib.qt_gemm3_plan_check(Tm, Tn, c_end, tri, ctypes.byref(n_items), ctypes.byref(n_slabs),
                                         ctypes.byref(longest))
            assert rc == 0, (K, nb, J0, tri, rc)
            total = sum(Tm * max(0, c_end - (4 * tj if tri else 0)) for tj in range(Tn))
            tiles = sum(Tm for tj in range(Tn) if c_end > (4 * tj if tri else 0))
            if tiles <= 256:
                assert n_items.value <= 256
                # near-equal pieces: the longest item is within one chunk of what 256 equal items would take,
                # or the two-chunk floor
                assert longest.value <= max(2, -(-total // max(1, 256 - tiles)) + 1), (K, nb, J0, tri)
            checked += 1
    assert checked > 0


def test_degenerate_shapes(lib):
    z = ctypes.c_int()
    assert lib.qt_gemm3_plan_check(1, 1, 1, 0, ctypes.byref(z), ctypes.byref(z), ctypes.byref(z)) == 0
    assert lib.qt_gemm3_plan_check(1, 300, 8, 0, ctypes.byref(z), ctypes.byref(z), ctypes.byref(z)) == 0   # more tiles than CUs
    assert z.value == 8
    assert lib.qt_gemm3_plan_check(0, 1, 1, 0, None, None, None) < 0
