"""The Gram kernel's tile table (csrc/xtx.hip: xtx_tile_order) -- host logic only, through the library's
self-check entry point (no GPU call): every lower-triangular tile exactly once for every order and awkward K,
and the locality figures DESIGN.md 4.1 quotes for the shipped order."""
import ctypes
import os
import subprocess
import sys

import pytest


def _check(K, order=None):
    """Runs in a child process: the table of a K is built once per process with the order of that moment."""
    code = (
        "import ctypes\n"
        "from quantool_amd.hip import _lib\n"
        "lib = _lib.load()\n"
        "n, c, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()\n"
        "for K in %r:\n"
        "    rc = lib.qt_xtx_tile_table_check(K, ctypes.byref(n), ctypes.byref(c), ctypes.byref(r))\n"
        "    print(K, rc, n.value, c.value, r.value)\n" % (list(K),))
    env = dict(os.environ)
    env.pop("QT_XTX_ORDER", None)
    if order is not None:
        env["QT_XTX_ORDER"] = str(order)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    return {int(a): (int(b), int(c), int(d), int(e)) for a, b, c, d, e in (l.split() for l in out.stdout.splitlines())}


KS = [8, 256, 264, 768, 1000, 2048, 2056, 4096, 5120, 8192, 11008, 13824, 14336, 28672]


@pytest.mark.parametrize("order", [None, 0, 1, 2])
def test_every_lower_tile_exactly_once(order):
    got = _check(KS, order)
    for K in KS:
        nt = (K + 255) // 256
        rc, n, chunk, rnd = got[K]
        assert rc == 0 and n == nt * (nt + 1) // 2, (K, order, got[K])
        assert nt <= chunk <= 2 * n and nt <= rnd <= chunk


def test_default_order_is_the_aligned_one_from_one_round_of_tiles_up_and_its_locality_figures():
    dflt, pairs, old = _check([4096, 14336, 28672]), _check([4096, 14336, 28672], 1), _check([4096, 14336, 28672], 0)
    aligned = _check([4096, 14336, 28672], 2)
    assert dflt[14336] == aligned[14336] and dflt[28672] == aligned[28672] and dflt[4096] == pairs[4096]
    # distinct panels per 32-entry chunk, summed over the table
    assert [aligned[K][2] for K in (4096, 14336, 28672)] == [48, 588, 2352]
    assert [pairs[K][2] for K in (4096, 14336, 28672)] == [53, 701, 2659]
    assert [old[K][2] for K in (4096, 14336, 28672)] == [59, 898, 3491]
    assert aligned[14336][3] == 232 and pairs[14336][3] == 255


def test_invalid_k_is_refused():
    from quantool_amd.hip import _lib

    assert _lib.load().qt_xtx_tile_table_check(0, None, None, None) != 0
