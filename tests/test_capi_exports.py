"""The C-ABI library loads and exports every symbol include/quantool_amd.h declares, the ctypes
table mirrors the header, and the product tree never reaches into oracle/."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "quantool_amd.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(qt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from quantool_amd.hip import _lib

    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as g

        g.build()
    return _lib.LIB_PATH


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("qt_xtx_accumulate", "qt_hessian_prepare", "qt_cholesky_inverse_upper", "qt_group_minmax_qparams",
                 "qt_gptq_sweep", "qt_pack_int4", "qt_last_error", "qt_cholesky_inverse_upper_batched",
                 "qt_gptq_sweep_grouped", "qt_weight_gather_qparams"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(lib_path)], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in _declared() if n not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_table_matches_header(lib_path):
    from quantool_amd.hip import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    lib = ctypes.CDLL(str(lib_path))          # loading needs no GPU
    for name in _declared():
        assert hasattr(lib, name)
    # host-only entry points are callable without a device
    lib.qt_version.restype = ctypes.c_int
    assert lib.qt_version() >= 100
    lib.qt_xtx_workspace_bytes.restype = ctypes.c_size_t
    lib.qt_xtx_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    assert lib.qt_xtx_workspace_bytes(196608, 4096) > 0
    assert lib.qt_xtx_workspace_bytes(0, 4096) == 0


def test_product_tree_never_imports_the_oracle():
    offenders = []
    for py in (ROOT / "quantool_amd").rglob("*.py"):
        src = py.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "reference_path" in src:
            offenders.append(str(py.relative_to(ROOT)))
    assert not offenders, f"product code must not use oracle/: {offenders}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from quantool_amd.hip import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
