#!/usr/bin/env python3
"""tools/sweep_only.py on a lab build of the library (QT_LAB_LIB=n -> quantool_amd/lib/lab/libquantool_hip_lab<n>.so; 0 = the
shipped one).  Timing only: lab builds leave parts of sweep_quad_kernel out.   usage: sweep_lab.py K R"""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantool_amd.hip import _lib

n = int(os.environ.get("QT_LAB_LIB", "0"))
if n:
    _lib.LIB_PATH = ROOT / "quantool_amd" / "lib" / "lab" / f"libquantool_hip_lab{n}.so"
sys.argv = [str(ROOT / "tools" / "sweep_only.py"), sys.argv[1], sys.argv[2], "2"]
runpy.run_path(sys.argv[0], run_name="__main__")
