#!/usr/bin/env python3
"""Run one of the tools on an alternative build of the library: QT_LAB_SO=<path to a .so> lab_run.py <tool.py> [args].
(The product always loads quantool_amd/lib/libquantool_hip.so; lab builds exist for A/B timing only.)"""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantool_amd.hip import _lib

so = os.environ.get("QT_LAB_SO")
if so:
    _lib.LIB_PATH = Path(so) if os.path.isabs(so) else ROOT / so
sys.argv = [str(ROOT / sys.argv[1])] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
