"""`__graft_entry__.build()` followed by `smoke()` in ONE process, the library loaded before anything touched the GPU
(round 3: loaded ahead of torch, the library bound the system HIP runtime instead of the one torch ships, and its
first pinned allocation -- the Gram kernel's tile table -- failed; `_lib.load()` now imports torch first)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_build_then_smoke_in_one_process():
    env = dict(os.environ)
    env.pop("QT_XTX_ORDER", None)
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "smoke ok" in out.stdout


def test_library_loaded_before_torch_touches_the_gpu():
    code = ("from quantool_amd.hip import _lib\n"
            "lib = _lib.load()\n"
            "import torch\n"
            "from quantool_amd.hip import ops\n"
            "X = torch.randn(512, 256, device='cuda:0').to(torch.bfloat16)\n"
            "G = torch.zeros(256, 256, device='cuda:0')\n"
            "ops.xtx_accumulate(X, G)\n"
            "torch.cuda.synchronize()\n"
            "ref = X.float().t() @ X.float()\n"
            "assert torch.allclose(torch.tril(G), torch.tril(ref), rtol=1e-4, atol=1e-3)\n"
            "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
