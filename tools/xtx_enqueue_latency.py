"""Does qt_xtx_accumulate block the host?  Enqueue n calls on one stream and compare the host time
of the enqueue loop with the device time of the work (a blocking call makes them equal)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantool_amd.hip import ops  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
X = torch.randn(196608, K, device=dev).to(torch.bfloat16)
G = torch.zeros(K, K, device=dev)
ops.xtx_accumulate(X, G)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ops.xtx_accumulate(X, G)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"K={K} n={n}: host enqueue {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms")
