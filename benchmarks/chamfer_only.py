"""Runs only the dominant Chamfer launch (B=32, 16384 x 16384) a few times — the target of the
rocprofv3 passes whose summaries live under profiles/ (kernel times, HBM traffic, VALU activity).
Optional argv: iterations, n, m, kind(uniform|bench)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
m = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
torch.manual_seed(0)
x = torch.rand(32, n, 3, device="cuda") - 0.5
y = torch.rand(32, m, 3, device="cuda") - 0.5
for _ in range(it):
    nat.chamfer_forward(x, y)
torch.cuda.synchronize()
print("done")
