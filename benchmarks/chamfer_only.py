"""Runs only the dominant Chamfer launch (B=32, 16384 x 16384) a few times — the target of the
rocprofv3 --pmc passes whose summaries live under profiles/ (HBM traffic, VALU activity)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

torch.manual_seed(0)
x = torch.rand(32, 16384, 3, device="cuda") - 0.5
y = torch.rand(32, 16384, 3, device="cuda") - 0.5
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    nat.chamfer_forward(x, y)
torch.cuda.synchronize()
print("done")
