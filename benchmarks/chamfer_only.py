"""Runs only the dominant Chamfer launch (B=32, 16384 x 16384) a few times — the target of the
rocprofv3 passes whose summaries live under profiles/ (kernel times, HBM traffic, VALU activity).
Optional argv: iterations, n, m, algorithm (auto|brute|sweep|mfma)."""
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
algo = {"auto": nat.CHAMFER_AUTO, "brute": nat.CHAMFER_BRUTE, "sweep": nat.CHAMFER_SWEEP,
        "mfma": nat.CHAMFER_MFMA}[sys.argv[4] if len(sys.argv) > 4 else "auto"]
for _ in range(it):
    nat.chamfer_forward(x, y, algorithm=algo)
torch.cuda.synchronize()
print("done")
