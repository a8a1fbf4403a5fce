"""The FPS and kNN launches of one bs = 32 train step (shapes of SURVEY 8a rows a4 / a6), a few times each: the target
of the rocprofv3 --pmc passes behind profiles/round3/primitives_pmc.json (north_star: "rocprof HBM GB/s for FPS / kNN").

    python3 benchmarks/primitives_only.py fps|knn [iterations]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fps"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = 32
torch.manual_seed(0)
nat.load()
if which == "fps":
    # one step: the grouper's two samplings, 2048 -> 512 (its prefixes also serve the 256 query seeds and the 64 denoise
    # seeds) and 512 -> 256 on the down-sampled cloud
    x = torch.rand(B, 2048, 3, device="cuda") - 0.5
    y = torch.rand(B, 512, 3, device="cuda") - 0.5
    for _ in range(it):
        nat.furthest_point_sample(x, 512)
        nat.furthest_point_sample(y, 256)
else:
    # (S x N, k) of one step: grouper (2048x2048, 512x2048, 512x512, 256x512; k 16), encoder (256x256, 8), decoder cross
    # (576x256, 8), decoder block-0 self (512x512 + 64x576, 8), loss (64x16384, 32)
    shapes = [(2048, 2048, 16), (512, 2048, 16), (512, 512, 16), (256, 512, 16), (256, 256, 8), (576, 256, 8),
              (512, 512, 8), (64, 576, 8), (64, 16384, 32)]
    clouds = {n: torch.rand(B, n, 3, device="cuda") - 0.5 for n in {s[1] for s in shapes} | {s[0] for s in shapes}}
    for _ in range(it):
        for S, N, k in shapes:
            nat.knn(k, clouds[N], clouds[S] if S != N else clouds[N])
torch.cuda.synchronize()
print("done")
