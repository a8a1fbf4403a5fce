"""FPS launch shapes (threads per cloud x points per lane) at the two training sizes, B = 32 (needs the tuning build of the
library: `make -C imgadapointr_amd/csrc clean && make -C imgadapointr_amd/csrc TUNING=1`; rebuild without TUNING afterwards)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


nat.load()
torch.manual_seed(0)
for N, npoint, codes in ((2048, 512, ("", "12816", "25608", "51204", "102402")), (512, 256, ("", "6408", "12804", "25602", "51201"))):
    x = torch.rand(32, N, 3, device="cuda") - 0.5
    ref = None
    for code in codes:
        if code:
            os.environ["PTOPS_TUNE_FPS"] = code
        else:
            os.environ.pop("PTOPS_TUNE_FPS", None)
        out = nat.furthest_point_sample(x, npoint)
        ref = out if ref is None else ref
        assert torch.equal(out, ref)
        t = timeit(lambda: nat.furthest_point_sample(x, npoint))
        print(f"N={N} -> {npoint}  config {code or 'default':>8s}  {t:7.1f} us  ({t / (npoint - 1):.3f} us per pick)", flush=True)
os.environ.pop("PTOPS_TUNE_FPS", None)
