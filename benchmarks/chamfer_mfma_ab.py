"""A/B of the fragment prefetch in chamfer_mfma_kernel (csrc/chamfer_mfma.hip) in ONE process, interleaved rounds: the
bench's Chamfer launch (B = 32, 16384 x 16384, both directions) on uniform clouds and on a collapsed prediction, outputs
compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timed(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    nat.load()
    torch.manual_seed(0)
    B, n = 32, 16384
    cases = {"uniform": (torch.rand(B, n, 3, device="cuda") - 0.5, torch.rand(B, n, 3, device="cuda") - 0.5)}
    y = torch.rand(B, n, 3, device="cuda") - 0.5
    centres = y[:, torch.randint(0, n, (64,))]
    cases["collapsed prediction"] = ((centres[:, torch.randint(0, 64, (n,))] + 1e-3 * torch.randn(B, n, 3, device="cuda")).contiguous(), y)
    for name, (a, b) in cases.items():
        res = {}
        for mode in (0, 1):
            nat.set_option("PTOPS_CHAMFER_MFMA_PREFETCH", mode)
            res[mode] = nat.chamfer_forward(a, b, algorithm=nat.CHAMFER_MFMA)
        assert all(torch.equal(u, v) for u, v in zip(res[0], res[1]))
        t = {0: [], 1: []}
        for _ in range(7):
            for mode in (1, 0):
                nat.set_option("PTOPS_CHAMFER_MFMA_PREFETCH", mode)
                t[mode].append(timed(lambda: nat.chamfer_forward(a, b, algorithm=nat.CHAMFER_MFMA)))
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        print(f"{name:22s} with prefetch {med[1]:8.1f} us   without prefetch {med[0]:8.1f} us   (prep + search, median of 7 x 5)", flush=True)
    nat.set_option("PTOPS_CHAMFER_MFMA_PREFETCH", 0)


if __name__ == "__main__":
    main()
