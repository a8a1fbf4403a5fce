"""Chamfer backward: the reference's atomicAdd form against the deterministic form (transposed neighbour map), B = 32 at the
three shapes of a train step, on uniform clouds and on a collapsed prediction (what a randomly initialised model emits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
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
    B = 32
    print(f"{'case':44s} {'atomics us':>11s} {'deterministic us':>17s}")
    for n, m in ((2048, 2048), (512, 16384), (16384, 16384)):
        y = torch.rand(B, m, 3, device="cuda") - 0.5
        for kind in ("uniform", "collapsed"):
            if kind == "uniform":
                x = torch.rand(B, n, 3, device="cuda") - 0.5
            else:
                c = y[:, torch.randint(0, m, (48,))]
                x = (c[:, torch.randint(0, 48, (n,))] * 0.05 + 0.02 * torch.randn(B, n, 3, device="cuda")).contiguous()
            d1, d2, i1, i2 = nat.chamfer_forward(x, y)
            g1, g2 = torch.randn_like(d1), torch.randn_like(d2)
            ta = timed(lambda: nat.chamfer_backward(x, y, i1, i2, g1, g2, deterministic=False))
            td = timed(lambda: nat.chamfer_backward(x, y, i1, i2, g1, g2, deterministic=True))
            rows = int(torch.bincount(i2[0].long(), minlength=n).max())
            print(f"{n:5d} x {m:5d} {kind:10s} (longest list {rows:5d})   {ta:11.1f} {td:17.1f}", flush=True)


if __name__ == "__main__":
    main()
