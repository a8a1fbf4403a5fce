"""Sweeps the Chamfer kernel's tuning knobs (env PTOPS_TUNE_*) on the hot-path shapes, interleaved
rounds in one process (cdna guide §5.4 rule 24).

Needs a tuning build: make -C imgadapointr_amd/csrc clean && make -C imgadapointr_amd/csrc TUNING=1
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def run(x, y, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        nat.chamfer_forward(x, y)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = 32
    torch.manual_seed(0)
    shapes = [(16384, 16384), (512, 16384), (2048, 2048)]
    variants = [dict(Q=q, PK=pk, SPLITS=sp, TARGET=t) for q, pk in ((2, 0), (4, 0), (8, 0), (2, 1), (4, 1), (8, 1))
                for sp, t in ((0, 8), (1, 8), (0, 16), (0, 4))]
    for n, m in shapes:
        x = torch.rand(B, n, 3, device="cuda") - 0.5
        y = torch.rand(B, m, 3, device="cuda") - 0.5
        ref = None
        res = {}
        for rnd in range(3):
            for v in variants:
                for k, val in v.items():
                    os.environ[f"PTOPS_TUNE_{k}"] = str(val)
                if rnd == 0:
                    out = nat.chamfer_forward(x, y)
                    if ref is None:
                        ref = out
                    else:
                        assert all(torch.equal(a, b) for a, b in zip(out, ref)), v
                res.setdefault(tuple(v.items()), []).append(run(x, y))
        print(f"shape {B}x{n}x{m}")
        for k, ts in sorted(res.items(), key=lambda kv: min(kv[1])):
            print(f"  {dict(k)}  min {min(ts):.4f} ms  med {sorted(ts)[1]:.4f} ms  "
                  f"{2.0 * B * n * m / min(ts) / 1e9:.2f} Tpairs/s")


if __name__ == "__main__":
    main()
