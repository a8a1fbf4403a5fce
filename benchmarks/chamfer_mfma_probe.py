"""Times the three exact Chamfer algorithms (brute force, pruned sweep, MFMA filter) on the geometries that matter and
checks that they agree bit for bit.

    python benchmarks/chamfer_mfma_probe.py > gpurun_out/chamfer_mfma_probe.log
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def clouds(kind, B, n, m, dev):
    g = torch.Generator(device=dev).manual_seed(1)
    gt = torch.rand(B, m, 3, device=dev, generator=g) - 0.5
    if kind == "uniform":
        a = torch.rand(B, n, 3, device=dev, generator=g) - 0.5
    elif kind == "near":
        a = gt[:, torch.randperm(m, device=dev)[:n]] + torch.randn(B, n, 3, device=dev, generator=g) * 0.01
    elif kind == "clusters":
        c = torch.rand(B, 8, 3, device=dev, generator=g) * 0.2 - 0.1
        a = c[:, torch.randint(0, 8, (n,), device=dev)] + torch.randn(B, n, 3, device=dev, generator=g) * 0.004
    elif kind == "lattice":
        a = torch.randint(0, 24, (B, n, 3), device=dev, generator=g).float() / 32 - 0.4
        gt = torch.randint(0, 24, (B, m, 3), device=dev, generator=g).float() / 32 - 0.4
    else:
        raise ValueError(kind)
    return a.contiguous(), gt.contiguous()


def main():
    dev = torch.device("cuda", 0)
    _native.load()
    names = {_native.CHAMFER_BRUTE: "brute", _native.CHAMFER_SWEEP: "sweep", _native.CHAMFER_MFMA: "mfma"}
    for B, n, m in ((32, 16384, 16384), (32, 512, 16384), (32, 2048, 2048), (8, 8192, 8192), (1, 16384, 16384)):
        for kind in ("uniform", "near", "clusters", "lattice"):
            a, gt = clouds(kind, B, n, m, dev)
            ref = _native.chamfer_forward(a, gt, algorithm=_native.CHAMFER_BRUTE)
            row = []
            algos = [_native.CHAMFER_BRUTE, _native.CHAMFER_MFMA]
            if _native.load().ptops_chamfer_sweep_available(B, n, m):
                algos.append(_native.CHAMFER_SWEEP)
            for algo in algos:
                out = _native.chamfer_forward(a, gt, algorithm=algo)
                same = all(torch.equal(x, y) for x, y in zip(out, ref))
                ms = timed(lambda: _native.chamfer_forward(a, gt, algorithm=algo))
                row.append(f"{names[algo]} {ms:7.3f} ms{'' if same else ' MISMATCH'}")
            # per direction for the MFMA filter
            for d, pair in enumerate(((_native.CHAMFER_MFMA, _native.CHAMFER_BRUTE), (_native.CHAMFER_BRUTE, _native.CHAMFER_MFMA))):
                ms = timed(lambda: _native.chamfer_forward(a, gt, algorithm=pair))
                row.append(f"mfma-dir{d}+brute {ms:7.3f}")
            stats = torch.zeros(4, dtype=torch.int64, device=dev)
            lib = _native.load()
            wsb = lib.ptops_chamfer_forward_workspace_bytes(B, n, m)
            ws = torch.empty((wsb // 8 + 2,), dtype=torch.int64, device=dev)
            o = [torch.empty(B, n, device=dev), torch.empty(B, m, device=dev),
                 torch.empty(B, n, dtype=torch.int32, device=dev), torch.empty(B, m, dtype=torch.int32, device=dev)]
            _native._call("ptops_chamfer_forward_ex", _native._ptr(a), _native._ptr(gt), B, n, m, *[_native._ptr(t) for t in o],
                          _native._ptr(ws), 3, 3, _native._ptr(stats), _native._stream(a))
            st = stats.tolist()
            fr = [st[2 * d] / max(st[2 * d + 1], 1) for d in range(2)]
            print(f"B={B:2d} {n:5d}x{m:5d} {kind:9s} " + " | ".join(row) + f" | exact fraction {fr[0]:.5f} {fr[1]:.5f}", flush=True)


if __name__ == "__main__":
    main()
