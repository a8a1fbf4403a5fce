"""Host-side cost of one dense-layer GEMM call (enqueue only, the GPU runs behind): torch's F.linear / mm with the shipped
per-shape kernel selection (TunableOp) on and off, and libptops' split-bf16 GEMM through ctypes.

    python benchmarks/gemm_dispatch_cost.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def host_us(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


def main():
    _native.load()
    dev = torch.device("cuda", 0)
    x = torch.randn(9216, 384, device=dev)
    w = torch.randn(384, 384, device=dev) * 0.05
    b = torch.randn(384, device=dev)
    g = torch.randn(9216, 384, device=dev)
    small = torch.randn(64, 384, device=dev)       # a GEMM whose GPU time is negligible: the host never waits for the queue
    rows = []
    for label in ("TunableOp off", "TunableOp on (shipped selections)"):
        if label.startswith("TunableOp on"):
            print(enable_tuned_gemm())
        rows.append((label, host_us(lambda: torch.nn.functional.linear(small, w, b)), host_us(lambda: small @ w),
                     host_us(lambda: small.t() @ small)))
    for label, a, c, d in rows:
        print(f"{label:36s} F.linear {a:6.1f} us   mm {c:6.1f} us   mm (transposed operand) {d:6.1f} us   per call, host side")
    print(f"{'libptops gemm_split3 via ctypes':36s} {host_us(lambda: _native.gemm_split3(small, w, b)):6.1f} us per call")
    print(f"{'torch.empty (one output alloc)':36s} {host_us(lambda: torch.empty((64, 384), device=dev)):6.1f} us per call")
    print(f"{'an element-wise add':36s} {host_us(lambda: small + small):6.1f} us per call")


if __name__ == "__main__":
    main()
