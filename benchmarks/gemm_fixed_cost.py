"""What a split-bf16 GEMM launch costs besides its k-loop: time against K at fixed M, N (slope = one k-step, intercept =
launch + prologue + epilogue + tail), and the same with the epilogue's stores switched off.

    python benchmarks/gemm_fixed_cost.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native  # noqa: E402


def timed(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    _native.load()
    dev = torch.device("cuda", 0)
    M = 18432
    for N in (384, 1152):
        rows = []
        for K in (16, 64, 192, 384, 768, 1536):
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            t = {}
            for ns in (0, 1):
                _native.set_option("PTOPS_GEMM_DEBUG_NO_STORE", ns)
                t[ns] = timed(lambda: _native.gemm_split3(x, w, None))
            _native.set_option("PTOPS_GEMM_DEBUG_NO_STORE", 0)
            t["lib"] = timed(lambda: x @ w.t())
            rows.append((K, t[0], t[1], t["lib"]))
            print(f"M {M} N {N:5d} K {K:5d}: split3 {t[0]:7.1f} us   without stores {t[1]:7.1f} us   library {t['lib']:7.1f} us", flush=True)
        (k0, a0, b0, _), (k1, a1, b1, _) = rows[3], rows[5]
        slope = (a1 - a0) / ((k1 - k0) / 16)
        print(f"   per 16-deep k-step {slope:.3f} us; intercept {a0 - slope * k0 / 16:.1f} us with stores, "
              f"{b0 - (b1 - b0) / ((k1 - k0) / 16) * k0 / 16:.1f} us without")


if __name__ == "__main__":
    main()
