"""fp32 GEMM on the bf16 matrix pipe (csrc/gemm_split3.hip) against torch's fp32 GEMM (hipBLASLt / rocBLAS with the shipped
per-shape selections): error of both against float64, and time, on the dense-layer shapes of the bs = 32 train step.

    python benchmarks/gemm_split3_probe.py [rows,rows,...] > gpurun_out/gemm_split3_probe.log
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


def main():
    _native.load()
    print(enable_tuned_gemm(), flush=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    if len(sys.argv) > 1:          # row counts given: the layer widths of the step at those rows (e.g. 9216,4096 = bs 16)
        shapes = [(int(m), n, k) for m in sys.argv[1].split(",")
                  for n, k in ((384, 384), (1152, 384), (768, 384), (384, 768), (1024, 384), (1024, 1024), (384, 1024), (512, 384))]
    else:
        shapes = None
    shapes = shapes or [(18432, 384, 384), (18432, 1152, 384), (18432, 768, 384), (18432, 384, 768), (18432, 1024, 384),
              (18432, 1024, 1024), (18432, 384, 1024), (8192, 384, 384), (8192, 1152, 384), (8192, 768, 384),
              (8192, 384, 768), (16384, 512, 512), (65536, 512, 512), (18432, 512, 768)]
    print(f"{'M':>6} {'N':>5} {'K':>5} | fwd x.W^T: torch us  split3 us  speedup | dgrad dy.W: torch  split3  speedup | "
          f"wgrad dy^T.x: torch  split3 (+db) | rel err vs f64: torch  split3")
    tot_t = tot_s = 0.0
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        dy = torch.randn(M, N, device=dev)
        bias = torch.randn(N, device=dev)
        # forward: y = x w^T + b
        y_t = torch.nn.functional.linear(x, w, bias)
        y_s = _native.gemm_split3(x, w, bias)
        ref = (x[:2048].double() @ w.double().t() + bias.double())
        scale = float(ref.abs().max())
        e_t = float((y_t[:2048].double() - ref).abs().max()) / scale
        e_s = float((y_s[:2048].double() - ref).abs().max()) / scale
        t_f = timed(lambda: torch.nn.functional.linear(x, w, bias))
        s_f = timed(lambda: _native.gemm_split3(x, w, bias))
        # input gradient: dx = dy w     (A = dy [M,N] k-contiguous; B logical [K_in, N]: stored as w [N, K_in] = k strided)
        dx_t = dy @ w
        dx_s = _native.gemm_split3(dy, w, None, True, False)
        refd = dy[:2048].double() @ w.double()
        ed_t = float((dx_t[:2048].double() - refd).abs().max()) / float(refd.abs().max())
        ed_s = float((dx_s[:2048].double() - refd).abs().max()) / float(refd.abs().max())
        t_d = timed(lambda: dy @ w)
        s_d = timed(lambda: _native.gemm_split3(dy, w, None, True, False))
        # weight gradient: dw = dy^T x  (A logical [N, M]: stored dy [M, N] = k strided; B logical [K_in, M]: stored x [M, K_in])
        dw_t = dy.t() @ x
        dw_s, db_s = _native.gemm_split3_wgrad(dy, x, True)
        refw = dy.double().t() @ x.double()
        ew_t = float((dw_t.double() - refw).abs().max()) / float(refw.abs().max())
        ew_s = float((dw_s.double() - refw).abs().max()) / float(refw.abs().max())
        t_w = timed(lambda: dy.t() @ x)
        s_w = timed(lambda: _native.gemm_split3_wgrad(dy, x, True))
        e_b = float((db_s.double() - dy.double().sum(0)).abs().max()) / float(dy.double().sum(0).abs().max())
        tot_t += t_f + t_d
        tot_s += s_f + s_d
        tf = 2.0 * M * N * K / 1e6
        print(f"{M:6d} {N:5d} {K:5d} | {t_f:8.1f} ({tf / t_f:5.0f} TF) {s_f:8.1f} ({tf / s_f:5.0f} TF) {t_f / s_f:5.2f}x | "
              f"{t_d:8.1f} {s_d:8.1f} {t_d / s_d:5.2f}x | {t_w:8.1f} {s_w:8.1f} | "
              f"fwd {e_t:.1e} {e_s:.1e}  dgrad {ed_t:.1e} {ed_s:.1e}  wgrad {ew_t:.1e} {ew_s:.1e}  db {e_b:.1e}", flush=True)
    print(f"sum fwd + dgrad: torch {tot_t:.0f} us, split3 {tot_s:.0f} us")


if __name__ == "__main__":
    main()
