"""Pre-split operand GEMM (gemm_planes_experiment.hip in this directory) against gemm_split3 and torch: error vs float64
and time, forward shapes of the bs = 32 step.  The pack of the activation is timed separately.

The experiment is NOT part of libptops (result: no faster than gemm_split3, see profiles/round2/gemm_split3_notes.txt).
To re-run it: copy the .hip next to csrc/gemm_split3.hip, add it to SRCS in csrc/Makefile, and add ctypes wrappers
`split3_pack(x, transposed=False)` / `gemm_planes(a_packed, b_packed, M, N, K, bias=None)` for ptops_split3_pack_bytes,
ptops_split3_pack and ptops_gemm_f32_planes to imgadapointr_amd/_native.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    enable_tuned_gemm()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    shapes = [(18432, 384, 384), (18432, 1152, 384), (18432, 768, 384), (18432, 384, 768), (18432, 1024, 1024),
              (8192, 384, 384), (8192, 1152, 384), (8192, 384, 768), (65536, 512, 512), (300, 200, 64), (257, 129, 32)]
    print(f"{'M':>6} {'N':>5} {'K':>5} | torch us | split3 us | planes us (TF)  pack x us  pack w us | rel err vs f64: torch split3 planes")
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev) * torch.logspace(-2, 2, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        if os.environ.get("PROBE_ZERO"):          # power / clock experiment: all-zero operands draw the least power
            x.zero_(), w.zero_(), bias.zero_()
        xp, wp = _native.split3_pack(x), _native.split3_pack(w)
        y_p = _native.gemm_planes(xp, wp, M, N, K, bias)
        y_t = torch.nn.functional.linear(x, w, bias)
        y_s = _native.gemm_split3(x, w, bias) if K % 16 == 0 else y_t
        rows = slice(0, min(M, 2048))
        ref = x[rows].double() @ w.double().t() + bias.double()
        sc = float(ref.abs().max()) or 1.0
        errs = [float((y[rows].double() - ref).abs().max()) / sc for y in (y_t, y_s, y_p)]
        # transposed pack (the operand is w^T stored [K, N]): same bytes as packing the transpose explicitly
        wt = w.t().contiguous()
        assert torch.equal(_native.split3_pack(wt, transposed=True), wp)
        t_t = timed(lambda: torch.nn.functional.linear(x, w, bias))
        t_s = timed(lambda: _native.gemm_split3(x, w, bias))
        t_p = timed(lambda: _native.gemm_planes(xp, wp, M, N, K, bias))
        t_px = timed(lambda: _native.split3_pack(x))
        t_pw = timed(lambda: _native.split3_pack(w))
        print(f"{M:6d} {N:5d} {K:5d} | {t_t:8.1f} | {t_s:8.1f} | {t_p:8.1f} ({2.0 * M * N * K / 1e6 / t_p:5.0f})  {t_px:8.1f}  {t_pw:8.1f} | "
              f"{errs[0]:.1e} {errs[1]:.1e} {errs[2]:.1e}", flush=True)


if __name__ == "__main__":
    main()
