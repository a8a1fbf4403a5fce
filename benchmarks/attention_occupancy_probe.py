"""Single-pass attention backward (dec self shape, 576 x 576 masked; dec cross 576 x 256) as a function of the number
of workgroups in flight: time per launch for B = 4 .. 43 clouds (6 heads each; one 8-wave workgroup per head and
256-key block, one workgroup per CU).  Answers whether a launch that fills 1.5 rounds of CU slots loses the idle half
round, or whether per-workgroup speed drops when all 256 CUs run the kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    nat.load()
    os.environ["PTOPS_ATTN_QSPLIT"] = "1"
    H, d = 6, 64
    for name, Nq, Nk, split, limit, blocks_per_head in (("dec self (mask)", 576, 576, 512, 512, 2), ("dec cross", 576, 256, 0, 256, 1)):
        for B in (4, 8, 16, 21, 24, 32, 42, 43, 64):
            q = torch.randn(B, H, Nq, d, device="cuda")
            k = torch.randn(B, H, Nk, d, device="cuda")
            v = torch.randn(B, H, Nk, d, device="cuda")
            out, lse = nat.attention_forward(q, k, v, split, limit, d ** -0.5)
            go = torch.randn_like(out)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            t = timeit(lambda: nat.attention_backward(q, k, v, out, go, lse, split, limit, d ** -0.5, dq, dk, dv, fused=True))
            wg = B * H * blocks_per_head
            print(f"{name:16s} B={B:3d}  full workgroups {wg:4d} ({wg / 256:5.2f} rounds of 256 CUs)  {t:7.1f} us  "
                  f"{t / wg * 256:7.1f} us per round-equivalent", flush=True)


if __name__ == "__main__":
    main()
