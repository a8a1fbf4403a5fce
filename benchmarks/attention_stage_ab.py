"""In-process A/B of a run-time switch of the attention kernels (default: PTOPS_ATTN_STAGE_EARLY = split + LDS writes of
the next tile in front of the barrier, as in rounds 2-3, vs right after it): forward and the dq + dk/dv backward pair at
the three training shapes (B = 32, 6 heads), interleaved rounds, best and median of the rounds.

    python benchmarks/attention_stage_ab.py [OPTION_NAME]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

nat.load()
OPT = sys.argv[1] if len(sys.argv) > 1 else "PTOPS_ATTN_STAGE_EARLY"
B, H, d = 32, 6, 64
torch.manual_seed(0)


def timed(fn, n=20):
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


for Nq, Nk, split, limit in ((576, 576, 512, 512), (576, 256, 0, 256), (256, 256, 0, 256)):
    q = torch.randn(B, H, Nq, d, device="cuda")
    k = torch.randn(B, H, Nk, d, device="cuda")
    v = torch.randn(B, H, Nk, d, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    go = torch.randn(B, Nq, H, d, device="cuda")
    out, lse = nat.attention_forward(q, k, v, split, limit, d ** -0.5)
    res = {0: {"fwd": [], "bwd": []}, 1: {"fwd": [], "bwd": []}}
    for rnd in range(5):
        for val in (0, 1):
            nat.set_option(OPT, val)
            res[val]["fwd"].append(timed(lambda: nat.attention_forward(q, k, v, split, limit, d ** -0.5)))
            res[val]["bwd"].append(timed(lambda: nat.attention_backward(q, k, v, out, go, lse, split, limit, d ** -0.5, dq, dk, dv,
                                                                        fused=False)))
    nat.set_option(OPT, 0)
    for kind in ("fwd", "bwd"):
        a, b = res[0][kind], res[1][kind]
        print(f"{Nq:4d} x {Nk:4d} {kind}: {OPT}=0 best {min(a):7.1f} median {statistics.median(a):7.1f} us | =1 best {min(b):7.1f} "
              f"median {statistics.median(b):7.1f} us", flush=True)
