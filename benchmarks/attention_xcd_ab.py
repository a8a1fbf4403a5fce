"""A/B of the XCD-contiguous job order of the split-bf16 attention kernels (csrc/attention.hip: xcd_job) against the
plain blockIdx order, in ONE process with interleaved rounds (cdna guide rule 24): forward and the dq + dk/dv backward
pair at the train step's shapes (bs 32 and the bs 16 of the image / segmentation variants).  Results must be bit-equal."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timed(fn, iters=10):
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
    H, d = 6, 64
    print(f"{'shape':34s} {'fwd plain':>10s} {'fwd xcd':>10s} {'bwd plain':>10s} {'bwd xcd':>10s}   us (median of 7 rounds)")
    for B in (32, 16):
        for name, Nq, Nk, split, limit in (("dec self (mask) 576x576", 576, 576, 512, 512), ("dec cross 576x256", 576, 256, 0, 256),
                                           ("enc self 256x256", 256, 256, 0, 256)):
            qkv = torch.randn(B, Nq, 3, H, d, device="cuda")
            q = qkv.permute(2, 0, 3, 1, 4)[0]
            kv = torch.randn(B, Nk, 2, H, d, device="cuda")
            k, v = kv.permute(2, 0, 3, 1, 4)[0], kv.permute(2, 0, 3, 1, 4)[1]
            scale = d ** -0.5
            dout = torch.randn(B, Nq, H, d, device="cuda")
            res = {}
            for mode in (0, 1):
                nat.set_option("PTOPS_ATTN_NO_XCD_REMAP", mode)
                out, lse = nat.attention_forward(q, k, v, split, limit, scale)
                dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                nat.attention_backward(q, k, v, out, dout, lse, split, limit, scale, dq, dk, dv, fused=False)
                res[mode] = (out, lse, dq.clone(), dk.clone(), dv.clone())
            assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
            out, lse = res[0][0], res[0][1]
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            t = {(m, w): [] for m in (0, 1) for w in "fb"}
            for _ in range(7):
                for mode in (1, 0):
                    nat.set_option("PTOPS_ATTN_NO_XCD_REMAP", mode)
                    t[(mode, "f")].append(timed(lambda: nat.attention_forward(q, k, v, split, limit, scale)))
                    t[(mode, "b")].append(timed(lambda: nat.attention_backward(q, k, v, out, dout, lse, split, limit, scale,
                                                                               dq, dk, dv, fused=False)))
            med = {kk: sorted(vv)[len(vv) // 2] for kk, vv in t.items()}
            print(f"B={B:2d} {name:29s} {med[(1, 'f')]:10.1f} {med[(0, 'f')]:10.1f} {med[(1, 'b')]:10.1f} {med[(0, 'b')]:10.1f}", flush=True)
    nat.set_option("PTOPS_ATTN_NO_XCD_REMAP", 0)


if __name__ == "__main__":
    main()
