"""Event-timed forward and backward pair of one attention shape (for quick A/B runs under env switches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402
from imgadapointr_amd.models import Transformer_utils as TU  # noqa: E402

SHAPES = {"self": (576, 576, 512, 512), "cross": (576, 256, 0, 256), "enc": (256, 256, 0, 256)}


def timeit(fn, iters=30):
    for _ in range(3):
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
    B, H, d = int(os.environ.get("ATTN_B", "32")), 6, 64
    torch.manual_seed(0)
    contiguous = os.environ.get("ATTN_LAYOUT", "packed") == "contiguous"    # packed: views of [B,N,3,H,d] as in the model
    for which in sys.argv[1:] or list(SHAPES):
        Nq, Nk, split, limit = SHAPES[which]
        qkv = torch.randn(B, Nq, 3, H, d, device="cuda")
        q = qkv.permute(2, 0, 3, 1, 4)[0].detach()
        kv = torch.randn(B, Nk, 2, H, d, device="cuda")
        k = kv.permute(2, 0, 3, 1, 4)[0].detach()
        v = kv.permute(2, 0, 3, 1, 4)[1].detach()
        if contiguous:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        q, k, v = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
        go = torch.randn(B, Nq, H, d, device="cuda")
        t_f = timeit(lambda: nat.attention_forward(q.detach(), k.detach(), v.detach(), split, limit, d ** -0.5))
        res = [f"{which:6s} fwd {t_f:7.1f} us"]
        for mode in ("split", "fused"):
            TU.USE_MFMA_ATTENTION_BWD = mode
            o = TU._MFMAAttention.apply(q, k, v, split, limit, d ** -0.5)
            res.append(f"bwd[{mode}] {timeit(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True)):7.1f} us")
        print("   ".join(res), flush=True)


if __name__ == "__main__":
    main()
