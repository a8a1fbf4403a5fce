"""One attention shape, forward + the dq / dk,dv backward pair, for rocprofv3 --kernel-trace --stats.

usage: attn_bwd_only.py [self|cross|enc] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd.models import Transformer_utils as TU  # noqa: E402

SHAPES = {"self": (576, 576, 512, 512), "cross": (576, 256, 0, 256), "enc": (256, 256, 0, 256)}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "self"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    Nq, Nk, split, limit = SHAPES[which]
    B, H, d = 32, 6, 64
    torch.manual_seed(0)
    qkv = torch.randn(B, Nq, 3, H, d, device="cuda")
    q = qkv.permute(2, 0, 3, 1, 4)[0].detach().requires_grad_(True)
    kv = torch.randn(B, Nk, 2, H, d, device="cuda")
    k = kv.permute(2, 0, 3, 1, 4)[0].detach().requires_grad_(True)
    v = kv.permute(2, 0, 3, 1, 4)[1].detach().requires_grad_(True)
    go = torch.randn(B, Nq, H, d, device="cuda")
    TU.USE_MFMA_ATTENTION_BWD = os.environ.get("ATTN_BWD", "split")
    o = TU._MFMAAttention.apply(q, k, v, split, limit, d ** -0.5)
    for _ in range(iters):
        torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
    torch.cuda.synchronize()
    print("done", which, iters)


if __name__ == "__main__":
    main()
