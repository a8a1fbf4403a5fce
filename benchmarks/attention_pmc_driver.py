"""The attention launches of one train step (B = 32, 6 heads): forward and backward (the dq + dk/dv pair by default,
`fused` as first argument for the single-pass f32 kernel) at the three training shapes, a few repetitions each - the
program run under `rocprofv3 --pmc ...` for profiles/roundN/attention_*pmc*.json (see benchmarks/pmc_attention.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

nat.load()
FUSED = len(sys.argv) > 1 and sys.argv[1] == "fused"
B, H, d = 32, 6, 64
torch.manual_seed(0)
for Nq, Nk, split, limit in ((576, 576, 512, 512), (576, 256, 0, 256), (256, 256, 0, 256)):
    q = torch.randn(B, H, Nq, d, device="cuda")
    k = torch.randn(B, H, Nk, d, device="cuda")
    v = torch.randn(B, H, Nk, d, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    go = torch.randn(B, Nq, H, d, device="cuda")
    for _ in range(6):
        out, lse = nat.attention_forward(q, k, v, split, limit, d ** -0.5)
        nat.attention_backward(q, k, v, out, go, lse, split, limit, d ** -0.5, dq, dk, dv, fused=FUSED)
    torch.cuda.synchronize()
