"""Probe: libptops attention forward paired with PyTorch-ROCm's (AOTriton) memory-efficient attention
backward through aten::_efficient_attention_backward — gradient parity and timing vs plain SDPA."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

LN2 = 0.6931471805599453


def run(Nq, Nk, split, limit, B=32, H=6, d=64):
    torch.manual_seed(0)
    qkv = torch.randn(B, Nq, 3, H, d, device="cuda", requires_grad=True)
    kv = torch.randn(B, Nk, 2, H, d, device="cuda", requires_grad=True)
    scale = d ** -0.5
    go = torch.randn(B, H, Nq, d, device="cuda")
    keep = bias = None
    if split:
        blocked = torch.zeros(Nq, Nk, dtype=torch.bool, device="cuda")
        blocked[:split, limit:] = True
        keep = ~blocked
        bias = torch.zeros(Nq, Nk, device="cuda").masked_fill(blocked, float("-inf")).expand(B, H, Nq, Nk)

    def views():
        q = qkv.permute(2, 0, 3, 1, 4)[0]
        k, v = kv.permute(2, 0, 3, 1, 4)[0], kv.permute(2, 0, 3, 1, 4)[1]
        return q, k, v

    # reference: SDPA fwd + bwd
    q, k, v = views()
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=scale)
    g_ref = torch.autograd.grad(out, (qkv, kv), go)

    # ours: forward kernel + aten backward
    with torch.no_grad():
        q, k, v = views()
        o, lse2 = nat.attention_forward(q, k, v, split, limit, scale)          # o [B,Nq,H,d]
        lse = lse2 * LN2
        zero = torch.zeros((), dtype=torch.int64, device="cuda")
        gq, gk, gv, _ = torch.ops.aten._efficient_attention_backward(
            go.transpose(1, 2), q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), bias, o, None, None,
            Nq, Nk, lse, 0.0, zero, zero, 0, False, scale=scale)
    # gq [B,Nq,H,d] is the gradient of the q view; fold back into qkv / kv layouts
    g_qkv = torch.zeros_like(qkv)
    g_qkv[:, :, 0] = gq
    g_kv = torch.stack([gk, gv], dim=2)
    e1 = (g_qkv[:, :, 0] - g_ref[0][:, :, 0]).abs().max().item()
    e2 = (g_kv - g_ref[1]).abs().max().item()
    print(f"Nq={Nq} Nk={Nk} mask={bool(split)}: max|dq diff| {e1:.2e}  max|dkv diff| {e2:.2e}  (|grad| max {g_ref[1].abs().max().item():.2f})")


if __name__ == "__main__":
    run(576, 576, 512, 512)
    run(576, 256, 0, 256)
    run(256, 256, 0, 256)
