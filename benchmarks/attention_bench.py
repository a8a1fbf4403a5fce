"""fp32 attention: libptops MFMA kernel vs F.scaled_dot_product_attention on the path's shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    torch.manual_seed(0)
    B, H, d = 32, 6, 64
    for name, Nq, Nk, split, limit in (("dec self (mask)", 576, 576, 512, 512), ("dec cross", 576, 256, 0, 256),
                                       ("enc self", 256, 256, 0, 256), ("dec self eval", 512, 512, 0, 512)):
        qkv = torch.randn(B, Nq, 3, H, d, device="cuda")
        q = qkv.permute(2, 0, 3, 1, 4)[0]
        kv = torch.randn(B, Nk, 2, H, d, device="cuda")
        k, v = kv.permute(2, 0, 3, 1, 4)[0], kv.permute(2, 0, 3, 1, 4)[1]
        scale = d ** -0.5
        keep = None
        if split:
            blocked = torch.zeros(Nq, Nk, dtype=torch.bool, device="cuda")
            blocked[:split, limit:] = True
            keep = ~blocked
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=scale)          # [B,H,Nq,d]
        out, lse = nat.attention_forward(q, k, v, split, limit, scale)                        # [B,Nq,H,d]
        err = (out.permute(0, 2, 1, 3) - ref).abs().max().item()
        # fp64 reference for both
        a = (q.double() @ k.double().transpose(-1, -2)) * scale
        if keep is not None:
            a = a.masked_fill(~keep, float("-inf"))
        gold = (a.softmax(-1) @ v.double())
        e_ours = (out.permute(0, 2, 1, 3).double() - gold).abs().max().item()
        e_sdpa = (ref.double() - gold).abs().max().item()
        lse_gold = torch.logsumexp(a, -1) / 0.6931471805599453
        e_lse = (lse.double() - lse_gold).abs().max().item()
        t_ref = timeit(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=scale))
        t_ours = timeit(lambda: nat.attention_forward(q, k, v, split, limit, scale))
        tw = {}
        for w in "1234":                          # forward waves per workgroup, forced
            os.environ["PTOPS_ATTN_FWD_WAVES"] = w
            o_w, _ = nat.attention_forward(q, k, v, split, limit, scale)
            assert torch.equal(o_w, out)
            tw[w] = timeit(lambda: nat.attention_forward(q, k, v, split, limit, scale)) * 1e3
        os.environ.pop("PTOPS_ATTN_FWD_WAVES")
        print(f"{name:18s} forward by waves per workgroup (1..4): " + "  ".join(f"{tw[w]:.1f}" for w in "1234") + f" us   auto {t_ours*1e3:.1f} us")
        # backward: library kernels vs aten::_efficient_attention_backward (what the step used before)
        from imgadapointr_amd.models import Transformer_utils as TU
        go = torch.randn(B, Nq, H, d, device="cuda")
        qr, kr, vr = (t.detach().requires_grad_(True) for t in (q, k, v))
        default_bwd = TU.USE_MFMA_ATTENTION_BWD
        tb = {}
        for nb in ("fused", "split", False):
            TU.USE_MFMA_ATTENTION_BWD = nb
            o = TU._MFMAAttention.apply(qr, kr, vr, split, limit, scale)
            tb[nb] = timeit(lambda: torch.autograd.grad(o, (qr, kr, vr), go, retain_graph=True))
        TU.USE_MFMA_ATTENTION_BWD = "fused"
        qsplit = {}
        for B2 in (32, 16):                       # query-range split of the single-pass kernel: forced off / on
            q2, k2, v2 = (t[:B2].detach().requires_grad_(True) for t in (q, k, v))
            for n in ("1", "2"):
                os.environ["PTOPS_ATTN_QSPLIT"] = n
                o2 = TU._MFMAAttention.apply(q2, k2, v2, split, limit, scale)
                qsplit[(B2, n)] = timeit(lambda: torch.autograd.grad(o2, (q2, k2, v2), go[:B2], retain_graph=True)) * 1e3
        os.environ.pop("PTOPS_ATTN_QSPLIT")
        print(f"{name:18s} single-pass backward, query split off/on: B=32 {qsplit[(32, '1')]:.1f} / {qsplit[(32, '2')]:.1f} us   "
              f"B=16 {qsplit[(16, '1')]:.1f} / {qsplit[(16, '2')]:.1f} us")
        keys_eff = (split * limit + (Nq - split) * Nk) if split else Nq * Nk
        print(f"{name:18s} backward: single-pass {tb['fused']*1e3:7.1f} us ({10.0*keys_eff*d*B*H/tb['fused']/1e9:6.1f} TF/s)  "
              f"dq + dk/dv pair {tb['split']*1e3:7.1f} us ({14.0*keys_eff*d*B*H/tb['split']/1e9:6.1f} TF/s executed)  aten {tb[False]*1e3:7.1f} us")
        TU.USE_MFMA_ATTENTION_BWD = default_bwd
        fl = 4.0 * keys_eff * d * B * H
        print(f"{name:18s} Nq={Nq} Nk={Nk}: ours {t_ours*1e3:7.1f} us ({fl/t_ours/1e9:6.1f} TF/s)  sdpa {t_ref*1e3:7.1f} us "
              f"({4.0*Nq*Nk*d*B*H/t_ref/1e9:6.1f} TF/s)  |ours-sdpa| {err:.2e}  err64 ours {e_ours:.2e} sdpa {e_sdpa:.2e} lse {e_lse:.2e}")


if __name__ == "__main__":
    main()
