"""Which of the two fp32 edge-conv paths (fused kernels / literal chain) is closer to a float64
evaluation of the same grouper?  Diagnostic for tests/test_kernels_gpu.py::test_fused_edge_conv_*."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from imgadapointr_amd.models import base_blocks as BB
from imgadapointr_amd.models.Transformer_utils import knn_idx
from golden_util import seeded_state_dict, synthetic_batch

B, N = 2, 2048
g = BB.DGCNN_Grouper(k=16)
sd = seeded_state_dict(g.state_dict(), "grouper-fused", std=0.3)
for key in sd:
    if key.endswith("1.weight"):
        sd[key] = sd[key] * torch.where(torch.arange(sd[key].numel()) % 3 == 0, -1.0, 1.0)
g.load_state_dict(sd)
g = g.cuda()
x = torch.from_numpy(synthetic_batch(B, N, 64, seed=N, structured=False)[0]).cuda()


def run(fused):
    BB.FUSED_EDGE_CONV = fused
    xin = x.clone().requires_grad_(True)
    coor, f = g(xin, [N // 4, N // 8])
    w = torch.linspace(-1, 1, f.numel(), device="cuda").view_as(f)
    grads = torch.autograd.grad((f * w).sum(), [xin] + list(g.parameters()))
    return f.detach(), [t.detach() for t in grads]


class G64(BB.DGCNN_Grouper):
    def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        idx = knn_idx(self.k, coor_k.float().transpose(-1, -2).contiguous(), coor_q.float().transpose(-1, -2).contiguous()).long()
        Bq, C, Nk = x_k.shape
        Nq = x_q.shape[2]
        nb = torch.gather(x_k.unsqueeze(2).expand(Bq, C, Nq, Nk), 3, idx.unsqueeze(1).expand(Bq, C, Nq, self.k))
        feat = torch.cat([nb - x_q.unsqueeze(-1), x_q.unsqueeze(-1).expand_as(nb)], dim=1)
        return layer(feat).max(dim=-1)[0]

    @staticmethod
    def fps_downsample(coor, x, num_group, fps_idx=None):
        from imgadapointr_amd import _native
        idx = _native.furthest_point_sample(coor.float().transpose(1, 2).contiguous(), num_group).long()
        comb = torch.cat([coor, x], dim=1)
        out = torch.gather(comb, 2, idx.unsqueeze(1).expand(-1, comb.size(1), -1))
        return out[:, :3], out[:, 3:]


g64 = G64(k=16)
g64.load_state_dict(sd)
g64 = g64.cuda().double()
xin = x.double().clone().requires_grad_(True)
coor, f64 = g64(xin, [N // 4, N // 8])
w = torch.linspace(-1, 1, f64.numel(), device="cuda").view_as(f64).double()
ref = [t.detach() for t in torch.autograd.grad((f64 * w).sum(), [xin] + list(g64.parameters()))]
fa, ga = run(True)
fb, gb = run(False)
print("forward  |fused-f64| %.3e   |literal-f64| %.3e   max|f64| %.3e" % ((fa - f64).abs().max(), (fb - f64).abs().max(), f64.abs().max()))
for (name, _), a, b, r in zip([("x", None)] + list(g.named_parameters()), ga, gb, ref):
    print("%-24s |fused-f64| %.3e   |literal-f64| %.3e   |fused-literal| %.3e   max|f64| %.3e" % (
        name, (a - r).abs().max(), (b - r).abs().max(), (a - b).abs().max(), r.abs().max()))

# per-stage check on the real P / R / idx of the chain
print("\nper-stage: _FusedEdgeConv on the chain's own inputs vs float64")
orig = BB._FusedEdgeConv.apply
def checked(P, R, idx, gamma, beta, G, eps, slope):
    out = orig(P, R, idx, gamma, beta, G, eps, slope)
    with torch.enable_grad():
        Bq, C, Nk = P.shape; Nq, k = idx.shape[1], idx.shape[2]
        gen = torch.Generator().manual_seed(1)
        wt = torch.linspace(-1, 1, Bq * C * Nq, device='cuda').view(Bq, Nq, C).transpose(1, 2)
        Pl, Rl, gl, bl = [t.detach().clone().requires_grad_(True) for t in (P, R, gamma, beta)]
        o2 = orig(Pl, Rl, idx, gl, bl, G, eps, slope)
        got = torch.autograd.grad((o2 * wt).sum(), [Pl, Rl, gl, bl])
        Pd, Rd, gd, bd = [t.detach().double().requires_grad_(True) for t in (P, R, gamma, beta)]
        nb = torch.gather(Pd.unsqueeze(2).expand(Bq, C, Nq, Nk), 3, idx.long().unsqueeze(1).expand(Bq, C, Nq, k))
        y = torch.nn.functional.group_norm(nb + Rd.unsqueeze(-1), G, gd, bd, eps)
        ref = torch.nn.functional.leaky_relu(y, slope).max(dim=-1)[0]
        want = torch.autograd.grad((ref * wt.double()).sum(), [Pd, Rd, gd, bd])
        print("stage C=%d Nq=%d Nk=%d |P| %.2e |R| %.2e |y| %.2e  fwd %.2e" % (C, Nq, Nk, P.abs().max(), R.abs().max(), (nb + Rd.unsqueeze(-1)).abs().max(), (o2.double() - ref).abs().max()),
              " ".join("%s %.2e/%.2e" % (n, (a.double() - b).abs().max(), b.abs().max()) for n, a, b in zip(("dP", "dR", "dg", "db"), got, want)))
    return out
BB._FusedEdgeConv.apply = checked
run(True)

# variant A: P/R split, but the stage itself as plain torch fp32 ops (autograd backward)
BB._FusedEdgeConv.apply = orig
class _TorchStage:
    @staticmethod
    def apply(P, R, idx, gamma, beta, G, eps, slope):
        Bq, C, Nk = P.shape; Nq, k = idx.shape[1], idx.shape[2]
        nb = torch.gather(P.unsqueeze(2).expand(Bq, C, Nq, Nk), 3, idx.long().unsqueeze(1).expand(Bq, C, Nq, k))
        y = torch.nn.functional.group_norm(nb + R.unsqueeze(-1), G, gamma, beta, eps)
        return torch.nn.functional.leaky_relu(y, slope).max(dim=-1)[0]
saved = BB._FusedEdgeConv
BB._FusedEdgeConv = _TorchStage
fc, gc = run(True)
BB._FusedEdgeConv = saved
print("\nvariant A (split GEMMs + torch stage) vs f64, and ours vs variant A")
for (name, _), a, c, r in zip([("x", None)] + list(g.named_parameters()), ga, gc, ref):
    print("%-24s |A-f64| %.3e   |ours-A| %.3e" % (name, (c - r).abs().max(), (a - c).abs().max()))

# how many selections / LeakyReLU signs differ between the two fp32 formulations?
print("\nselection flips per stage (literal conv vs split P+R), same inputs")
from imgadapointr_amd import _native
def probe(self, layer, coor_q, x_q, coor_k, x_k):
    conv, norm, act = layer[0], layer[1], layer[2]
    idx = knn_idx(self.k, coor_k.transpose(-1, -2).contiguous(), coor_q.transpose(-1, -2).contiguous())
    feat = self.get_graph_feature(coor_q, x_q, coor_k, x_k)
    yl = conv(feat)                                           # literal edge values [B,C',Nq,k]
    C = x_k.size(1)
    w = conv.weight.view(conv.out_channels, 2 * C)
    P = torch.matmul(w[:, :C], x_k); R = torch.matmul(w[:, C:] - w[:, :C], x_q)
    Bq, Co, Nk = P.shape; Nq = idx.shape[1]
    ys = torch.gather(P.unsqueeze(2).expand(Bq, Co, Nq, Nk), 3, idx.long().unsqueeze(1).expand(Bq, Co, Nq, self.k)) + R.unsqueeze(-1)
    zl = act(norm(yl)); zs = act(norm(ys))
    jl = zl.argmax(-1); js = zs.argmax(-1)
    top2 = zl.topk(2, dim=-1)[0]
    gap = (top2[..., 0] - top2[..., 1])
    print("stage C'=%d: |y_lit - y_split| max %.2e (|y| %.2e)  argmax differs %d / %d   top-2 gap < 1e-6: %d   exact ties: %d   sign flips %d" % (
        Co, (yl - ys).abs().max(), yl.abs().max(), (jl != js).sum(), jl.numel(), (gap < 1e-6).sum(), (gap == 0).sum(),
        ((zl.max(-1)[0] > 0) != (zs.max(-1)[0] > 0)).sum()))
    return zl.max(-1)[0]
BB.DGCNN_Grouper.edge_conv = probe
with torch.no_grad():
    g(x, [N // 4, N // 8])

# variant A64: the split formulation in float64; variant Ac: x_q made contiguous before the GEMM
class GA(G64):
    def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        idx = knn_idx(self.k, coor_k.float().transpose(-1, -2).contiguous(), coor_q.float().transpose(-1, -2).contiguous())
        conv, norm, act = layer[0], layer[1], layer[2]
        C = x_k.size(1)
        w = conv.weight.view(conv.out_channels, 2 * C)
        if self.contig:
            x_q = x_q.contiguous()
        P = torch.matmul(w[:, :C], x_k); R = torch.matmul(w[:, C:] - w[:, :C], x_q)
        return _TorchStage.apply(P, R, idx, norm.weight, norm.bias, norm.num_groups, norm.eps, act.negative_slope)
for dt, contig in ((torch.float64, False), (torch.float32, False), (torch.float32, True)):
    m = GA(k=16); m.load_state_dict(sd); m = m.cuda().to(dt); m.contig = contig
    xin = x.to(dt).clone().requires_grad_(True)
    coor, fo = m(xin, [N // 4, N // 8])
    gr = torch.autograd.grad((fo * w.to(dt)).sum(), [xin] + list(m.parameters()))
    print("split %s contig=%s: " % (dt, contig) + " ".join("%.2e" % (a.double() - r).abs().max() for a, r in zip(gr, ref)))

# variant B: split GEMMs with the weight gradient accumulated in float64
class _SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, x_k, x_q):
        C = x_k.size(1)
        w1 = w[:, :C]; wd = w[:, C:] - w1
        ctx.save_for_backward(w1, wd, x_k, x_q)
        return torch.matmul(w1, x_k), torch.matmul(wd, x_q)
    @staticmethod
    def backward(ctx, dP, dR):
        w1, wd, x_k, x_q = ctx.saved_tensors
        dxk = torch.matmul(w1.t(), dP); dxq = torch.matmul(wd.t(), dR)
        a = torch.einsum("bon,bcn->oc", dP.double(), x_k.double())
        b = torch.einsum("boq,bcq->oc", dR.double(), x_q.double())
        return torch.cat([a - b, b], dim=1).float(), dxk, dxq
class GB(G64):
    def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        idx = knn_idx(self.k, coor_k.float().transpose(-1, -2).contiguous(), coor_q.float().transpose(-1, -2).contiguous())
        conv, norm, act = layer[0], layer[1], layer[2]
        P, R = _SplitConv.apply(conv.weight.view(conv.out_channels, -1), x_k, x_q)
        return _TorchStage.apply(P, R, idx, norm.weight, norm.bias, norm.num_groups, norm.eps, act.negative_slope)
m = GB(k=16); m.load_state_dict(sd); m = m.cuda()
xin = x.clone().requires_grad_(True)
coor, fo = m(xin, [N // 4, N // 8])
gr = torch.autograd.grad((fo * w.float()).sum(), [xin] + list(m.parameters()))
print("split f32 + f64 weight-grad: " + " ".join("%.2e" % (a.double() - r).abs().max() for a, r in zip(gr, ref)))

# where does the split fp32 chain first deviate from the split f64 chain?
print("\nper-stage tensors, split f32 chain vs split f64 chain (max abs diff / max abs)")
class GC(G64):
    def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        idx = knn_idx(self.k, coor_k.float().transpose(-1, -2).contiguous(), coor_q.float().transpose(-1, -2).contiguous())
        conv, norm, act = layer[0], layer[1], layer[2]
        C = x_k.size(1)
        w = conv.weight.view(conv.out_channels, 2 * C)
        P = torch.matmul(w[:, :C], x_k); R = torch.matmul(w[:, C:] - w[:, :C], x_q)
        out = _TorchStage.apply(P, R, idx, norm.weight, norm.bias, norm.num_groups, norm.eps, act.negative_slope)
        rec = dict(x_k=x_k.detach(), x_q=x_q.detach(), P=P.detach(), R=R.detach(), out=out.detach(), idx=idx)
        P.register_hook(lambda t, rec=rec: rec.__setitem__("dP", t.detach()))
        R.register_hook(lambda t, rec=rec: rec.__setitem__("dR", t.detach()))
        out.register_hook(lambda t, rec=rec: rec.__setitem__("go", t.detach()))
        self.recs.append(rec)
        return out
recs = {}
for dt in (torch.float64, torch.float32):
    m = GC(k=16); m.load_state_dict(sd); m = m.cuda().to(dt); m.recs = []
    xin = x.to(dt).clone().requires_grad_(True)
    coor, fo = m(xin, [N // 4, N // 8])
    torch.autograd.grad((fo * w.to(dt)).sum(), [xin] + list(m.parameters()))
    recs[dt] = m.recs
for s, (a, b) in enumerate(zip(recs[torch.float32], recs[torch.float64])):
    print("stage %d idx equal %s  " % (s + 1, torch.equal(a["idx"], b["idx"])) + "  ".join("%s %.1e/%.1e" % (kk, (a[kk].double() - b[kk]).abs().max(), b[kk].abs().max()) for kk in ("x_k", "x_q", "P", "R", "out", "go", "dP", "dR")))
