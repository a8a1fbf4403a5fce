"""Diagnostic: which stage of the eval forward deviates when a HIP graph of it is replayed after unrelated work."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_checks as pc  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models.Transformer_utils import linear_on_broadcast_concat  # noqa: E402

_native.load()
model, _ = pc.build_adapointr("cuda")
model.eval()
print("tunable enabled:", torch.cuda.tunable.is_enabled(), "tuning:", torch.cuda.tunable.tuning_is_enabled(), flush=True)


class Staged(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, xyz):
        m = self.m
        q, coarse, dl = m.base_model(xyz)
        B, M, C = q.shape
        gf_tok = m._increase_dim(q)
        gf = torch.max(gf_tok, dim=1)[0]
        rf = linear_on_broadcast_concat(m.reduce_map, gf, torch.cat([q, coarse], dim=-1))
        rel = m.decode_head(rf)
        pts = (rel + coarse.unsqueeze(-2)).reshape(B, -1, 3).contiguous()
        return coarse, q, gf_tok, gf, rf, rel, pts


staged = Staged(model).eval()
names = ["coarse", "q", "gf_tok", "gf", "rf", "rel", "pts"]


def run(variant):
    with torch.no_grad():
        x0 = torch.rand(1, 2048, 3, device="cuda") - 0.5
        static_in = x0.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                staged(static_in)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = staged(static_in)
        flat = list(out)
        if variant == "dummy":
            dummy = torch.zeros(1000, device="cuda") + 1.0
        torch.cuda.synchronize()
        gen = torch.Generator(device="cuda").manual_seed(77)
        for i in range(2):
            x = torch.rand(1, 2048, 3, device="cuda", generator=gen) - 0.5
            want = staged(x)
            static_in.copy_(x)
            graph.replay()
            got = [t.clone() for t in flat]
            print(variant, i, {n: float((a - b).abs().max()) for n, a, b in zip(names, got, want)}, flush=True)


for v in ("none", "dummy", "none"):
    run(v)
