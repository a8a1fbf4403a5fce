"""Hypothesis: hipGraph MEMSET nodes stop working after a host synchronisation.  Test with a graph whose only memset
is libptops' own hipMemsetAsync (split Chamfer), and look at the graph's node list."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imgadapointr_amd import _native  # noqa: E402

_native.load()
torch.manual_seed(0)


def chamfer(t):     # t [4, 512+2048, 3]: split brute force -> hipMemsetAsync(0xFF) + atomicMin + unpack
    d1, d2, i1, i2 = _native.chamfer_forward(t[:, :512].contiguous(), t[:, 512:].contiguous(), algorithm=_native.CHAMFER_BRUTE)
    return d1


FNS = {"chamfer_split": (chamfer, (4, 2560, 3)), "sum_dim1": (lambda t: t.sum(dim=1), (1, 512, 1024)),
       "pointwise": (lambda t: t * 2 + 1, (1, 512, 1024))}


def run(name, flow, dump=False):
    fn, shape = FNS[name]
    with torch.no_grad():
        static_in = torch.rand(*shape, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn(static_in)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        if dump:
            graph.enable_debug_mode()
        with torch.cuda.graph(graph):
            out = fn(static_in)
        if dump:
            path = f"/tmp/graph_{name}.dot"
            try:
                graph.debug_dump(path)
                txt = open(path).read()
                import re
                kinds = re.findall(r'label="([^"\\]*)', txt)
                print(name, "dot bytes", len(txt), "memset nodes:", txt.upper().count("MEMSET"), "labels:", kinds[:12], flush=True)
            except Exception as exc:
                print(name, "debug_dump failed:", repr(exc), flush=True)
            return None
        res = []
        for tok in flow.split():
            if tok == "K":
                d = torch.zeros(1000, device="cuda") + 1.0
            elif tok == "S":
                torch.cuda.synchronize()
            elif tok == "R":
                x = torch.rand(*shape, device="cuda")
                want = fn(x)
                static_in.copy_(x)
                graph.replay()
                got = out.clone()
                res.append("ok" if float((got - want).abs().max()) == 0 else "BAD")
        return res


for name in FNS:
    run(name, "", dump=True)
    for flow in ("R R R", "K S R R R", "R K R S R R"):
        print(f"{name:14s} {flow:14s}", run(name, flow), flush=True)
