"""Micro repro matrix: graph replay ordering against neighbouring stream work, with mitigations.
Tokens: K unrelated kernel, S device sync, E eager call, R copy new input + replay + check."""
import torch

torch.manual_seed(0)
FNS = {"sum_dim1": lambda t: t.sum(dim=1), "pointwise": lambda t: t * 2.0 + 1.0,
       "max_dim1": lambda t: torch.max(t, dim=1)[0], "mm": lambda t: t[0] @ t[0].t()}


def run(fn, flow, mitigation="none", shape=(1, 512, 1024)):
    with torch.no_grad():
        static_in = torch.randn(*shape, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn(static_in)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fn(static_in)
        res = []
        for tok in flow.split():
            if tok == "K":
                d = torch.zeros(1000, device="cuda") + 1.0
            elif tok == "S":
                torch.cuda.synchronize()
            elif tok == "E":
                fn(torch.randn(*shape, device="cuda"))
            elif tok == "R":
                x = torch.randn(*shape, device="cuda")
                want = fn(x)
                static_in.copy_(x)
                if mitigation in ("sync_before", "sync_both"):
                    torch.cuda.current_stream().synchronize()
                if mitigation == "event_before":
                    ev = torch.cuda.Event()
                    ev.record()
                    ev.synchronize()
                graph.replay()
                if mitigation in ("sync_after", "sync_both"):
                    torch.cuda.current_stream().synchronize()
                got = out.clone()
                res.append("ok" if float((got - want).abs().max()) == 0 else "BAD")
        return res


for name, fn in FNS.items():
    for flow in ("K S R R R", "R K R S R E R"):
        for mit in ("none", "sync_before", "sync_after", "sync_both", "event_before"):
            print(f"{name:10s} {flow:16s} {mit:13s}", run(fn, flow, mit), flush=True)
