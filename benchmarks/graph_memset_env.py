"""Do HIP-graph MEMSET nodes replay reliably on this stack, and does a runtime flag change that?

Round 2 found (benchmarks/graph_diag2.py, graph_diag3.py) that a captured hipMemsetAsync stops taking effect on replays that
follow a host synchronisation preceded by other work.  A whole train step of the image / segmentation models contains
memsets this build does not issue (ATen's multi-block reductions, rocPRIM-backed sort / top-k / scatter, MIOpen), so a
graphed step needs them to work.  This runs three small graphs - an explicit hipMemsetAsync through ctypes, ATen's
`sum(dim=1)`, and a pointwise control - through the flows that exposed the problem, once per runtime setting, each in its own
process (the flags are read when the HIP runtime loads).

    python benchmarks/graph_memset_env.py            # driver: one child per setting
"""
import ctypes
import os
import subprocess
import sys

SETTINGS = [{}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}, {"DEBUG_HIP_FORCE_GRAPH_QUEUES": "1"},
            {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"}]


def child():
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int

    def memset_then_add(t):          # out = 0 (hipMemsetAsync node) ; out += t
        out = torch.empty_like(t)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(out.data_ptr()), 0, out.numel() * 4,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        out += t
        return out

    fns = {"hipMemsetAsync": (memset_then_add, (1, 512, 1024)), "aten_sum_dim1": (lambda t: t.sum(dim=1), (1, 512, 1024)),
           "aten_topk": (lambda t: t.topk(8, dim=-1)[0], (4, 2048, 256)),
           "pointwise": (lambda t: t * 2 + 1, (1, 512, 1024))}

    def run(name, flow):
        fn, shape = fns[name]
        with torch.no_grad():
            static_in = torch.rand(*shape, device="cuda")
            for _ in range(2):
                fn(static_in)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn(static_in)
            res = []
            for tok in flow.split():
                if tok == "K":
                    _ = torch.zeros(1000, device="cuda") + 1.0
                elif tok == "S":
                    torch.cuda.synchronize()
                elif tok == "R":
                    x = torch.rand(*shape, device="cuda")
                    want = fn(x)
                    static_in.copy_(x)
                    graph.replay()
                    res.append("ok" if float((out - want).abs().max()) == 0 else "BAD")
            return res

    for name in fns:
        for flow in ("R R R", "K S R R R", "R K R S R R", "R S R S R K S R"):
            print(f"  {name:16s} {flow:18s} {' '.join(run(name, flow))}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for env in SETTINGS:
            print("setting:", env or "default", flush=True)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=300)
            print(r.stdout, end="")
            if r.returncode != 0:
                print("  child failed:", r.stderr[-1500:])
