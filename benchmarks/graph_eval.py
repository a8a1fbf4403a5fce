"""Validation-style forward (bs = 1, eval mode): eager vs one HIP graph replay."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from imgadapointr_amd import _native
from imgadapointr_amd.models import build_model_from_cfg
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm

def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

dev = torch.device("cuda", 0)
_native.load(); enable_tuned_gemm()
torch.manual_seed(0)
model = build_model_from_cfg(bench.load_cfg().model).to(dev).eval()
for B in (1, 8):
    x = (torch.rand(B, 2048, 3) - 0.5).to(dev)
    with torch.no_grad():
        eager = timed(lambda: model(x))
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): model(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(x)
        torch.cuda.synchronize()
        ref = model(x)
        g.replay(); torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print(f"bs={B}: eager {eager:.2f} ms  graph {timed(g.replay):.2f} ms  identical outputs: {same}")
