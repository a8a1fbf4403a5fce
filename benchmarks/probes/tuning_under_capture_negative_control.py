import os, sys
sys.path.insert(0, os.getcwd())
import torch
from imgadapointr_amd.utils import graph_runner
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm
enable_tuned_gemm()
class Off:
    def __enter__(self): return self
    def __exit__(self, *a): return False
graph_runner._no_gemm_tuning = Off
class M(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.w = torch.nn.Parameter(torch.randn(328, 200))
    def forward(self, x):
        w = self.w[:, :136] if torch.cuda.is_current_stream_capturing() else self.w
        return x[:, :w.size(1)] @ w.t()
m = M().cuda().eval()
gf = graph_runner.GraphedForward(m, verify=False)
with torch.no_grad():
    x = torch.randn(1000, 200, device="cuda")
    try:
        for _ in range(3):
            gf(x)
        print("NEGATIVE CONTROL: enabled =", gf.enabled, "|", (gf.disabled_reason or "")[:160])
    except Exception as e:
        print("NEGATIVE CONTROL raised:", type(e).__name__, str(e)[:160])
