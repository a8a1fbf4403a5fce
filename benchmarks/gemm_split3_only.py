"""One shape of the split3 GEMM a few times (target of rocprofv3 passes). argv: M N K iterations"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native
_native.load()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (18432, 1024, 1024)
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
for _ in range(it):
    _native.gemm_split3(x, w, b)
torch.cuda.synchronize()
print("done")
