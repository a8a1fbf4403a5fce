"""Sweep vs brute force on uniform and on 'bench-like' data (clustered prediction vs uniform gt).
Needs a tuning build (PTOPS_TUNE_* honoured)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat

def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

torch.manual_seed(0)
B, n = 32, 16384
gt = torch.rand(B, n, 3, device="cuda") - 0.5
uni = torch.rand(B, n, 3, device="cuda") - 0.5
centres = torch.cat([torch.randn(B, 256, 3, device="cuda") * 0.01, torch.rand(B, 256, 3, device="cuda") - 0.5], 1)
clustered = (centres[:, :, None, :] + torch.randn(B, 512, 32, 3, device="cuda") * 0.003).reshape(B, n, 3).contiguous()
surface = (gt + torch.randn_like(gt) * 0.01)
for name, pred in (("uniform", uni), ("bench-like clusters", clustered), ("near-surface", surface)):
    res = {}
    for sweep in (0, 1):
        for q in ((0,) if not sweep else (0, 2, 4)):
            os.environ["PTOPS_TUNE_SWEEP"] = str(sweep); os.environ["PTOPS_TUNE_SWEEP_Q"] = str(q)
            out = nat.chamfer_forward(pred, gt)
            if "ref" in res: assert all(torch.equal(a, b) for a, b in zip(out, res["ref"]))
            else: res["ref"] = out
            res[(sweep, q)] = t(lambda: nat.chamfer_forward(pred, gt))
    print(name, {k: round(v, 3) for k, v in res.items() if k != "ref"})
