"""What do the two Chamfer operands of the benchmark step look like (random-init network, uniform targets)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(dev).train()
    partial = (torch.rand(8, bench.N_PARTIAL, 3) - 0.5).to(dev)
    gt = (torch.rand(8, bench.N_GT, 3) - 0.5).to(dev)
    with torch.no_grad():
        ret = model(partial)
    for name, t in (("coarse", ret[0] if isinstance(ret, (tuple, list)) else None), ("dense", ret[-1])):
        if t is None or t.dim() != 3:
            continue
        print(name, tuple(t.shape), "min", t.amin(dim=(0, 1)).tolist(), "max", t.amax(dim=(0, 1)).tolist(),
              "std per axis", t.std(dim=1).mean(0).tolist())
    dense = ret[-1].contiguous()
    d1, d2, i1, i2 = _native.chamfer_forward(dense, gt, algorithm=_native.CHAMFER_BRUTE)
    print("pred->gt  NN dist: mean %.4f max %.4f" % (d1.sqrt().mean(), d1.sqrt().max()))
    print("gt->pred  NN dist: mean %.4f max %.4f" % (d2.sqrt().mean(), d2.sqrt().max()))
    print("distinct pred points used as NN of gt: %.1f per cloud" % (sum(len(torch.unique(r)) for r in i2) / len(i2)))
    for alg, nm in ((_native.CHAMFER_BRUTE, "brute"), (_native.CHAMFER_SWEEP, "sweep")):
        for _ in range(3):
            _native.chamfer_forward(dense, gt, algorithm=alg)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _native.chamfer_forward(dense, gt, algorithm=alg)
        e.record()
        torch.cuda.synchronize()
        print(nm, "B=8: %.3f ms" % (s.elapsed_time(e) / 10))


if __name__ == "__main__":
    main()
