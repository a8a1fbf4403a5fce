"""Experiment: the whole train step (forward + loss + backward + clip + fused AdamW) captured in one HIP graph.
Measures eager vs graph-replay step time on one GPU (bs 32).  Chamfer runs with a fixed algorithm inside the graph
(the adaptive policy reads statistics back on the host, which a captured stream cannot do)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(dev).train()
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (no_decay if p.dim() == 1 or name.endswith(".bias") else decay).append(p)
    kw = dict(cfg.optimizer.kwargs)
    opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": kw["weight_decay"]}],
                            fused=True, capturable=True, **{k: v for k, v in kw.items()})
    params = [p for p in model.parameters() if p.requires_grad]
    orig = _native.chamfer_forward
    _native.chamfer_forward = lambda a, b, algorithm=None: orig(a, b, algorithm=_native.CHAMFER_BRUTE)
    partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3) - 0.5).to(dev)
    gt = (torch.rand(bench.BATCH, bench.N_GT, 3) - 0.5).to(dev)
    noise = torch.zeros(bench.BATCH, 64, 3, device=dev)

    def step():
        ret = model(partial)
        l1, l2 = model.get_loss(ret, gt, 1)
        (l1 + l2).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 10.0)
        opt.step()
        return l1.detach(), l2.detach()

    def eager():
        out = step()
        opt.zero_grad(set_to_none=True)
        return out

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            eager()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(f"eager  : {timed(eager):.2f} ms/step", flush=True)

    opt.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    print(f"graph  : {timed(g.replay):.2f} ms/step   losses x1000 {[round(float(v) * 1000, 3) for v in static_out]}", flush=True)


if __name__ == "__main__":
    main()
