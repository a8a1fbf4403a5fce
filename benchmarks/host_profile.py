"""Where the HOST time of a train step goes (cProfile over a few steps; the GPU runs behind, nothing waits for it).

    python benchmarks/host_profile.py [adapointr|img|seg] [steps]
"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder, runner  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "adapointr"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    device = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    spec = bench.MODELS[which]
    cfg = bench.load_cfg(which)
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(device)
    if spec["cls"]:
        for p in model.base_model.segmentator.parameters():
            p.requires_grad = False
    model.train()
    train = runner.TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    B = spec["batch"]
    partial = (torch.rand(B, bench.N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(B, bench.N_GT, 3) - 0.5).to(device)
    extra = []
    if spec["img"]:
        extra.append(torch.rand(B, 3, 224, 224).to(device))
    if spec["cls"]:
        cls_vec = torch.zeros(B, 16)
        cls_vec[torch.arange(B), torch.tensor([0, 4, 8, 15])[torch.arange(B) % 4]] = 1.0
        extra.append(cls_vec.to(device))
    extra = tuple(extra)
    for _ in range(6):
        train(partial, gt, extra)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train(partial, gt, extra)
    t_host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps
    print(f"{which}: host enqueue {t_host * 1e3:.2f} ms per step, device-complete {t_all * 1e3:.2f} ms per step")
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(steps):
        train(partial, gt, extra)
    prof.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    st = pstats.Stats(prof, stream=out)
    st.sort_stats("tottime").print_stats(45)
    print(out.getvalue().replace(ROOT + "/", "")[:9000])
    out = io.StringIO()
    pstats.Stats(prof, stream=out).sort_stats("cumtime").print_stats(40)
    print(out.getvalue().replace(ROOT + "/", "")[:8000])


if __name__ == "__main__":
    main()
