"""Host-side (enqueue) time of the bs = 32 train step next to its GPU time: how far the CPU runs ahead of the GPU.
With eight ranks per node the host side must stay well below the device side, or the job turns launch-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(device)
    model.train()
    step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(bench.BATCH, bench.N_GT, 3) - 0.5).to(device)
    for _ in range(8):
        step(partial, gt)
    torch.cuda.synchronize()
    n = int(os.environ.get("HOST_STEPS", "2"))   # few enough steps that the launch queue does not throttle the host
    t0 = time.perf_counter()
    for _ in range(n):
        step(partial, gt)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host enqueue {t_host / n * 1e3:.2f} ms per step, device-complete {t_all / n * 1e3:.2f} ms per step "
          f"({os.cpu_count()} host cores visible)")


if __name__ == "__main__":
    main()
