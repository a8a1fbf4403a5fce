"""Which Python lines issue the aten::copy_ / cat / contiguous kernels of a train step (torch.profiler with stacks)."""
import collections
import os
import sys

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
    model = build_model_from_cfg(cfg.model).to(device).train()
    step = TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
    partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(bench.BATCH, bench.N_GT, 3) - 0.5).to(device)
    for _ in range(4):
        step(partial, gt)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(partial, gt)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    names = sys.argv[1:] or ["aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_"]
    for e in prof.events():
        if e.name in names and e.device_time > 0:
            frames = [f for f in (e.stack or []) if "imgadapointr_amd" in f or "bench" in f][:2]
            key = (e.name, str(e.input_shapes)[:60], " <- ".join(f.split("/")[-1][:70] for f in frames))
            agg[key][0] += e.device_time
            agg[key][1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    for (name, shp, where), (t, n) in rows[:60]:
        print(f"{t / 1e3:7.3f} ms x{n:3d} {name:12s} {shp:60s} {where}")


if __name__ == "__main__":
    main()
