"""Peak device memory of the bs = 32 train step with the weight gradients queued until the end of backward (their dY operands
stay alive that long) and computed per layer.

    python benchmarks/peak_memory.py
"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from imgadapointr_amd import _native
from imgadapointr_amd.models import build_model_from_cfg
from imgadapointr_amd.tools import builder, runner
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm
dev = torch.device("cuda", 0); _native.load(); enable_tuned_gemm()
cfg = bench.load_cfg(); torch.manual_seed(0)
model = build_model_from_cfg(cfg.model).to(dev).train()
step = runner.TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
p = (torch.rand(32, 2048, 3) - 0.5).to(dev); g = (torch.rand(32, 16384, 3) - 0.5).to(dev)
for grouped in (True, False):
    _native.GROUPED_WGRAD = grouped
    for _ in range(3): step(p, g)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    for _ in range(3): step(p, g)
    torch.cuda.synchronize()
    print(f"grouped weight gradients {grouped}: peak allocated {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
