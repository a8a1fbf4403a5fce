"""A/B of run-time switches on the WHOLE bs = 32 train step in one process, interleaved rounds (one box, one model, one
optimizer state: the rows are comparable to ~0.1 ms).  Each row flips ONE switch away from the shipped default.

    python benchmarks/step_ab.py [rounds] [steps per round] [adapointr|img|seg]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import Transformer_utils as TU  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.extensions import chamfer_dist  # noqa: E402
from imgadapointr_amd.pointnet2_ops import pointnet2_utils  # noqa: E402
from imgadapointr_amd.tools import builder, runner  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    which = sys.argv[3] if len(sys.argv) > 3 else "adapointr"          # adapointr | img | seg (bench.py --model)
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

    def step(p, g):
        return train(p, g, extra)

    def opt(name, value):
        return (lambda: _native.set_option(name, value), lambda: _native.set_option(name, 0))

    def attr(obj, name, value):
        old = getattr(obj, name)
        return (lambda: setattr(obj, name, value), lambda: setattr(obj, name, old))

    arms = [("shipped defaults", (lambda: None, lambda: None)),
            ("Chamfer backward with atomics (reference form)", attr(_native, "DETERMINISTIC_BACKWARD", False)),
            ("forward GEMM at three workgroups per CU", opt("PTOPS_GEMM_OCC3", 1)),
            ("attention without the XCD job order", opt("PTOPS_ATTN_NO_XCD_REMAP", 1)),
            ("immediate folds (and weight gradients)", attr(runner, "DEFER_FOLDS", False)),
            ("weight gradients per layer, not grouped", attr(_native, "GROUPED_WGRAD", False)),
            ("k_map and v_map as two projections", attr(TU, "FUSED_KV_PROJECTION", False)),
            ("farthest point sampling on a side stream", attr(pointnet2_utils, "OVERLAP_FPS", True)),
            ("Chamfer reductions as the literal torch chain", attr(chamfer_dist, "FUSED_REDUCTIONS", False)),
            ("Chamfer MFMA with fragment prefetch", opt("PTOPS_CHAMFER_MFMA_PREFETCH", 1))]
    for _ in range(8):
        step(partial, gt)
    torch.cuda.synchronize()
    times = {name: [] for name, _ in arms}
    for _ in range(rounds):
        for name, (on, off) in arms:
            on()
            step(partial, gt)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(steps):
                step(partial, gt)
            e.record()
            torch.cuda.synchronize()
            off()
            times[name].append(s.elapsed_time(e) / steps)
    base = sorted(times["shipped defaults"])[rounds // 2]
    for name, _ in arms:
        v = sorted(times[name])
        print(f"{name:52s} median {v[rounds // 2]:7.3f} ms/step  (min {v[0]:7.3f})   delta vs defaults {v[rounds // 2] - base:+.3f} ms", flush=True)


if __name__ == "__main__":
    main()
