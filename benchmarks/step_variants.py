"""A/B of host-side evaluation choices for the train step (interleaved rounds, one process):
SDPA vs explicit attention, split broadcast-concat Linear, fused graph attention, BLAS backend,
TunableOp.  Prints ms/step per variant."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_cfg, N_GT, N_PARTIAL  # noqa: E402
from imgadapointr_amd.models import Transformer_utils as TU, build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402


def time_steps(step, partial, gt, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(partial, gt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tunable", action="store_true")
    ap.add_argument("--tunable-file", default="gpurun_out/tunableop.csv")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--tune-ms", type=int, default=30)
    ap.add_argument("--tune-iters", type=int, default=20)
    ap.add_argument("--skip-ab", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(dev).train()
    step = TrainStep(model, builder.build_optimizer(model, cfg))
    partial = (torch.rand(32, N_PARTIAL, 3) - 0.5).to(dev)
    gt = (torch.rand(32, N_GT, 3) - 0.5).to(dev)

    def setv(sdpa=True, split=True, fused=True, blas=None):
        TU.USE_SDPA, TU.SPLIT_BROADCAST_LINEAR, TU.FUSED_GRAPH_ATTENTION = sdpa, split, fused
        if blas:
            torch.backends.cuda.preferred_blas_library(blas)

    variants = {
        "default(sdpa,split,fused)": dict(),
        "no-split": dict(split=False),
        "no-fused-graph": dict(fused=False),
        "explicit-attention": dict(sdpa=False),
        "rocblas": dict(blas="hipblas"),
        "hipblaslt": dict(blas="hipblaslt"),
    }
    res = {k: [] for k in variants}
    for rnd in range(0 if a.skip_ab else 3):
        for name, v in variants.items():
            setv(**v)
            time_steps(step, partial, gt, 2)
            res[name].append(time_steps(step, partial, gt, a.steps))
    setv(blas="default")
    for k, ts in res.items():
        if not ts:
            continue
        print(f"{k:32s} min {min(ts):7.2f} ms  med {sorted(ts)[1]:7.2f} ms")
    print("preferred blas now:", torch.backends.cuda.preferred_blas_library())

    if a.tunable:
        setv()
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(a.tune_ms)
        torch.cuda.tunable.set_max_tuning_iterations(a.tune_iters)
        torch.cuda.tunable.set_filename(a.tunable_file)
        t0 = time.perf_counter()
        time_steps(step, partial, gt, 2)
        print(f"tunableop tuning pass: {time.perf_counter() - t0:.1f} s", flush=True)
        torch.cuda.tunable.tuning_enable(False)
        ts = [time_steps(step, partial, gt, a.steps) for _ in range(3)]
        print(f"{'tunableop(tuned)':32s} min {min(ts):7.2f} ms  med {sorted(ts)[1]:7.2f} ms")
        print("tunable results:", len(torch.cuda.tunable.get_results()))


if __name__ == "__main__":
    main()
