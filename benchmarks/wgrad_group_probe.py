"""Grouped weight gradients (csrc/gemm_split3.hip, `ptops_gemm_f32_split3_wgrad_grouped`) on the weight-gradient GEMMs
of one real bs = 32 train step: the (dY, X) pairs the step queues are captured and replayed as

  (a) the grouped launch, for several job lengths (16-row steps per job),
  (b) what the step ran before: the library's dY^T.X (+ the column-sum kernel for the bias gradient) or the per-layer
      split-bf16 kernel where `_use_split3_wgrad` picked it,
  (c) the library alone.

Errors are against float64 on a few of the layers.

    python benchmarks/wgrad_group_probe.py
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import Transformer_utils as TU  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder, runner  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    device = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(device).train()
    step = runner.TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(bench.BATCH, bench.N_GT, 3) - 0.5).to(device)
    for _ in range(3):
        step(partial, gt)
    captured = []
    flush = _native.wgrad_flush

    def spy():
        captured.extend((q[0], q[1], q[3] != 0) for q in _native._wgrad_queue)
        flush()

    _native.wgrad_flush = spy
    step(partial, gt)
    _native.wgrad_flush = flush
    torch.cuda.synchronize()
    problems = [(dy.clone(), x.clone(), wb) for dy, x, wb in captured]
    del captured
    hist = collections.Counter((dy.size(0), dy.size(1), x.size(1), wb) for dy, x, wb in problems)
    flops = sum(2.0 * dy.size(0) * dy.size(1) * x.size(1) for dy, x, _ in problems)
    print(f"{len(problems)} weight gradients queued by one step, {flops / 1e9:.1f} GFLOP")
    for (M, N, K, wb), c in sorted(hist.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
        print(f"   x{c:3d}  M {M:6d}  N {N:5d}  K {K:5d}  {'+db' if wb else '   '}  {2e-9 * M * N * K * c:8.1f} GFLOP")

    outs = []

    def grouped():
        outs.clear()
        for dy, x, wb in problems:
            outs.append(_native.wgrad_defer(dy, x, wb))
        _native.wgrad_flush()

    def before():
        for dy, x, wb in problems:
            if TU._use_split3_wgrad(dy, x):
                _, db = _native.gemm_split3_wgrad(dy, x, wb)
                if wb and db is None:
                    _native.colsum(dy)
            else:
                dy.t() @ x
                if wb:
                    _native.colsum(dy)

    def library():
        for dy, x, wb in problems:
            dy.t() @ x
            if wb:
                _native.colsum(dy)

    t_before, t_lib = timed(before), timed(library)
    print(f"per layer, as the step did before : {t_before:7.3f} ms")
    print(f"per layer, library + column sums  : {t_lib:7.3f} ms")
    for target in (96, 144, 192, 288, 384, 576, 1152):
        _native.set_option("PTOPS_WGRAD_GROUP_STEPS", target)
        _native._wgrad_plans.clear()
        t = timed(grouped)
        (_, _, ws_bytes) = next(iter(_native._wgrad_plans.values()))
        print(f"grouped, {target:5d} steps per job      : {t:7.3f} ms   {flops / t / 1e9:6.1f} TFLOP/s   workspace {ws_bytes / 2**20:7.1f} MiB", flush=True)
    _native.set_option("PTOPS_WGRAD_GROUP_STEPS", 0)
    _native._wgrad_plans.clear()
    for occ3 in (0, 1, 0, 1):
        _native.set_option("PTOPS_WGRAD_OCC3", occ3)
        t = timed(grouped, 20)
        print(f"grouped, default job length, {'three' if occ3 else 'two'} workgroups per CU: {t:7.3f} ms   {flops / t / 1e9:6.1f} TFLOP/s", flush=True)
    _native.set_option("PTOPS_WGRAD_OCC3", 0)

    grouped()
    torch.cuda.synchronize()
    worst = collections.defaultdict(float)
    seen = set()
    for (dy, x, wb), (dw, db) in zip(problems, outs):
        shape = (dy.size(0), dy.size(1), x.size(1))
        if shape in seen:
            continue
        seen.add(shape)
        ref = dy.double().t() @ x.double()
        scale = ref.abs().max().item()
        worst["grouped dW"] = max(worst["grouped dW"], (dw.double() - ref).abs().max().item() / scale)
        worst["library dW"] = max(worst["library dW"], ((dy.t() @ x).double() - ref).abs().max().item() / scale)
        if wb:                                           # column sums: error against the sum of magnitudes
            rb, mag = dy.double().sum(0), dy.double().abs().sum(0)
            worst["grouped db"] = max(worst["grouped db"], ((db.double() - rb).abs() / mag).max().item())
            worst["colsum kernel db"] = max(worst["colsum kernel db"], ((_native.colsum(dy).double() - rb).abs() / mag).max().item())
            worst["torch sum(0) db"] = max(worst["torch sum(0) db"], ((dy.sum(0).double() - rb).abs() / mag).max().item())
    print("max error / output range vs float64 over the distinct shapes:", {k: f"{v:.2e}" for k, v in worst.items()})
    grouped()
    again = [(dw.clone(), None if db is None else db.clone()) for dw, db in outs]
    grouped()
    same = all(torch.equal(a, c) and (b is None or torch.equal(b, d)) for (a, b), (c, d) in zip(again, outs))
    print("bit-identical across runs:", same)


if __name__ == "__main__":
    main()
