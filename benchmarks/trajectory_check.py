"""End-to-end safety check of the round-2 fast paths: the same 30 train steps at bs = 32 (same seed, same synthetic batches) with
every switch on and with every switch off (library GEMMs, f32-MFMA attention, immediate folds, stock clip + AdamW; the
Chamfer algorithms are bit-identical and tested as such); the two loss trajectories must agree to fp32-training noise.

    python benchmarks/trajectory_check.py            # runs both arms in subprocesses and compares
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OFF = {"PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0", "PTOPS_ATTN_FWD_F32": "1", "PTOPS_ATTN_BWD_F32": "1",
       "PTOPS_DEFER_FOLDS": "0", "PTOPS_FUSED_ADAMW": "0", "PTOPS_FUSED_LINEAR_GELU": "0"}
STEPS = 30


def arm(out_path):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from imgadapointr_amd import _native
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.tools import builder
    from imgadapointr_amd.tools.runner import TrainStep
    from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm
    device = torch.device("cuda", 0)
    _native.load()
    if os.environ.get("TRAJ_UNTUNED_GEMM") != "1":
        enable_tuned_gemm()
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(device)
    model.train()
    step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    g = torch.Generator().manual_seed(1)
    losses = []
    for _ in range(STEPS):
        partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3, generator=g) - 0.5).to(device)
        gt = (torch.rand(bench.BATCH, bench.N_GT, 3, generator=g) - 0.5).to(device)
        a, b = step(partial, gt)
        losses.append((float(a), float(b)))
    with open(out_path, "w") as fh:
        json.dump(losses, fh)


def main():
    if len(sys.argv) > 1:
        return arm(sys.argv[1])
    out = {}
    # "control": the plain arm with the library's default GEMM kernels instead of the tuned selection - another set of
    # fp32 summation orders, i.e. the size of the difference that rounding alone produces over these steps
    for name, env in (("fast", {}), ("plain", OFF), ("control", {**OFF, "TRAJ_UNTUNED_GEMM": "1"})):
        path = os.path.join(ROOT, "gpurun_out", f"trajectory_{name}.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env={**os.environ, **env})
        out[name] = json.load(open(path))
    def rel_diff(x, y):
        return [max(abs(a0 - b0) / abs(b0), abs(a1 - b1) / abs(b1)) for (a0, a1), (b0, b1) in zip(x, y)]

    fast, ctrl = rel_diff(out["fast"], out["plain"]), rel_diff(out["control"], out["plain"])
    for i in range(STEPS):
        if i < 6 or i % 5 == 0 or i == STEPS - 1:
            print(f"step {i:2d}: plain {out['plain'][i][0] * 1e3:9.4f} {out['plain'][i][1] * 1e3:9.4f}   "
                  f"|fast - plain| {fast[i]:.2e}   |control - plain| {ctrl[i]:.2e}")
    print(f"largest relative difference over {STEPS} steps: fast {max(fast):.2e}, control {max(ctrl):.2e}; "
          f"first step: fast {fast[0]:.2e}, control {ctrl[0]:.2e}")
    assert fast[0] < 1e-4, "the first step must agree to rounding"
    assert max(fast) < 5 * max(max(ctrl), 1e-2), "the fast paths drift further from the plain ones than rounding alone explains"
    last = [sum(v[1] for v in out[k][-5:]) / 5 for k in ("fast", "plain")]
    assert abs(last[0] - last[1]) < 0.1 * last[1], "the arms do not converge alike"


if __name__ == "__main__":
    main()
