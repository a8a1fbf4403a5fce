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
OFF = {"PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0", "PTOPS_ATTN_FWD_F32": "1", "PTOPS_ATTN_BWD": "fused",
       "PTOPS_DEFER_FOLDS": "0", "PTOPS_FUSED_ADAMW": "0", "PTOPS_FUSED_LINEAR_GELU": "0"}
STEPS = 30
N_ULP = 5


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
    seed = int(os.environ.get("TRAJ_PERTURB_SEED", "0"))
    if seed:
        # rounding-level perturbation of the initial weights: every element moved by +-1 fp32 ulp (relative 2^-23 at
        # most), the smallest change a different summation order anywhere upstream could have produced
        gp = torch.Generator(device=device).manual_seed(seed)
        with torch.no_grad():
            for p in model.parameters():
                sign = torch.where(torch.rand(p.shape, device=device, generator=gp) < 0.5, -1.0, 1.0)
                p.mul_(1.0 + sign * 2.0 ** -23)
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
    # fp32 summation orders; "ulp k": the plain arm started from weights moved by +-1 ulp (seed k) - the envelope of what
    # rounding-level differences alone do to these 30 steps (round-2 review: one control arm is not a spread)
    arms = [("fast", {}), ("plain", OFF), ("control", {**OFF, "TRAJ_UNTUNED_GEMM": "1"})]
    arms += [(f"ulp {k}", {**OFF, "TRAJ_PERTURB_SEED": str(k)}) for k in range(1, N_ULP + 1)]
    arms += [("fast ulp 1", {"TRAJ_PERTURB_SEED": "1"})]
    for name, env in arms:
        path = os.path.join(ROOT, "gpurun_out", f"trajectory_{name.replace(' ', '_')}.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env={**os.environ, **env})
        out[name] = json.load(open(path))

    def rel_diff(x, y):
        return [max(abs(a0 - b0) / abs(b0), abs(a1 - b1) / abs(b1)) for (a0, a1), (b0, b1) in zip(x, y)]

    fast = rel_diff(out["fast"], out["plain"])
    ctrl = {k: rel_diff(v, out["plain"]) for k, v in out.items() if k not in ("fast", "plain")}
    env = [max(c[i] for c in ctrl.values()) for i in range(STEPS)]
    for i in range(STEPS):
        if i < 6 or i % 5 == 0 or i == STEPS - 1:
            print(f"step {i:2d}: plain {out['plain'][i][0] * 1e3:9.4f} {out['plain'][i][1] * 1e3:9.4f}   "
                  f"|fast - plain| {fast[i]:.2e}   envelope of {len(ctrl)} rounding-level arms {env[i]:.2e}  ("
                  + " ".join(f"{c[i]:.1e}" for c in ctrl.values()) + ")")
    worst = {k: max(c) for k, c in ctrl.items()}
    print(f"largest relative difference from the plain arm over {STEPS} steps: fast {max(fast):.2e}; rounding-level arms "
          + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    print(f"first step: fast {fast[0]:.2e}; rounding-level arms " + ", ".join(f"{k} {c[0]:.2e}" for k, c in ctrl.items()))
    inside = sum(1 for i in range(STEPS) if fast[i] <= 1.5 * env[i] + 1e-5)
    print(f"steps at which the fast arm lies within 1.5 x the envelope: {inside} / {STEPS}")
    assert fast[0] < 1e-4, "the first step must agree to rounding"
    assert max(fast) < 2.0 * max(max(worst.values()), 1e-2), "the fast paths drift further from the plain ones than rounding alone explains"
    assert inside >= STEPS - 3, "the fast arm leaves the envelope of the rounding-level arms"
    last = [sum(v[1] for v in out[k][-5:]) / 5 for k in ("fast", "plain")]
    assert abs(last[0] - last[1]) < 0.1 * last[1], "the arms do not converge alike"


if __name__ == "__main__":
    main()
