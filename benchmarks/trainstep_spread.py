"""How well-conditioned are the gradients of the round-3 train-step fixtures (tests/golden/trainstep_*.npz)?
Runs the production step of tests/parity_checks.py::check_trainstep on the GPU with the fast paths on and with them
switched off (library GEMMs, f32-MFMA attention, immediate folds, stock clip + AdamW, literal edge-conv / graph attention),
one process per arm, and prints each arm's total squared gradient norm and its distance from the reference's value and from
the other arm — the spread that rounding alone produces on this fixture.

    python benchmarks/trainstep_spread.py [img|seg|segenc ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = [("fast paths (default)", {}),
        ("library GEMMs", {"PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0", "PTOPS_FUSED_LINEAR_GELU": "0"}),
        ("library GEMMs + f32 attention", {"PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0", "PTOPS_FUSED_LINEAR_GELU": "0",
                                           "PTOPS_ATTN_FWD_F32": "1", "PTOPS_ATTN_BWD": "fused"}),
        ("all off + literal edge-conv / graph attention", {"PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0",
                                                           "PTOPS_FUSED_LINEAR_GELU": "0", "PTOPS_ATTN_FWD_F32": "1",
                                                           "PTOPS_ATTN_BWD": "fused", "PTOPS_DEFER_FOLDS": "0",
                                                           "PTOPS_FUSED_ADAMW": "0", "PTOPS_SPREAD_LITERAL": "1"})]


def child(case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import parity_checks as pc
    from golden_util import trainstep_inputs
    from imgadapointr_amd import _native
    from imgadapointr_amd.tools import builder
    from imgadapointr_amd.tools.runner import TrainStep
    _native.load()
    if os.environ.get("PTOPS_SPREAD_LITERAL") == "1":
        from imgadapointr_amd.models import Transformer_utils as TU, base_blocks as BB
        TU.FUSED_GRAPH_ATTENTION = False
        BB.FUSED_EDGE_CONV = False
    dev = "cuda"
    g = pc.load(f"trainstep_{case}.npz")
    model, cfg, spec = pc.build_trainstep_model(case, dev)
    model.train()
    named = dict(model.named_parameters())
    step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    partial, gt, img, cls = trainstep_inputs(case)
    extra = []
    if img is not None:
        extra.append(pc.t(img, dev))
    elif spec["takes_img"]:
        extra.append(None)
    if cls is not None:
        extra.append(pc.t(cls, dev))
    for i, name in enumerate(("dp1", "dp2")):
        if f"dropout_mask_{i}" in g.files:
            shape = tuple(int(x) for x in g[f"dropout_mask_{i}_shape"])
            mask = np.unpackbits(g[f"dropout_mask_{i}"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)
            setattr(model.base_model.segmentator, name, pc._FixedDropout(pc.t(mask, dev), 0.4))
    grads = {}
    step.before_update = lambda: grads.update({n: p.grad.detach().double().cpu() for n, p in named.items() if p.grad is not None})
    l_den, l_rec = step(pc.t(partial, dev), pc.t(gt, dev), tuple(extra), epoch=spec["epoch"],
                        denoise_noise=pc.t(g["train_noise"], dev))
    out = {"loss": [l_den.item(), l_rec.item()], "total": sum(float((v ** 2).sum()) for v in grads.values()),
           "want_total": float(g["grad_sq_norm_total"]), "want_loss": [float(g["loss_denoised"]), float(g["loss_recon"])]}
    torch.save(grads, os.environ["PTOPS_SPREAD_OUT"])
    print("RESULT " + json.dumps(out), flush=True)


def main():
    import torch
    cases = sys.argv[1:] or ["img", "seg", "segenc"]
    for case in cases:
        print(f"== {case}")
        saved = []
        for i, (name, env) in enumerate(ARMS):
            e = dict(os.environ)
            e.update(env)
            e["PTOPS_SPREAD_CASE"] = case
            e["PTOPS_SPREAD_OUT"] = f"/tmp/spread_{case}_{i}.pt"
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"  [{name}] FAILED: {r.stderr[-600:]}")
                continue
            o = json.loads(line[0][7:])
            grads = torch.load(e["PTOPS_SPREAD_OUT"])
            saved.append((name, grads))
            print(f"  [{name:48s}] losses x1000 {o['loss'][0] * 1e3:.5f} / {o['loss'][1] * 1e3:.5f}  "
                  f"(ref {o['want_loss'][0] * 1e3:.5f} / {o['want_loss'][1] * 1e3:.5f})   |g|^2 {o['total']:.6f}  "
                  f"vs reference {o['total'] / o['want_total'] - 1:+.3e}", flush=True)
        base = saved[-1][1] if saved else None
        for name, grads in saved[:-1]:
            num = sum(float(((grads[n] - base[n]) ** 2).sum()) for n in base) ** 0.5
            den = sum(float((base[n] ** 2).sum()) for n in base) ** 0.5
            print(f"  gradient of [{name}] vs [{saved[-1][0]}]: relative L2 {num / den:.3e}")


if __name__ == "__main__":
    if "PTOPS_SPREAD_CASE" in os.environ:
        child(os.environ["PTOPS_SPREAD_CASE"])
    else:
        main()
