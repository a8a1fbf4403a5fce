"""Referee for tests/parity_checks.py::check_baseline_forms: which of the two execution forms of the Chamfer-only
baselines (token-major rows with split first layers / the reference's literal channel-first stacks) is the one that
moves when they disagree?  Both forms run in fp32 on the GPU and are compared with a float64 evaluation of the SAME
modules (literal form, Chamfer-L2 by brute force in float64): outputs, total gradient, and the parameters whose gradient
is furthest from float64.  The float64 run is also repeated with the inputs perturbed by one fp32 ulp, which gives the
condition of the function itself (how far rounding-level input noise moves the gradient).

    python benchmarks/baseline_forms_referee.py            # needs the GPU (libptops)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import seeded_state_dict, synthetic_batch  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg, pointwise  # noqa: E402
from imgadapointr_amd.utils.config import EasyDict  # noqa: E402

CASES = (dict(NAME='PCN', num_pred=1024, encoder_channel=1024),
         dict(NAME='FoldingNet', num_pred=1024, encoder_channel=256),
         dict(NAME='TopNet', node_feature=8, num_pred=2048, encoder_channel=248, nlevels=6))


class ChamferL2F64(torch.nn.Module):
    def forward(self, a, b):
        d = (a.unsqueeze(2) - b.unsqueeze(1)).pow(2).sum(-1)
        return d.min(2)[0].mean() + d.min(1)[0].mean()


def run(cfg, partial, gt, token_major, dtype):
    pointwise.TOKEN_MAJOR = token_major
    try:
        model = build_model_from_cfg(EasyDict(cfg))
        model.load_state_dict(seeded_state_dict(model.state_dict(), "forms/" + cfg['NAME'], std=0.1))
        model = model.to(partial.device).to(dtype).train()
        if dtype == torch.float64:
            model.loss_func = ChamferL2F64()
        ret = model(partial.to(dtype))
        l0, l1 = model.get_loss(ret, gt.to(dtype))
        (l0 + l1).backward()
    finally:
        pointwise.TOKEN_MAJOR = True
    return ret[1].detach().double(), {n: p.grad.double() for n, p in model.named_parameters()}


def rel(a, b):
    return float((a - b).norm() / b.norm())


def grad_err(g, ref):
    tot = sum(float((g[n] - ref[n]).pow(2).sum()) for n in ref) ** 0.5 / sum(float(ref[n].pow(2).sum()) for n in ref) ** 0.5
    norm = sum(float(ref[n].pow(2).sum()) for n in ref) ** 0.5
    # (a bias in front of a train-mode BatchNorm has a gradient that is zero in exact arithmetic: left out)
    worst = sorted(((float((g[n] - ref[n]).norm() / ref[n].norm()), n) for n in ref if float(ref[n].norm()) > 1e-6 * norm),
                   reverse=True)[:3]
    return tot, worst


def main():
    if torch.cuda.is_available():
        _native.load()
        dev = torch.device("cuda", 0)
    else:       # dry run of the script's own logic on the CPU oracle backend (tests only; says nothing about the GPU)
        from oracle import ref as oracle
        from oracle_backend import OracleBackend
        oracle.build()
        _native.set_backend_for_testing(OracleBackend())
        dev = torch.device("cpu")
    partial, gt = synthetic_batch(8, 256, 512, seed=51, structured=True)
    partial, gt = torch.from_numpy(partial).to(dev), torch.from_numpy(gt).to(dev)
    for cfg in CASES:
        o64, g64 = run(cfg, partial, gt, False, torch.float64)
        eps = torch.finfo(torch.float32).eps
        sign = torch.where(torch.rand(partial.shape, device=dev, generator=torch.Generator(dev).manual_seed(1)) < 0.5, -1., 1.)
        o64p, g64p = run(cfg, partial.double() * (1 + eps * sign.double()), gt, False, torch.float64)
        rows = [("float64, inputs moved by 1 fp32 ulp", o64p, g64p)]
        for name, tm in (("token-major fp32", True), ("literal fp32", False)):
            o, g = run(cfg, partial, gt, tm, torch.float32)
            rows.append((name, o, g))
        print(f"== {cfg['NAME']}")
        for name, o, g in rows:
            tot, worst = grad_err(g, g64)
            print(f"  {name:38s} output vs f64 {rel(o, o64):.3e}   gradient vs f64 {tot:.3e}   worst: "
                  + ", ".join(f"{n} {e:.2e}" for e, n in worst))
        tot, worst = grad_err(rows[1][2], rows[2][2])
        print(f"  {'token-major vs literal (form vs form)':38s} output {rel(rows[1][1], rows[2][1]):.3e}   gradient {tot:.3e}")


if __name__ == "__main__":
    main()
