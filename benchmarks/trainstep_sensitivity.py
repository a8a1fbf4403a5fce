"""What a rounding-level difference in the image branch does to the gradients of the B = 2 train-step fixture.

Round-3's one failing GPU run (`gpurun_out/pytest_gpu_r3_v15.log`: total |g|^2 3.1255 vs 3.1926, outputs and losses inside
their bounds) could not say which tensor moved.  This runs the same step on the CPU (oracle backend: deterministic) with the
image encoder's output multiplied by (1 + eps * N(0,1)) for a handful of seeds — a stand-in for "MIOpen picked another
convolution kernel" — and prints, per seed, the total |g|^2, the tensors that moved most and whether the set of selected
queries / the global-feature arg-max changed.

    python benchmarks/trainstep_sensitivity.py [case] [eps] [n_seeds] [img|seg_in|seg_out]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(case, eps, seed, device="cpu", where="img"):
    import parity_checks as pc
    from golden_util import trainstep_inputs
    g = pc.load(f"trainstep_{case}.npz")
    model, cfg, spec = pc.build_trainstep_model(case, device)
    model.train()
    partial, gt, img, cls = trainstep_inputs(case)
    extra = []
    if img is not None:
        extra.append(pc.t(img, device))
    elif spec["takes_img"]:
        extra.append(None)
    if cls is not None:
        extra.append(pc.t(cls, device))
    for i, name in enumerate(("dp1", "dp2")):
        if f"dropout_mask_{i}" in g.files:
            shape = tuple(int(x) for x in g[f"dropout_mask_{i}_shape"])
            mask = np.unpackbits(g[f"dropout_mask_{i}"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)
            setattr(model.base_model.segmentator, name, pc._FixedDropout(pc.t(mask, device), 0.4))
    rec = {}
    hooks = []
    gen = torch.Generator().manual_seed(seed)

    def noisy(x):
        return x * (1 + eps * torch.randn(x.shape, generator=gen).to(x.device))
    if eps > 0 and where == "img":
        hooks.append(model.base_model.im_encoder.register_forward_hook(lambda m, a, o: noisy(o)))
    if eps > 0 and where == "seg_out":
        hooks.append(model.base_model.segmentator.register_forward_hook(lambda m, a, o: (noisy(o[0]),) + tuple(o[1:])))
    if eps > 0 and where == "seg_in":       # the point cloud handed to the segmentation branch (predicted coarse cloud / centres)
        hooks.append(model.base_model.segmentator.register_forward_pre_hook(lambda m, a: (noisy(a[0]),) + tuple(a[1:])))
    hooks.append(model.base_model.query_ranking.register_forward_hook(
        lambda m, a, o: rec.__setitem__("order", o.detach().squeeze(-1).argsort(dim=1, descending=True)[:, :512].cpu())))
    ret = model(pc.t(partial, device), *extra, denoise_noise=pc.t(g["train_noise"], device))
    l_den, l_rec = model.get_loss(ret, pc.t(gt, device), spec["epoch"])
    (l_den + l_rec).backward()
    for h in hooks:
        h.remove()
    gsq = {n: float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters() if p.grad is not None}
    return gsq, (l_den.item(), l_rec.item()), rec["order"], g


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "img"
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    where = sys.argv[4] if len(sys.argv) > 4 else "img"
    device = "cuda" if torch.cuda.is_available() else "cpu"
    if device == "cpu":
        from imgadapointr_amd import _native
        from oracle_backend import OracleBackend
        _native.set_backend_for_testing(OracleBackend())
    base, loss0, order0, g = run(case, 0.0, 0, device)
    tot0 = sum(base.values())
    print(f"case {case} on {device}, eps {eps:g} at {where}: fixture |g|^2 {float(g['grad_sq_norm_total']):.6f}, this build unperturbed {tot0:.6f}, "
          f"losses x1000 {loss0[0] * 1e3:.4f} / {loss0[1] * 1e3:.4f}")
    top = sorted(base, key=base.get, reverse=True)[:8]
    print("largest tensors:", ", ".join(f"{n.replace('base_model.', '')} {base[n] / tot0:.3f}" for n in top))
    for s in range(n):
        gsq, loss, order, _ = run(case, eps, 100 + s, device, where)
        tot = sum(gsq.values())
        moved = sorted(base, key=lambda k: abs(gsq[k] - base[k]), reverse=True)[:4]
        same_set = all(set(order[b].tolist()) == set(order0[b].tolist()) for b in range(order.shape[0]))
        print(f"seed {s}: |g|^2 {tot:.6f} ({(tot - tot0) / tot0:+.2e}), losses x1000 {loss[0] * 1e3:.4f} / {loss[1] * 1e3:.4f}, "
              f"query set {'same' if same_set else 'CHANGED'}, order {'same' if torch.equal(order, order0) else 'changed'}; "
              "moved: " + ", ".join(f"{k.replace('base_model.', '')} {(gsq[k] - base[k]) / tot0:+.2e}" for k in moved))


if __name__ == "__main__":
    main()
