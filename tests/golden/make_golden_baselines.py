"""Golden fixture for the Chamfer-only baselines (PCN, FoldingNet, TopNet), from the REFERENCE's Python.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_baselines.py

Uses make_golden.py's stand-ins for the absent packages and the oracle-backed `chamfer` module.  Two extra
accommodations for running the reference's constructors on this CPU-only container, neither touching its arithmetic:
`Tensor.cuda()` is made the identity (PCN.py:48 / FoldingNet.py:49 pin their lattices to the GPU at construction), and
TopNet is built with `num_pred: 2048` plus an `encoder_feature` key (its own YAML makes the constructor fail twice over,
see models/chamfer_baselines.py).  Writes tests/golden/baselines.npz: inputs and outputs only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from golden_util import seeded_state_dict, synthetic_batch  # noqa: E402


def main():
    mg.oracle.build()
    mg.install_standins()
    os.chdir(mg.REF)
    sys.path.insert(0, mg.REF)
    import builtins
    real_print, builtins.print = builtins.print, (lambda *a, **k: None)
    from utils.config import cfg_from_yaml_file
    from models import build_model_from_cfg
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.set_num_threads(8)

    # 8 clouds: TopNet's root layer normalises over the batch rows alone, and 3 rows made its gradients hinge on rounding
    partial, gt = synthetic_batch(8, 512, 2048, seed=41, structured=True)
    partial, gt = torch.from_numpy(partial), torch.from_numpy(gt)
    out = {"partial": partial.numpy(), "gt_head": gt.numpy()}
    shapes = {}
    for name in ("PCN", "FoldingNet", "TopNet"):
        cfg = cfg_from_yaml_file(f"cfgs/PCN_models/{name}.yaml").model
        if name == "TopNet":
            cfg.num_pred = 2048
            cfg.encoder_feature = cfg.encoder_channel
        model = build_model_from_cfg(cfg)
        sd = model.state_dict()
        shapes[name] = {k: list(v.shape) for k, v in sd.items()}
        model.load_state_dict(seeded_state_dict(sd, f"baseline/{name}", std=0.1))
        model.eval()
        with torch.no_grad():
            coarse, fine = model(partial)
        out[f"{name}_eval_coarse"] = coarse.numpy()[:, :512]
        out[f"{name}_eval_fine_head"] = fine.numpy()[:, :1024]
        model.train()
        ret = model(partial)
        g = gt
        l0, l1 = model.get_loss(ret, g)
        (l0 + l1).backward()
        out[f"{name}_train_fine_head"] = ret[1].detach().numpy()[:, :1024]
        out[f"{name}_losses"] = np.asarray([l0.item(), l1.item()], np.float64)
        out[f"{name}_grad_sq_norm"] = np.float64(sum(float((p.grad.double() ** 2).sum())
                                                     for p in model.parameters() if p.grad is not None))
        out[f"{name}_running_mean_first"] = model.first_conv[1].running_mean.numpy().copy()
    builtins.print = real_print
    np.savez_compressed(os.path.join(HERE, "baselines.npz"), **out)
    import json
    with open(os.path.join(HERE, "baselines_meta.json"), "w") as fh:
        json.dump({"state_dict": shapes}, fh, indent=1, sort_keys=True)
    print("baselines.npz", os.path.getsize(os.path.join(HERE, "baselines.npz")), "B")


if __name__ == "__main__":
    main()
