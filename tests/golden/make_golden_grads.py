"""Element-wise gradients of a handful of small AdaPoinTr parameters from the REFERENCE's own Python.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grads.py

Same recipe and same inputs as make_golden.py section (4)+(5) (the train forward of `adapointr.npz`: its
`partial`, `gt`, the seed-123 denoise jitter and `seeded_state_dict` weights); writes `adapointr_grads.npz` with the
full gradient vectors of bias / norm parameters.  The squared-norm check of the other fixture cannot see a sign flip or
a permutation inside a gradient; these vectors can (bias-gradient hand-off).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

# biases fed by the LayerNorm-backward / GELU-backward hand-off, plain column-sum biases and norm weights
PARAMS = [
    "base_model.encoder.blocks.blocks.0.merge_map.bias",
    "base_model.encoder.blocks.blocks.0.mlp.fc1.bias",
    "base_model.encoder.blocks.blocks.0.mlp.fc2.bias",
    "base_model.encoder.blocks.blocks.0.attn.proj.bias",
    "base_model.encoder.blocks.blocks.0.attn.qkv.bias",
    "base_model.encoder.blocks.blocks.3.norm1.weight",
    "base_model.encoder.blocks.blocks.3.norm2.bias",
    "base_model.decoder.blocks.blocks.0.self_attn_merge_map.bias",
    "base_model.decoder.blocks.blocks.0.cross_attn_merge_map.bias",
    "base_model.decoder.blocks.blocks.0.cross_attn.proj.bias",
    "base_model.decoder.blocks.blocks.0.cross_attn.v_map.bias",   # (k_map.bias has a zero gradient: softmax is shift-invariant)
    "base_model.decoder.blocks.blocks.5.mlp.fc1.bias",
    "base_model.decoder.blocks.blocks.5.mlp.fc2.bias",
    "base_model.decoder.blocks.blocks.5.norm_q.weight",
    "base_model.decoder.blocks.blocks.5.norm_v.bias",
    "base_model.decoder.blocks.blocks.7.self_attn.proj.bias",
    "base_model.mlp_query.2.bias",
    "base_model.increase_dim.2.bias",
    "increase_dim.1.weight",
    "increase_dim.1.bias",
    "increase_dim.3.bias",
    "reduce_map.bias",
    "decode_head.layer.fc1.bias",
    "decode_head.layer.fc2.bias",
]


def main():
    mg.oracle.build()
    mg.install_standins()
    os.chdir(mg.REF)
    sys.path.insert(0, mg.REF)
    import builtins
    real_print = builtins.print
    from utils.config import cfg_from_yaml_file
    from models import build_model_from_cfg

    torch.set_num_threads(8)
    g = np.load(os.path.join(HERE, "adapointr.npz"))
    cfg = cfg_from_yaml_file('cfgs/PCN_models/AdaPoinTr.yaml')
    model = build_model_from_cfg(cfg.model)
    model.load_state_dict(mg.seeded_state_dict(model.state_dict(), "adapointr"))
    model.train()
    torch.manual_seed(123)                      # reproduces `train_noise` of adapointr.npz
    builtins.print = lambda *a, **k: None       # get_loss prints seven debug lines per call (AdaPoinTr.py:195-206)
    try:
        ret = model(torch.from_numpy(g["partial"]))
        l_den, l_rec = model.get_loss(ret, torch.from_numpy(g["gt"]))
    finally:
        builtins.print = real_print
    assert abs(l_den.item() - float(g["loss_denoised"])) < 1e-7 and abs(l_rec.item() - float(g["loss_recon"])) < 1e-7, \
        "not the train forward of adapointr.npz"
    (l_den + l_rec).backward()
    named = dict(model.named_parameters())
    out = {}
    for n in PARAMS:
        assert named[n].grad is not None, n
        out[n] = named[n].grad.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "adapointr_grads.npz"), **out)
    print("wrote adapointr_grads.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
