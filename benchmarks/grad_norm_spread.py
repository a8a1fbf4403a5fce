"""Run-to-run spread of the total squared gradient norm of the fixture train step (tests/parity_checks.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parity_checks as pc
g = pc.load("adapointr.npz"); mt = pc.meta()
if len(sys.argv) > 1 and sys.argv[1] == "warm":       # what the earlier tests of the file do first
    pc.check_functional("cuda"); pc.check_grouper("cuda"); pc.check_blocks("cuda"); pc.check_eval_forward("cuda")
for i in range(6):
    model, _ = pc.build_adapointr("cuda")
    model.train()
    ret = model(pc.t(g["partial"], "cuda"), denoise_noise=pc.t(g["train_noise"], "cuda"))
    l1, l2 = model.get_loss(ret, pc.t(g["gt"], "cuda"))
    (l1 + l2).backward()
    tot = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None)
    print("run %d: total %.6f  (reference %.6f, rel diff %+.2e)  losses x1000 %.4f %.4f" % (i, tot, mt["grad_sq_norm_total"], tot / mt["grad_sq_norm_total"] - 1, l1.item() * 1000, l2.item() * 1000))
