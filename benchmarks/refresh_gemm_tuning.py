"""Extends the shipped per-shape GEMM selections (imgadapointr_amd/tuning/gemm_gfx950.csv) with the shapes of every
model family of the path: runs a few train steps and an eval forward of each on one MI355X with TunableOp tuning the
shapes it has no entry for (existing entries are kept), and writes the merged file to gpurun_out/gemm_gfx950.csv.

    python benchmarks/refresh_gemm_tuning.py && cp gpurun_out/gemm_gfx950.csv imgadapointr_amd/tuning/
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402
from imgadapointr_amd.utils.config import cfg_from_yaml_file  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _native.load()
    print(enable_tuned_gemm(max_tuning_ms=50, max_tuning_iters=30), flush=True)
    os.chdir(ROOT)
    n0 = len(torch.cuda.tunable.get_results())
    for yaml, B, img, cls in (("cfgs/PCN_models/AdaPoinTr.yaml", 32, False, False),
                              ("cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml", 16, True, False),
                              ("cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml", 16, True, True),
                              ("cfgs/PCN_models/PCN.yaml", 32, False, False),
                              ("cfgs/PCN_models/FoldingNet.yaml", 32, False, False),
                              ("cfgs/PCN_models/TopNet.yaml", 32, False, False)):
        cfg = cfg_from_yaml_file(yaml)
        torch.manual_seed(0)
        model = build_model_from_cfg(cfg.model).to(dev).train()
        if cls:
            for p in model.base_model.segmentator.parameters():
                p.requires_grad = False
        step = TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
        extra = []
        if img:
            extra.append(torch.rand(B, 3, 224, 224, device=dev))
        if cls:
            v = torch.zeros(B, 16, device=dev)
            v[:, 0] = 1
            extra.append(v)
        t0 = time.perf_counter()
        for b in (B, 1):                                   # training batch, then the one-cloud validation forward
            partial = (torch.rand(b, 2048, 3) - 0.5).to(dev)
            gt = (torch.rand(b, 16384, 3) - 0.5).to(dev)
            ex = tuple(e[:b] for e in extra)
            if b == B:
                for _ in range(2):
                    step(partial, gt, ex)
            model.eval()
            with torch.no_grad():
                model(partial, *ex)
            model.train()
        torch.cuda.synchronize()
        print(f"{yaml}: {len(torch.cuda.tunable.get_results())} entries after {time.perf_counter() - t0:.1f} s", flush=True)
        del model, step
        torch.cuda.empty_cache()
    out = os.path.join(ROOT, "gpurun_out", "gemm_gfx950.csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tun = torch.cuda.tunable
    with open(out, "w") as fh:                         # the format TunableOp itself writes and reads
        for key, val in tun.get_validators():
            fh.write(f"Validator,{key},{val}\n")
        for op, params, solution, ms in tun.get_results():
            fh.write(f"{op},{params},{solution},{ms}\n")
    work = tun.get_filename()
    if os.path.exists(work):
        import shutil
        shutil.copyfile(work, out + ".tunableop_own_file")
    print(f"{n0} -> {len(torch.cuda.tunable.get_results())} entries, written to {out}")


if __name__ == "__main__":
    main()
