"""Train-step time of the three model families of the path (BASELINE configs[1], [2], [4] shapes) and the
validation forward, on one MI355X with synthetic inputs.  Not the headline metric (bench.py is) — a coverage check
that the image / segmentation variants run on the same native kernels at sensible speed."""
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


def timed(fn, n):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    if len(sys.argv) == 1:
        # one fresh process per variant: after a bs=32 AdaPoinTr run in the same process the segmentation variant
        # measured 50 ms instead of 42.6 ms (allocator / workspace state), which says nothing about the variant
        import subprocess
        for key in ("AdaPoinTr (PCN)", "ImgResNetEnc", "ImgEncSegDec"):
            subprocess.run([sys.executable, os.path.abspath(__file__), key], check=True)
        return
    dev = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    os.chdir(ROOT)
    rows = []
    only = sys.argv[1] if len(sys.argv) > 1 else ""       # substring filter on the variant name
    if os.environ.get("PTOPS_NOPAD_MHA") == "1":
        from imgadapointr_amd.models import pctransformer
        pctransformer.PADDED_HEAD_MFMA_ATTENTION = False
    for name, yaml, B, img, cls in (("AdaPoinTr (PCN)", "cfgs/PCN_models/AdaPoinTr.yaml", 32, False, False),
                                    ("ImgResNetEncAdaPoinTr (ImgPCN)", None, 16, True, False),
                                    ("ImgEncSegDecAdaPoinTr (SegImgPCN)", None, 16, True, True)):
        if only and only not in name:
            continue
        if yaml is None:
            sub = "ImgPCN_models" if not cls else "SegImgPCN_models"
            cands = sorted(f for f in os.listdir(os.path.join("cfgs", sub)) if f.endswith(".yaml"))
            yaml = os.path.join("cfgs", sub, cands[0])
        cfg = cfg_from_yaml_file(yaml)
        torch.manual_seed(0)
        model = build_model_from_cfg(cfg.model).to(dev).train()
        if cls and hasattr(model, "base_model") and hasattr(model.base_model, "segmentator"):
            for p in model.base_model.segmentator.parameters():
                p.requires_grad = False
        step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
        partial = (torch.rand(B, 2048, 3) - 0.5).to(dev)
        gt = (torch.rand(B, 16384, 3) - 0.5).to(dev)
        extra = []
        if img:
            extra.append(torch.rand(B, 3, 224, 224, device=dev))
        if cls:
            v = torch.zeros(B, 16, device=dev)
            v[:, 0] = 1
            extra.append(v)
        ms = timed(lambda: step(partial, gt, tuple(extra)), 10)
        model.eval()
        with torch.no_grad():
            ms_eval = timed(lambda: model(partial, *extra), 10)
        rows.append((name, yaml, B, ms, ms_eval))
        print(f"{name:36s} {yaml:60s} bs={B:3d}  train step {ms:7.2f} ms ({B / ms * 1e3:7.1f} clouds/s)   "
              f"eval forward {ms_eval:6.2f} ms", flush=True)
        del model, step
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
