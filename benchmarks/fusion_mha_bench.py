"""Wall time of the image / segmentation variants' train step with the fusion layers' 48-wide heads on the padded
MFMA attention path vs nn.MultiheadAttention's own forward (same model object, flag flipped between runs)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg, pctransformer  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402
from imgadapointr_amd.utils.config import cfg_from_yaml_file  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402

dev = torch.device("cuda", 0)
_native.load()
enable_tuned_gemm()
os.chdir(ROOT)
B = 16
for which, yaml in (("img", "cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml"),
                    ("seg", "cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml")):
    cfg = cfg_from_yaml_file(yaml)
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(dev).train()
    if which == "seg":
        for p in model.base_model.segmentator.parameters():
            p.requires_grad = False
    step = TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
    partial = (torch.rand(B, 2048, 3) - 0.5).to(dev)
    gt = (torch.rand(B, 16384, 3) - 0.5).to(dev)
    extra = [torch.rand(B, 3, 224, 224, device=dev)]
    if which == "seg":
        v = torch.zeros(B, 16, device=dev)
        v[:, 0] = 1
        extra.append(v)
    res = {}
    for rnd in range(2):
        for flag in (True, False):
            pctransformer.PADDED_HEAD_MFMA_ATTENTION = flag
            for _ in range(4):
                step(partial, gt, tuple(extra))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(15):
                step(partial, gt, tuple(extra))
            torch.cuda.synchronize()
            res.setdefault(flag, []).append((time.perf_counter() - t0) / 15 * 1e3)
    model.eval()
    ev = {}
    with torch.no_grad():
        for flag in (True, False, True, False):
            pctransformer.PADDED_HEAD_MFMA_ATTENTION = flag
            for _ in range(3):
                model(partial, *extra)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model(partial, *extra)
            torch.cuda.synchronize()
            ev.setdefault(flag, []).append(round((time.perf_counter() - t0) / 10 * 1e3, 2))
    print(f"{which}: eval forward padded {ev[True]} ms, nn.MultiheadAttention {ev[False]} ms", flush=True)
    pctransformer.PADDED_HEAD_MFMA_ATTENTION = True
    print(f"{which}: padded-head MFMA {min(res[True]):.2f} ms {res[True]}   nn.MultiheadAttention {min(res[False]):.2f} ms {res[False]}",
          flush=True)
    del model, step
    torch.cuda.empty_cache()
