"""Train-step time of the Chamfer-only baselines (cfgs/PCN_models/{PCN,FoldingNet,TopNet}.yaml, bs = 32, PCN shapes)
in the two execution forms of models/pointwise.py on one MI355X: token-major with split first layers (the product
path) and the literal channel-first form the reference runs (concatenate, Conv1d on [B, C, N])."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg, pointwise  # noqa: E402
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
    dev = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    os.chdir(ROOT)
    B = 32
    partial = (torch.rand(B, 2048, 3) - 0.5).to(dev)
    gt = (torch.rand(B, 16384, 3) - 0.5).to(dev)
    for name in ("PCN", "FoldingNet", "TopNet"):
        cfg = cfg_from_yaml_file(f"cfgs/PCN_models/{name}.yaml")
        line = f"{name:11s} bs={B}"
        for form, flag in (("token-major", True), ("literal", False)):
            pointwise.TOKEN_MAJOR = flag
            torch.manual_seed(0)
            model = build_model_from_cfg(cfg.model).to(dev).train()
            step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
            torch.cuda.reset_peak_memory_stats()
            ms = timed(lambda: step(partial, gt, ()), 10)
            mem = torch.cuda.max_memory_allocated() / 2 ** 30
            model.eval()
            with torch.no_grad():
                ms_eval = timed(lambda: model(partial), 10)
            line += f"   {form}: train {ms:7.2f} ms ({B / ms * 1e3:7.1f} clouds/s, peak {mem:5.1f} GiB), eval {ms_eval:6.2f} ms"
            del model, step
            torch.cuda.empty_cache()
        pointwise.TOKEN_MAJOR = True
        print(line, flush=True)


if __name__ == "__main__":
    main()
