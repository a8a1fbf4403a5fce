"""Top GPU kernels of one train step of a model variant (torch.profiler): img | seg | PCN | FoldingNet | TopNet."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imgadapointr_amd import _native
from imgadapointr_amd.models import build_model_from_cfg
from imgadapointr_amd.tools import builder
from imgadapointr_amd.tools.runner import TrainStep
from imgadapointr_amd.utils.config import cfg_from_yaml_file
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm

which = sys.argv[1] if len(sys.argv) > 1 else "img"
os.chdir(ROOT)
dev = torch.device("cuda", 0)
_native.load(); enable_tuned_gemm()
if len(sys.argv) > 2 and sys.argv[2] == "nopad":
    from imgadapointr_amd.models import pctransformer as _PT
    _PT.PADDED_HEAD_MFMA_ATTENTION = False
if len(sys.argv) > 2 and sys.argv[2] == "benchmark":
    torch.backends.cudnn.benchmark = True
yaml = {"img": "cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml",
        "seg": "cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml"}.get(which, f"cfgs/PCN_models/{which}.yaml")
cfg = cfg_from_yaml_file(yaml)
torch.manual_seed(0)
model = build_model_from_cfg(cfg.model).to(dev).train()
if which == "seg":
    for p in model.base_model.segmentator.parameters():
        p.requires_grad = False
step = TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
B = 16 if which in ("img", "seg") else 32
partial = (torch.rand(B, 2048, 3) - 0.5).to(dev); gt = (torch.rand(B, 16384, 3) - 0.5).to(dev)
extra = [torch.rand(B, 3, 224, 224, device=dev)] if which in ("img", "seg") else []
if which == "seg":
    v = torch.zeros(B, 16, device=dev); v[:, 0] = 1; extra.append(v)
for _ in range(4):
    step(partial, gt, tuple(extra))
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step(partial, gt, tuple(extra))
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if getattr(e, "self_device_time_total", 0) > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print("total device time per step %.2f ms" % (sum(e.self_device_time_total for e in rows if not e.key.startswith("aten::") and not e.key[0].isupper() or True) / 3e3 / 2))
for e in rows[:45]:
    print(f"{e.self_device_time_total / 3e3:8.3f} ms  x{e.count / 3:6.1f}  {e.key[:110]}")
