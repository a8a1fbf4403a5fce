import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native
from imgadapointr_amd.models import build_model_from_cfg
from imgadapointr_amd.utils.config import cfg_from_yaml_file
from imgadapointr_amd.utils.graph_runner import GraphedForward
import time
dev = torch.device("cuda", 0)
for name, yaml, seg in (("img", "cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml", False), ("seg", "cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml", True)):
    cfg = cfg_from_yaml_file(yaml)
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(dev).eval()
    x = (torch.rand(1, 2048, 3, device=dev) - 0.5); img = torch.rand(1, 3, 224, 224, device=dev)
    extra = [img]
    if seg:
        v = torch.zeros(1, 16, device=dev); v[:, 0] = 1; extra.append(v)
    gf = GraphedForward(model)
    with torch.no_grad():
        want = model(x, *extra)
        got = gf(x, *extra)
        got2 = gf(x, *extra)
        again = model(x, *extra)
        print(name, "graph enabled:", gf.enabled, "replays:", gf.replays, "equal:", all(torch.equal(a, b) for a, b in zip(got2, want)),
              "max |graph - eager|: %.3e" % max((a - b).abs().max().item() for a, b in zip(got2, want)),
              " eager vs eager again: %.3e" % max((a - b).abs().max().item() for a, b in zip(again, want)))
        def t(fn, n=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        print("   eager %.2f ms   graphed-or-fallback %.2f ms" % (t(lambda: model(x, *extra)), t(lambda: gf(x, *extra))))
