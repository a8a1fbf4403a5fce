"""How much of the side branches of the graphed img / seg train step is NOT hidden behind the main branch.

A kernel trace cannot answer this (under rocprofv3 the branches of a replayed HIP graph hardly overlap:
profiles/round4/timeline_graphed_img_r4_v1.txt), so the step is timed with parts of a branch removed:

    shipped        the step as bench.py runs it (GraphedTrainStep, parallel branches on)
    serial         the same with PTOPS_OVERLAP_IMAGE_BRANCH-style forks off (one branch)
    no_img_bwd     image encoder frozen: its backward disappears, its forward stays
    no_img         image tokens are a constant: the encoder disappears completely
    no_seg         (seg only) GDANet part embeddings are a constant: normals + GDANet disappear
    no_img_no_seg  both
    aten_topk      (seg only) GDANet's feature-space top-k on ATen instead of csrc/topk.hip
    literal_conv   (seg only) GDANet's local-operator stages as the reference writes them (edge tensor + Conv2d)
    nchw_stage     (seg only) first conv before the gather, but BatchNorm / ReLU / conv / max on ATen + MIOpen in NCHW

`shipped - no_img` is the exposed cost of the image branch (0 = fully hidden), `serial - no_img` its serial cost.

    python3 benchmarks/branch_exposure.py [img|seg] [steps]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402

import bench  # noqa: E402
from imgadapointr_amd.models import pctransformer  # noqa: E402
from imgadapointr_amd.models.segmentation import GDANet_util  # noqa: E402
from imgadapointr_amd.utils.graph_runner import GraphedTrainStep  # noqa: E402


def constant_tokens(fn_name):
    """Replace base_model.<fn_name> by its first result, detached (nothing of the branch is launched afterwards)."""
    def wrap(model):
        base = model.base_model
        real = getattr(base, fn_name)
        cache = {}

        def const(*a, **kw):
            if "t" not in cache:
                with torch.no_grad():
                    cache["t"] = real(*a, **kw).detach().clone()
            return cache["t"]
        setattr(base, fn_name, const)
        return model
    return wrap


def freeze_image_encoder(model):
    for p in model.base_model.im_encoder.parameters():
        p.requires_grad = False
    return model


def chain(*fs):
    def wrap(model):
        for f in fs:
            model = f(model)
        return model
    return wrap


def run(which, name, wrap, steps, overlap=True, native_topk=True, fused_conv=True, token_major=True):
    pctransformer.OVERLAP_IMAGE_BRANCH = overlap
    GDANet_util.USE_NATIVE_TOPK = native_topk
    GDANet_util.FUSED_LOCAL_CONV = fused_conv
    GDANet_util.TOKEN_MAJOR_STAGE = token_major
    model, cfg, spec, step, batch, B = bench.make_workload(which, torch.device("cuda:0"), wrap=wrap)
    g = GraphedTrainStep(step, warmup=3, verify=False)
    s, _ = bench.time_steps(g, batch, steps, 3 + 1 + 3)
    ok = g.enabled and g.replays >= steps
    print(f"{which:4s} {name:14s} {s * 1e3:7.2f} ms/step  graphed={ok}", flush=True)
    del g, step, model
    torch.cuda.empty_cache()
    return {"variant": name, "ms_per_step": round(s * 1e3, 3), "graphed": ok}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "img"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = [run(which, "shipped", None, steps),
            run(which, "serial", None, steps, overlap=False),
            run(which, "no_img_bwd", freeze_image_encoder, steps),
            run(which, "no_img", constant_tokens("_image_tokens"), steps)]
    if which == "seg":
        rows.append(run(which, "aten_topk", None, steps, native_topk=False))
        rows.append(run(which, "literal_conv", None, steps, fused_conv=False))
        rows.append(run(which, "nchw_stage", None, steps, token_major=False))
        rows.append(run(which, "no_seg", constant_tokens("_seg_tokens"), steps))
        rows.append(run(which, "no_img_no_seg", chain(constant_tokens("_image_tokens"), constant_tokens("_seg_tokens")), steps))
    rows.append(run(which, "shipped", None, steps))
    print(json.dumps({"model": which, "steps": steps, "rows": rows}))


if __name__ == "__main__":
    main()
