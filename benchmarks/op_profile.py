"""aten-level view of one train step (which ops own the element-wise / copy / reduce kernels).

    python benchmarks/op_profile.py [adapointr|img|seg] > gpurun_out/op_profile.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    which = sys.argv[1] if len(sys.argv) > 1 else "adapointr"          # adapointr | img | seg
    model, cfg, spec, _step, (partial, gt, extra), B = bench.make_workload(which, device)

    def step(p, g):
        return _step(p, g, extra)
    for _ in range(5):
        step(partial, gt)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    try:
        xcfg = torch._C._profiler._ExperimentalConfig(verbose=True)       # python stacks in key_averages(group_by_stack_n)
    except Exception:
        xcfg = None
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True,
                 experimental_config=xcfg) as prof:
        for _ in range(3):
            step(partial, gt)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=50, max_name_column_width=70))
    # the small aten ops (everything but the GEMMs), grouped by input shape, per step
    skip = ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::linear")
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key.startswith("aten::") and e.key not in skip and getattr(e, "self_device_time_total", 0) > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print(f"\nnon-GEMM aten ops: {tot / 3e3:.3f} ms per step")
    for e in rows[:90]:
        print(f"{e.self_device_time_total / 3e3:8.3f} ms  x{e.count / 3:6.1f}  {e.key:38s} {str(e.input_shapes)[:150]}")
    # the same small ops by CALL SITE (innermost frame inside this package), launches and device time per step
    import collections
    site = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        if not (e.key.startswith("aten::") and e.key not in skip and getattr(e, "self_device_time_total", 0) > 0):
            continue
        frames = [f for f in (e.stack or []) if "imgadapointr_amd" in f or "bench.py" in f]
        where = " <- ".join(f.split("imgadapointr_amd/")[-1].strip() for f in frames[:2]) if frames else "(autograd engine: backward of a torch op)"
        shp = str([x for x in e.input_shapes if x])[:60] if where.startswith("tools/runner.py") or where.startswith("(autograd") else ""
        site[(e.key + " " + shp, where)][0] += e.count
        site[(e.key + " " + shp, where)][1] += e.self_device_time_total
    print("\nnon-GEMM aten ops by call site (launch-count order)")
    for (op, where), (cnt, t) in sorted(site.items(), key=lambda kv: -kv[1][0])[:110]:
        print(f"  x{cnt / 3:6.1f} {t / 3e3:8.3f} ms  {op:74s} {where[:110]}")
    # the library GEMMs by shape (which dense layers still run on hipBLASLt / rocBLAS)
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key in ("aten::mm", "aten::addmm", "aten::bmm") and getattr(e, "self_device_time_total", 0) > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print(f"\nlibrary GEMMs (aten::mm / addmm / bmm): {tot / 3e3:.3f} ms per step")
    for e in rows[:60]:
        print(f"{e.self_device_time_total / 3e3:8.3f} ms  x{e.count / 3:6.1f}  {e.key:12s} {str(e.input_shapes)[:120]}")


if __name__ == "__main__":
    main()
