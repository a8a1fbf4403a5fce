"""Kernel profile of the segmentation side branch alone (normals + frozen GDANet + get_better_seg_size, forward +
backward to the query coordinates) at the bench shape: what the 5 ms the graphed seg step still exposes are made of."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model, cfg, spec, step, batch, B = bench.make_workload("seg", dev)
    base = model.base_model
    cls_vec = batch[2][1]
    torch.manual_seed(1)
    pts = (torch.rand(B, 512, 3, device=dev) - 0.5).requires_grad_(True)

    def once():
        tok = base._seg_tokens(pts, cls_vec)
        tok.square().sum().backward()

    for _ in range(5):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        once()
    b.record()
    torch.cuda.synchronize()
    print(f"seg branch forward + backward, eager: {a.elapsed_time(b) / 10:.2f} ms per call (wall, includes launch gaps)")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            once()
        torch.cuda.synchronize()
    from torch.autograd import DeviceType
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    for title, keep in (("operators (device time of the kernels they launch)", lambda e: e.device_type == DeviceType.CPU),
                        ("kernels", lambda e: e.device_type != DeviceType.CPU)):
        part = sorted((e for e in rows if keep(e)), key=lambda e: -e.self_device_time_total)
        total = sum(e.self_device_time_total for e in part) / 3
        print(f"\n{title}: {total / 1e3:.3f} ms per call, {sum(e.count for e in part) / 3:.0f} per call")
        for e in part[:32]:
            shapes = str(e.input_shapes)[:90] if e.input_shapes else ""
            print(f"{e.self_device_time_total / 3 / 1e3:8.3f} ms  x{e.count / 3:6.1f}  {e.key[:70]:70s} {shapes}")


if __name__ == "__main__":
    main()
