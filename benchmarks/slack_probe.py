"""Where does the bs = 32 train step have slack on the GPU?  A 0.5 ms single-workgroup spin kernel (torch.cuda._sleep)
is inserted at one point of the step per arm, interleaved rounds in one process: the step grows by the full 0.5 ms where
the GPU is the bottleneck and by less where the host's launch rate is (the GPU was waiting there anyway).

    python benchmarks/slack_probe.py [rounds] [steps per round]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.tools import builder, runner  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    device = torch.device("cuda", 0)
    _native.load()
    enable_tuned_gemm()
    cfg = bench.load_cfg()
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg.model).to(device).train()
    step = runner.TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    partial = (torch.rand(bench.BATCH, bench.N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(bench.BATCH, bench.N_GT, 3) - 0.5).to(device)
    # calibrate the spin: cycles for 0.5 ms
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    torch.cuda._sleep(10000000)
    e.record()
    torch.cuda.synchronize()
    cycles = int(10000000 * 0.5 / s.elapsed_time(e))
    where = {"at": None}
    net = model
    grouper = net.base_model.grouper
    encoder = net.base_model.encoder
    decoder = net.base_model.decoder
    hooks = []

    def spin(tag):
        def hook(*_):
            if where["at"] == tag:
                torch.cuda._sleep(cycles)
        return hook

    hooks.append(grouper.register_forward_pre_hook(spin("before the grouper (start of forward)")))
    hooks.append(grouper.register_forward_hook(spin("after the grouper")))
    hooks.append(encoder.register_forward_hook(spin("after the encoder")))
    hooks.append(decoder.register_forward_hook(spin("after the decoder (forward)")))
    hooks.append(decoder.register_full_backward_pre_hook(spin("before the decoder's backward")))
    hooks.append(encoder.register_full_backward_pre_hook(spin("before the encoder's backward")))
    hooks.append(grouper.register_full_backward_pre_hook(spin("before the grouper's backward")))
    step.before_update = lambda: torch.cuda._sleep(cycles) if where["at"] == "before the optimizer" else None
    arms = [None, "before the grouper (start of forward)", "after the grouper", "after the encoder", "after the decoder (forward)",
            "before the decoder's backward", "before the encoder's backward", "before the grouper's backward", "before the optimizer"]
    for _ in range(8):
        step(partial, gt)
    torch.cuda.synchronize()
    times = {a: [] for a in arms}
    for _ in range(rounds):
        for a in arms:
            where["at"] = a
            step(partial, gt)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(steps):
                step(partial, gt)
            e.record()
            torch.cuda.synchronize()
            times[a].append(s.elapsed_time(e) / steps)
    base = sorted(times[None])[rounds // 2]
    print(f"spin kernel: {cycles} cycles = 0.5 ms alone")
    for a in arms:
        v = sorted(times[a])[rounds // 2]
        print(f"{str(a or 'no spin'):44s} median {v:7.3f} ms/step   +{v - base:6.3f} ms", flush=True)


if __name__ == "__main__":
    main()
