"""Child process of tests/test_runner_gpu.py::test_flat_ddp_call_sequence_on_rccl: one rank, backend "nccl" (= RCCL),
the exact call sequence the multi-GPU bench issues (init with device_id, parameter broadcasts, asynchronous
ReduceOp.AVG all-reduces launched from gradient hooks, handle.wait(), barrier) on a small AdaPoinTr; compared with the
unwrapped model.  Averaging over one rank is the identity and, since round 3, every scatter-add backward of the path sums in
a fixed order (Chamfer, edge-conv) or in fixed point (gather-max) instead of float atomics in arrival order: losses and
parameters of the two runs are compared EXACTLY, over three steps."""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_runner_cpu import tiny_config
    from imgadapointr_amd import _native
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.tools import builder
    from imgadapointr_amd.tools.runner import TrainStep
    from imgadapointr_amd.utils.flat_ddp import FlatDDP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29631"), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev)
    _native.load()
    cfg = tiny_config()
    torch.manual_seed(0)
    plain = build_model_from_cfg(cfg.model).to(dev).train()
    wrapped = FlatDDP(copy.deepcopy(plain), bucket_mb=0.05, single_rank_collectives=True)   # several buckets
    assert wrapped._collective and wrapped._avg_in_collective
    steps = [TrainStep(m, builder.build_optimizer(m, cfg), 10, 1) for m in (plain, wrapped)]
    g = torch.Generator(device="cpu").manual_seed(1)
    for it in range(3):
        partial = (torch.rand(2, 256, 3, generator=g) - 0.5).to(dev)
        gt = (torch.rand(2, 512, 3, generator=g) - 0.5).to(dev)
        losses = []
        for s in steps:
            torch.manual_seed(100 + it)          # the denoise jitter draws from the global generator
            losses.append([float(x) for x in s(partial, gt)])
        assert losses[0] == losses[1], (it, losses)     # same weights, same inputs, deterministic forward AND backward
    dist.barrier()
    torch.cuda.synchronize()
    assert len(wrapped.grad_buffers()) > 1
    torch.set_grad_enabled(False)
    differ = [n for (n, a), b in zip(plain.named_parameters(), wrapped.module.parameters()) if not torch.equal(a, b)]
    assert not differ, (len(differ), differ[:6])
    dist.destroy_process_group()
    print("rccl single-rank ok:", len(wrapped.grad_buffers()), "buckets")


if __name__ == "__main__":
    main()
