"""Process-group helpers (interface of utils/dist_utils.py:9-54 of the reference).

One process per GPU; backend 'nccl' on PyTorch-ROCm is RCCL over xGMI.  `reduce_tensors`
packs any number of scalars into ONE all-reduce (the reference issues one 4-byte all-reduce
per scalar, runner.py:218-219,386-397)."""
import os

import torch
from torch import distributed as dist


def init_dist(launcher, backend='nccl', **kwargs):
    if launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    elif backend == 'nccl':
        backend = 'gloo'
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def reduce_tensor(tensor, args):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= args.world_size
    return rt


def reduce_tensors(tensors, world_size):
    """Mean over ranks of several scalar tensors with a single collective."""
    packed = torch.stack([t.detach().float().reshape(()) for t in tensors])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    packed /= world_size
    return list(packed.unbind(0))


def gather_tensor(tensor, args):
    outs = [torch.empty_like(tensor) for _ in range(args.world_size)]
    dist.all_gather(outs, tensor)
    return torch.cat(outs, dim=0)
