"""Helpers on the AdaPoinTr path (interface of utils/misc.py of the reference).

`fps` (misc.py:32-40), `jitter_points` (misc.py:12-19), the LambdaLR / BN-momentum schedules
(misc.py:46-62,119-149) and `set_random_seed` (misc.py:64-84).  ShapeNet/KITTI-only helpers
(seprate_point_cloud, random_dropping, visualisation) are out of scope.
"""
import random

import numpy as np
import torch
import torch.nn as nn

from ..pointnet2_ops import pointnet2_utils


def jitter_points(pc, std=0.01, clip=0.05):
    """In-place clipped Gaussian jitter, one RNG draw per sample (misc.py:12-19)."""
    for i in range(pc.size(0)):
        noise = torch.empty(pc.size(1), 3, dtype=pc.dtype, device=pc.device).normal_(
            mean=0.0, std=std).clamp_(-clip, clip)
        pc[i, :, 0:3] += noise
    return pc


def fps(data, number):
    """data [B,N,3] -> farthest-point subset [B,number,3] (misc.py:32-40)."""
    idx = pointnet2_utils.furthest_point_sample(data, number)
    return pointnet2_utils.gather_operation(
        data.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()


def fps_multi(data, numbers):
    """Several `fps(data, n)` results from ONE sampling run.

    FPS from index 0 is prefix-consistent (the first p picks of a longer run are the picks of
    the run with npoint = p), so the subsets for all requested sizes are prefixes of the
    largest one.  Returns them in the order requested."""
    longest = max(numbers)
    full = fps(data, longest)
    return [full[:, :n].contiguous() for n in numbers]


def build_lambda_sche(opti, config, last_epoch=-1):
    if config.get('decay_step') is None:
        raise NotImplementedError()
    warm = getattr(config, 'warmingup_e', 0)

    def lr_lbmd(e):
        if e >= warm:
            return max(config.lr_decay ** ((e - warm) / config.decay_step), config.lowest_decay)
        return max(e / warm, 0.001)

    return torch.optim.lr_scheduler.LambdaLR(opti, lr_lbmd, last_epoch=last_epoch)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError(f"Class '{type(model).__name__}' is not a PyTorch nn Module")
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))

    def get_momentum(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        return self.lmbd(epoch)


def build_lambda_bnsche(model, config, last_epoch=-1):
    if config.get('decay_step') is None:
        raise NotImplementedError()

    def bnm_lmbd(e):
        return max(config.bn_momentum * config.bn_decay ** (e / config.decay_step), config.lowest_decay)

    return BNMomentumScheduler(model, bnm_lmbd, last_epoch=last_epoch)


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)
