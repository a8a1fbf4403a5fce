"""Approximate earth-mover distance (auction algorithm).

Interface of extensions/emd/emd_module.py of the reference: the compiled module `emd` with
`forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters)` and
`backward(xyz1, xyz2, gradxyz, graddist, idx)` (emd.cpp:6-31), `emdFunction` (emd_module.py:31-72:
the caller allocates all 12 output/scratch tensors) and `emdModule`.  Kernels: csrc/emd.hip.
Points are expected in [0,1]^3 (the bid value is 3 - distance - price, emd_cuda.cu:145-146).
Bad shapes raise (the reference prints and returns -1, which its caller ignores).
"""
import types

import torch
import torch.nn as nn
from torch.autograd import Function

from ... import _native

emd = types.SimpleNamespace(forward=_native.emd_forward, backward=_native.emd_backward)


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert xyz1.size(0) == xyz2.size(0)
        assert n % 128 == 0
        assert batchsize <= 512
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        dev = xyz1.device
        i32 = dict(device=dev, dtype=torch.int32)
        dist = torch.zeros(batchsize, n, device=dev)
        assignment = torch.full((batchsize, n), -1, **i32)
        assignment_inv = torch.full((batchsize, m), -1, **i32)
        price = torch.zeros(batchsize, m, device=dev)
        bid = torch.zeros(batchsize, n, **i32)
        bid_increments = torch.zeros(batchsize, n, device=dev)
        max_increments = torch.zeros(batchsize, m, device=dev)
        unass_idx = torch.zeros(batchsize * n, **i32)
        max_idx = torch.zeros(batchsize * m, **i32)
        unass_cnt = torch.zeros(512, **i32)
        unass_cnt_sum = torch.zeros(512, **i32)
        cnt_tmp = torch.zeros(512, **i32)
        emd.forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                    unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        emd.backward(xyz1, xyz2, gradxyz1, graddist.contiguous(), assignment)
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)
