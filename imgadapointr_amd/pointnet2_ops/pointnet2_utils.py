"""Drop-in for the two `pointnet2_ops.pointnet2_utils` entry points on the AdaPoinTr path.

The reference imports the third-party CUDA package (models/Transformer_utils.py:14,
utils/misc.py:10) and calls `furthest_point_sample` (models/base_blocks.py:550, misc.py:37) and
`gather_operation` (base_blocks.py:555, misc.py:38).  Same names, argument meaning and dtypes
here, backed by libptops (imgadapointr_amd/csrc/fps.hip, gather.hip).
"""
from torch.autograd import Function

from .. import _native


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz float32 [B,N,3] (contiguous) -> int32 [B,npoint] indices; no gradient."""
        idx = _native.furthest_point_sample(xyz.detach(), npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_out):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features [B,C,N], idx int32 [B,npoint] -> [B,C,npoint]."""
        ctx.save_for_backward(idx)
        ctx.N = features.size(2)
        return _native.gather_forward(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _native.gather_backward(grad_out.contiguous(), idx, ctx.N), None


gather_operation = GatherOperation.apply
