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


# Farthest point sampling is a chain of `npoint` dependent steps per cloud: one workgroup per cloud, B = 32 workgroups on
# 256 CUs for ~0.5 ms.  Its result depends on coordinates only, so it can run on a second HIP stream underneath the
# kNN / edge-conv kernels that do not need it yet (PTOPS_OVERLAP_FPS=1).  Measured (profiles/round3/step_ab_v7.log,
# step_ab_v8.log + the kernel trace check of benchmarks/fps_overlap_check.py): both sampling launches DO run entirely
# underneath other kernels, and the train step does not get shorter (+0.01 ms): that part of the step is bound by the
# host's launch rate, not by the GPU, so the sampling was already free.  Off by default.
import os as _os

import torch as _torch

OVERLAP_FPS = _os.environ.get("PTOPS_OVERLAP_FPS", "0") == "1"
_side_streams = {}


class FpsHandle:
    """Result of `furthest_point_sample_async`: `result()` makes the caller's stream wait for the sampling stream (once)
    and returns the int32 [B,npoint] indices."""

    def __init__(self, idx, side=None):
        self._idx, self._side = idx, side

    def result(self):
        if self._side is not None:
            cur = _torch.cuda.current_stream(self._idx.device)
            cur.wait_stream(self._side)
            self._idx.record_stream(cur)          # allocated on the side stream, used (and later freed) on this one
            self._side = None
        return self._idx


def furthest_point_sample_async(xyz, npoint):
    """`furthest_point_sample(xyz, npoint)` started on a side stream of xyz's device (CUDA tensors; elsewhere it simply
    runs) -> FpsHandle.  The side stream first waits for the work already queued on the current stream (which produces
    xyz), and `result()` joins it again, so the pattern is also valid inside a HIP-graph capture."""
    # Inside a HIP-graph capture (utils/graph_runner.py) the fork / join is a pair of graph edges and the sampling becomes
    # a parallel branch beside the first kNN / edge-conv launches: always taken there.  In eager mode only on request.
    if not (xyz.is_cuda and _native.backend_for_testing() is None
            and (OVERLAP_FPS or _torch.cuda.is_current_stream_capturing())):
        return FpsHandle(furthest_point_sample(xyz, npoint))
    dev = xyz.device
    cur = _torch.cuda.current_stream(dev)
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = _torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with _torch.cuda.stream(side):
        idx = furthest_point_sample(xyz, npoint)
    xyz.record_stream(side)                       # allocated on `cur`, read on `side`
    return FpsHandle(idx, side)


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
