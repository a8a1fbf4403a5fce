"""Chamfer distance operator and its reductions.

Interface of extensions/chamfer_dist/__init__.py of the reference: the compiled module
`chamfer` with `forward(xyz1, xyz2) -> [dist1, dist2, idx1, idx2]` and
`backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2) -> [grad_xyz1, grad_xyz2]`
(chamfer_cuda.cpp:12-39), the autograd wrapper `ChamferFunction` (__init__.py:13-25) and the
reductions ChamferDistanceL2 / L2_split / L1 / L1_PM (__init__.py:28-103), including the
`ignore_zeros` rule (rows whose coordinate SUM is zero are dropped when the batch is 1).
The kernels are imgadapointr_amd/csrc/chamfer.hip.
"""
import os
import types

import torch

from ... import _native

# The reference does `import chamfer` (a pybind11 extension).  Same two functions here.
chamfer = types.SimpleNamespace(
    forward=lambda xyz1, xyz2: list(_native.chamfer_forward(xyz1, xyz2)),
    backward=lambda xyz1, xyz2, idx1, idx2, g1, g2: list(
        _native.chamfer_backward(xyz1, xyz2, idx1, idx2, g1, g2)),
)


class ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        dist1, dist2, idx1, idx2 = chamfer.forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        grad_xyz1, grad_xyz2 = chamfer.backward(xyz1, xyz2, idx1, idx2,
                                                grad_dist1.contiguous(), grad_dist2.contiguous())
        return grad_xyz1, grad_xyz2


# True: distance search + reduction as one autograd node with one reduction launch each way (csrc/chamfer.hip,
# `ptops_chamfer_reduce`), instead of the six (forward) + seven (backward) element-wise launches of the literal
# torch.mean / torch.sqrt chain below.  Same values: the backward uses autograd's formulas in autograd's order, the
# forward sums in another (fixed) order than torch.mean.
FUSED_REDUCTIONS = os.environ.get("PTOPS_FUSED_CHAMFER_REDUCTIONS", "1") != "0"

_SQRT = {"L2": False, "L2_split": False, "L1": True, "L1_PM": True}
_WHICH = {"L2": 2, "L2_split": 0, "L1": 3, "L1_PM": 0}


class _ChamferReducedFunction(torch.autograd.Function):
    """ChamferFunction followed by one of the four reductions, `kind` in L2 / L2_split / L1 / L1_PM."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, kind):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        dist1, dist2, idx1, idx2 = chamfer.forward(xyz1, xyz2)
        out = _native.chamfer_reduce(dist1, dist2, _SQRT[kind])
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2, dist1, dist2)
        ctx.kind = kind
        ctx.set_materialize_grads(False)
        if kind == "L2_split":
            return out[0], out[1]
        return out[_WHICH[kind]] if kind != "L1_PM" else out[0]

    @staticmethod
    def backward(ctx, go1, go2=None):
        xyz1, xyz2, idx1, idx2, dist1, dist2 = ctx.saved_tensors
        if go1 is None and go2 is None:
            return None, None, None
        g1, g2 = _native.chamfer_reduce_backward(dist1, dist2, _SQRT[ctx.kind], _WHICH[ctx.kind],
                                                 go1.contiguous() if go1 is not None else None,
                                                 go2.contiguous() if go2 is not None else None)
        grad_xyz1, grad_xyz2 = chamfer.backward(xyz1, xyz2, idx1, idx2, g1, g2)
        return grad_xyz1, grad_xyz2, None


def _drop_zero_rows(xyz1, xyz2):
    nz1 = torch.sum(xyz1, dim=2).ne(0)
    nz2 = torch.sum(xyz2, dim=2).ne(0)
    return xyz1[nz1].unsqueeze(dim=0), xyz2[nz2].unsqueeze(dim=0)


class _ChamferBase(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def _dists(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _drop_zero_rows(xyz1, xyz2)
        return ChamferFunction.apply(xyz1, xyz2)

    def _reduced(self, xyz1, xyz2, kind):
        """The fused node, or None where the literal chain runs (CPU tensors, the tests' oracle backend)."""
        if not (FUSED_REDUCTIONS and xyz1.is_cuda and xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32
                and _native.backend_for_testing() is None):
            return None
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _drop_zero_rows(xyz1, xyz2)
        return _ChamferReducedFunction.apply(xyz1, xyz2, kind)


class ChamferDistanceL2(_ChamferBase):
    """mean(dist1) + mean(dist2)  (__init__.py:28-44)."""

    def forward(self, xyz1, xyz2):
        fused = self._reduced(xyz1, xyz2, "L2")
        if fused is not None:
            return fused
        dist1, dist2 = self._dists(xyz1, xyz2)
        return torch.mean(dist1) + torch.mean(dist2)


class ChamferDistanceL2_split(_ChamferBase):
    """(mean(dist1), mean(dist2))  (__init__.py:46-62)."""

    def forward(self, xyz1, xyz2):
        fused = self._reduced(xyz1, xyz2, "L2_split")
        if fused is not None:
            return fused
        dist1, dist2 = self._dists(xyz1, xyz2)
        return torch.mean(dist1), torch.mean(dist2)


class ChamferDistanceL1(_ChamferBase):
    """(mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2  (__init__.py:64-84)."""

    def forward(self, xyz1, xyz2):
        fused = self._reduced(xyz1, xyz2, "L1")
        if fused is not None:
            return fused
        dist1, dist2 = self._dists(xyz1, xyz2)
        return (torch.mean(torch.sqrt(dist1)) + torch.mean(torch.sqrt(dist2))) / 2


class ChamferDistanceL1_PM(_ChamferBase):
    """mean(sqrt(dist1)) only  (__init__.py:86-103)."""

    def forward(self, xyz1, xyz2):
        fused = self._reduced(xyz1, xyz2, "L1_PM")
        if fused is not None:
            return fused
        dist1, _ = self._dists(xyz1, xyz2)
        return torch.mean(torch.sqrt(dist1))
