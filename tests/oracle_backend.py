"""Test-only backend that routes imgadapointr_amd._native's functions to the CPU oracle.

Lets the Python plumbing (autograd wrappers, model, runner) run on a machine without a GPU.
Installed explicitly by tests / the bench's cpu_baseline leg via
`_native.set_backend_for_testing(OracleBackend())`; the package never does this itself.
"""
import numpy as np
import torch

from oracle import ref


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _pt(a, like):
    return torch.from_numpy(np.ascontiguousarray(a)).to(like.device)


class OracleBackend:
    def chamfer_forward(self, xyz1, xyz2):
        return tuple(_pt(a, xyz1) for a in ref.chamfer_forward(_np(xyz1), _np(xyz2)))

    def chamfer_backward(self, xyz1, xyz2, idx1, idx2, g1, g2):
        return tuple(_pt(a, xyz1) for a in ref.chamfer_backward(
            _np(xyz1), _np(xyz2), _np(idx1), _np(idx2), _np(g1), _np(g2)))

    def furthest_point_sample(self, xyz, npoint):
        return _pt(ref.furthest_point_sample(_np(xyz), npoint), xyz)

    def gather_forward(self, feat, idx):
        return _pt(ref.gather_forward(_np(feat), _np(idx)), feat)

    def gather_backward(self, grad_out, idx, N):
        return _pt(ref.gather_backward(_np(grad_out), _np(idx), N), grad_out)

    def knn(self, k, xyz, new_xyz):
        return _pt(ref.knn(k, _np(xyz), _np(new_xyz)), xyz)

    def topk_rows(self, x, k):
        return torch.from_numpy(ref.topk_rows(_np(x), k)).to(x.device)

    def index_points_forward(self, points, idx):
        return _pt(ref.index_points_forward(_np(points), _np(idx)), points)

    def index_points_backward(self, grad_out, idx, N):
        return _pt(ref.index_points_backward(_np(grad_out), _np(idx), N), grad_out)

    def edge_group_forward(self, x_k, x_q, idx):
        return _pt(ref.edge_group_forward(_np(x_k), _np(x_q), _np(idx)), x_k)

    def edge_group_backward(self, grad_out, idx, Nk):
        return tuple(_pt(a, grad_out) for a in ref.edge_group_backward(_np(grad_out), _np(idx), Nk))

    def gather_max_forward(self, p, idx):
        return tuple(_pt(a, p) for a in ref.gather_max_forward(_np(p), _np(idx)))

    def gather_max_backward(self, grad_out, arg, N):
        return _pt(ref.gather_max_backward(_np(grad_out), _np(arg), N), grad_out)

    def emd_forward(self, xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments,
                    max_increments, unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
        d, a = ref.emd_forward(_np(xyz1), _np(xyz2), eps, iters)
        dist.copy_(_pt(d, dist))
        assignment.copy_(_pt(a, assignment))
        return 1

    def emd_backward(self, xyz1, xyz2, gradxyz, graddist, idx):
        gradxyz.add_(_pt(ref.emd_backward(_np(xyz1), _np(xyz2), _np(graddist), _np(idx)), gradxyz))
        return 1
