"""Point-cloud normals: restatement of pytorch3d.ops.points_normals.estimate_pointcloud_normals
(pytorch3d 0.6.2, third party; call sites models/internal/pctransformer.py:176,239,294).

Definition (pytorch3d 0.6.2 `get_point_covariances` + `estimate_pointcloud_normals`): for every
point take its `neighborhood_size` nearest points (itself included), the covariance of those
points around their own mean, and return the eigenvector of the smallest eigenvalue.  With
`disambiguate_directions=False` pytorch3d leaves the sign to the eigen-solver, which differs
between LAPACK and rocSOLVER; here the sign is made canonical (the component of largest magnitude
is positive) so that CPU and GPU runs agree.  PARITY UNPINNED: pytorch3d is not available here.
"""
import torch

from ..Transformer_utils import index_points, knn_idx


def estimate_pointcloud_normals(pointclouds, neighborhood_size=50, disambiguate_directions=True):
    """pointclouds [B,N,3] -> unit normals [B,N,3]."""
    if disambiguate_directions:
        raise NotImplementedError('only disambiguate_directions=False is used on the path')
    B, N, _ = pointclouds.shape
    k = min(int(neighborhood_size), N)
    pts = pointclouds.detach().contiguous()
    if k <= 64:
        idx = knn_idx(k, pts, pts)                       # libptops fused kNN
    else:
        d = torch.cdist(pts, pts)
        idx = d.topk(k, dim=-1, largest=False)[1].to(torch.int32)
    nbr = index_points(pts, idx)                         # [B,N,k,3]
    diff = nbr - nbr.mean(dim=2, keepdim=True)
    cov = (diff.unsqueeze(4) * diff.unsqueeze(3)).mean(dim=2)      # [B,N,3,3]
    # 3x3 symmetric eigenproblem on the host-independent path: batched eigh in float64 for stable
    # ordering of near-degenerate eigenvalues, smallest eigenvalue first
    _, vecs = torch.linalg.eigh(cov.double())
    normals = vecs[..., 0].to(pointclouds.dtype)
    lead = normals.gather(-1, normals.abs().argmax(dim=-1, keepdim=True))
    return normals * torch.where(lead < 0, -torch.ones_like(lead), torch.ones_like(lead))
