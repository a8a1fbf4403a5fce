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
    normals = smallest_eigenvector_sym3(cov.double()).to(pointclouds.dtype)
    lead = normals.gather(-1, normals.abs().argmax(dim=-1, keepdim=True))
    return normals * torch.where(lead < 0, -torch.ones_like(lead), torch.ones_like(lead))


# True: closed-form 3x3 eigen-solve (element-wise float64 ops); False: batched torch.linalg.eigh (rocSOLVER: 5 ms for
# the 16 x 2048 covariances of one step, a generic Householder solver on 3x3 matrices)
ANALYTIC_EIGEN = True


def smallest_eigenvector_sym3(A):
    """A [...,3,3] symmetric float64 -> unit eigenvector of the smallest eigenvalue [...,3].

    Eigenvalues from the trigonometric solution of the characteristic cubic (Smith 1961); the eigenvector is the
    largest of the three cross products of rows of A - lambda I (each is orthogonal to the row space).  Matrices
    whose two smallest eigenvalues coincide to rounding have no defined normal: they get a unit vector of that
    eigen-plane (see below)."""
    if not ANALYTIC_EIGEN:
        return torch.linalg.eigh(A)[1][..., 0]
    a11, a22, a33 = A[..., 0, 0], A[..., 1, 1], A[..., 2, 2]
    a12, a13, a23 = A[..., 0, 1], A[..., 0, 2], A[..., 1, 2]
    q = (a11 + a22 + a33) / 3.0
    p1 = a12 * a12 + a13 * a13 + a23 * a23
    b11, b22, b33 = a11 - q, a22 - q, a33 - q
    p = torch.sqrt((b11 * b11 + b22 * b22 + b33 * b33 + 2.0 * p1) / 6.0)
    ps = p.clamp_min(1e-100)
    det = (b11 * (b22 * b33 - a23 * a23) - a12 * (a12 * b33 - a23 * a13) + a13 * (a12 * a23 - b22 * a13)) / (ps * ps * ps)
    phi = torch.acos((det / 2.0).clamp(-1.0, 1.0)) / 3.0
    lam_max = q + 2.0 * p * torch.cos(phi)
    lam_min = q + 2.0 * p * torch.cos(phi + 2.0943951023931953)          # + 2 pi / 3
    lam_mid = 3.0 * q - lam_max - lam_min
    def null_vector(lam):
        """Largest cross product of two rows of A - lam I (orthogonal to its row space) and its norm."""
        r0 = torch.stack([a11 - lam, a12, a13], dim=-1)
        r1 = torch.stack([a12, a22 - lam, a23], dim=-1)
        r2 = torch.stack([a13, a23, a33 - lam], dim=-1)
        cands = torch.stack([torch.linalg.cross(r0, r1), torch.linalg.cross(r0, r2), torch.linalg.cross(r1, r2)], dim=-2)
        norms = cands.norm(dim=-1)
        best = norms.argmax(dim=-1, keepdim=True)
        vec = cands.gather(-2, best.unsqueeze(-1).expand(*best.shape, 3)).squeeze(-2)
        return vec, norms.gather(-1, best)

    vec, nb = null_vector(lam_min)
    vec = vec / nb.clamp_min(1e-300)
    # (near-)repeated smallest eigenvalue: the cross products vanish and A does not determine the direction.  Any unit
    # vector orthogonal to the eigenvector of the largest eigenvalue is then an eigenvector of the repeated pair; a
    # fully isotropic matrix gets e_x.  (No data-dependent branch: the forward stays capturable in a HIP graph.)
    scale = lam_max.abs().clamp_min(1e-300)
    bad = ((lam_mid - lam_min) <= 1e-9 * scale) | (nb.squeeze(-1) <= 1e-12 * scale * scale)
    bad = bad | ~torch.isfinite(vec).all(dim=-1)
    umax, nmax = null_vector(lam_max)
    umax = umax / nmax.clamp_min(1e-300)
    ex = torch.zeros_like(umax)
    ex[..., 0] = 1.0
    ey = torch.zeros_like(umax)
    ey[..., 1] = 1.0
    axis = torch.where((umax[..., :1].abs() < 0.9), ex, ey)
    alt = torch.linalg.cross(umax, axis)
    alt_n = alt.norm(dim=-1, keepdim=True)
    iso = ((lam_max - lam_min) <= 1e-9 * scale).unsqueeze(-1) | ~torch.isfinite(alt).all(dim=-1, keepdim=True) | (alt_n <= 1e-6)
    alt = torch.where(iso, ex, alt / alt_n.clamp_min(1e-300))
    return torch.where(bad.unsqueeze(-1), alt, vec)
