"""Building blocks of GDANet (interface of models/segmentation/GDANet_util.py of the reference):
feature-space kNN :5-11, local_operator :14-37, local_operator_withnorm :40-70, GDM :73-150,
SGCAM :153-212.  The kNN here is in feature space (up to 131 channels), so the distances stay a dense
matmul on PyTorch-ROCm; the clouds are the 512 coarse points, i.e. [B,512,512] matrices.
The selection is libptops' row-wise top-k (csrc/topk.hip: one wavefront per row, 0.03 ms instead of 0.11-0.17 ms per call).
"""
import os

import torch
from torch import nn

from ... import _native

# True: `pairwise.topk(k)[1]` on the GPU is libptops' `ptops_topk_rows` (same indices wherever the values differ; equal
# values in ascending column order, which torch.topk leaves open)
USE_NATIVE_TOPK = True


# True: in training mode on the GPU a local-operator stage - conv_a, BatchNorm2d, ReLU, conv_b, BatchNorm2d, ReLU, max over
# the k neighbours (GDANet_ptseg.py:59-60, 66-67, 72) - stays token-major ([B*N*k, C] rows) on libptops' row kernels:
# BatchNorm over rows + ReLU in one kernel each way (csrc/batchnorm.hip; MIOpen's NCHW BatchNorm ran 6x off the memory
# roofline on these [16, 64, 512, 30] tensors), the 1x1 conv_b as a GEMM over the rows, max over k with its arg-max
# (csrc/reduce.hip).  Same function, same running-statistics updates as nn.BatchNorm2d.
TOKEN_MAJOR_STAGE = os.environ.get("PTOPS_GDANET_TOKEN_MAJOR", "1") != "0"


def _topk_indices(pairwise, k):
    if (USE_NATIVE_TOPK and pairwise.is_cuda and pairwise.dtype == torch.float32 and pairwise.size(-1) <= 2048
            and _native.backend_for_testing() is None):
        return _native.topk_rows(pairwise.detach().contiguous(), k)
    return pairwise.topk(k=k, dim=-1)[1]


def knn(x, k):
    """x [B,C,N] -> (idx [B,N,k] of the k nearest incl. self, negative squared distances [B,N,N])."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise = -xx - inner - xx.transpose(2, 1)
    return _topk_indices(pairwise, k), pairwise


def _gather_rows(x_nc, idx):
    """x_nc [B,N,C], idx [B,N,k] -> [B,N,k,C]."""
    B, N, C = x_nc.shape
    k = idx.size(-1)
    flat = idx.reshape(B, N * k, 1).expand(-1, -1, C)
    return torch.gather(x_nc, 1, flat).view(B, N, k, C)


def local_operator(x, k, device=None):
    """[B,C,N] -> [B,2C,N,k] = cat(neighbour - centre, neighbour)."""
    idx, _ = knn(x, k=k)
    xt = x.transpose(2, 1).contiguous()
    nbr = _gather_rows(xt, idx)
    centre = xt.unsqueeze(2)
    return torch.cat((nbr - centre, nbr), dim=3).permute(0, 3, 1, 2)


def local_operator_withnorm(x, norm_plt, k, device=None):
    """x [B,3,N], norm_plt [B,N,3] (viewed as [B,3,N] like the reference) -> [B,9,N,k]."""
    B, _, N = x.shape
    norm_plt = norm_plt.contiguous().view(B, -1, N)      # the reference's raw view, not a transpose
    idx, _ = knn(x, k=k)
    xt = x.transpose(2, 1).contiguous()
    nt = norm_plt.transpose(2, 1).contiguous()
    nbr = _gather_rows(xt, idx)
    nbr_norm = _gather_rows(nt, idx)
    return torch.cat((nbr - xt.unsqueeze(2), nbr, nbr_norm), dim=3).permute(0, 3, 1, 2)


# True: the first 1x1 convolution of a local-operator stage is applied BEFORE the neighbour gather.  The stage input is
# cat(nbr - centre, nbr[, nbr_normals]) over channels, so with W = [Wd | Wn | Wm] the convolution is
#   y[:, n, j] = (Wd + Wn) x[idx[n, j]] + Wm normals[idx[n, j]] - Wd x[n] + b = P[idx[n, j]] + R[n]:
# two Conv1d over the N points (P, R) and a gather of Co channels, instead of a Conv2d over N*k edges of 2C (+3) channels
# - 15x fewer multiply-adds for k = 30, and the [B, 2C, N, k] edge tensor (257 MB for the 262-channel stage at B = 16), its
# cat / permute copies and their backward are never built.  Same function; sums are associated differently (1e-6 relative).
FUSED_LOCAL_CONV = os.environ.get("PTOPS_GDANET_FUSED_CONV", "1") != "0"


class _ContiguousGrad(torch.autograd.Function):
    """Identity whose gradient is made contiguous: the gradient of `local_conv`'s output arrives channel-major (it is the
    backward of a permute), and a scatter-add reading it token-major with a 61 KB stride between channels ran 11x slower."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def local_conv(x, conv, k, extra=None, token_major=False):
    """`conv(local_operator(x, k))` - or, with `extra` = the normals as [B,3,N] channels, `conv(local_operator_withnorm(...))`
    - for a 1x1 `nn.Conv2d`: x [B,C,N] -> [B,Co,N,k] (reference GDANet_util.py:14-70 followed by GDANet_ptseg.py:59,66,72);
    `token_major`: the same values as [B,N,k,Co] (one edge = one row of Co channels)."""
    B, C, N = x.shape
    Co = conv.out_channels
    idx, _ = knn(x, k=k)                                                 # [B,N,k] int64
    W = conv.weight.view(Co, -1)
    Wd = W[:, :C]
    src, Wp = x, Wd + W[:, C:2 * C]
    if extra is not None:
        src, Wp = torch.cat((x, extra), dim=1), torch.cat((Wp, W[:, 2 * C:]), dim=1)
    P = torch.nn.functional.conv1d(src, Wp.unsqueeze(-1))                # [B,Co,N]
    R = torch.nn.functional.conv1d(x, -Wd.unsqueeze(-1), conv.bias)      # [B,Co,N]
    # the gather runs token-major (a neighbour = one row of Co contiguous channels): its backward is a scatter-add whose
    # consecutive threads hit consecutive channels of ONE destination row (0.17 ms) instead of 30-fold contended scalar atomics
    # all over a channel's 512 points (0.74 ms for the 128-channel stage when gathered channel-major)
    y = _ContiguousGrad.apply(_gather_rows(P.transpose(2, 1).contiguous(), idx)) + R.transpose(2, 1).unsqueeze(2)   # [B,N,k,Co]
    return y if token_major else y.permute(0, 3, 1, 2).contiguous()


def GDM(x, M, device=None):
    """Geometry-disentangle module: the M sharpest and M gentlest points of x [B,C,N] as [B,M,C]."""
    k, tau, sigma = 64, 0.2, 2
    B, C, N = x.shape
    idx, p = knn(x, k=k)                                  # p = -squared distance
    mask = torch.sqrt(torch.abs(p)) < tau
    w = torch.exp(p / (sigma * sigma)) * mask.float()     # [B,N,N]
    # A = D w with D = diag(1 / column sums of w): row i of w scaled by 1/sum_r w[r,i]
    A = w / torch.sum(w, dim=1).unsqueeze(-1)
    nb = idx[:, :, 1:k]                                   # drop the first hit (the point itself)
    Aij = torch.gather(A, 2, nb)                          # [B,N,k-1]
    xt = x.transpose(2, 1).contiguous()                   # [B,N,C]
    n = (Aij.unsqueeze(-1) * _gather_rows(xt, nb)).sum(dim=2)
    pai = torch.norm(xt - n, dim=-1).pow(2)
    pais = pai.topk(k=M, dim=-1)[1]
    paig = (-pai).topk(k=M, dim=-1)[1]
    xs = torch.gather(xt, 1, pais.unsqueeze(-1).expand(-1, -1, C))
    xg = torch.gather(xt, 1, paig.unsqueeze(-1).expand(-1, -1, C))
    return xs, xg


class SGCAM(nn.Module):
    """Sharp-gentle complementary attention: non-local block of x over x_2."""

    def __init__(self, in_channels, inter_channels=None, bn_layer=True):
        super().__init__()
        self.in_channels = in_channels
        self.inter_channels = inter_channels or max(in_channels // 2, 1)
        self.g = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        if bn_layer:
            self.W = nn.Sequential(nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1),
                                   nn.BatchNorm1d(self.in_channels))
            nn.init.constant_(self.W[1].weight, 0)
            nn.init.constant_(self.W[1].bias, 0)
        else:
            self.W = nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1)
            nn.init.constant_(self.W.weight, 0)
            nn.init.constant_(self.W.bias, 0)
        self.theta = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.phi = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)

    def forward(self, x, x_2):
        g_x = self.g(x_2).permute(0, 2, 1)
        theta_x = self.theta(x).permute(0, 2, 1)
        phi_x = self.phi(x_2)
        W = torch.matmul(theta_x, phi_x)
        y = torch.matmul(W / W.size(-1), g_x).permute(0, 2, 1).contiguous()
        return self.W(y) + x
