"""The reference's Chamfer-only completion baselines on the token-major point-wise executor (models/pointwise.py):

    PCN         models/external/PCN.py:7-79          cfgs/PCN_models/PCN.yaml
    FoldingNet  models/external/FoldingNet.py:6-83   cfgs/PCN_models/FoldingNet.yaml
    TopNet      models/external/TopNet.py:27-109     cfgs/PCN_models/TopNet.yaml

Registered names, constructor keys, `forward(xyz[B,n,3]) -> (coarse, fine)`, `get_loss(ret, gt, epoch) ->
(ChamferL2(coarse, gt), ChamferL2(fine, gt))` and every state-dict key / shape are the reference's, so its
checkpoints load unchanged.  All three share the two-stage PointNet encoder (`first_conv`, `second_conv`); what
differs from the reference is how the stacks are executed, see models/pointwise.py.

Two reference defects are not reproduced: the folding grids are registered as non-persistent buffers (the reference
pins them to `.cuda()` at construction, PCN.py:48 / FoldingNet.py:49, so they ignore `.to(device)`), and TopNet reads
`encoder_feature` while its YAML provides `encoder_channel` (TopNet.py:32 vs TopNet.yaml:23) and doubles tree widths
with `np.where` on a Python list (TopNet.py:20-23, an IndexError for any num_pred > 2048) - here either key is
accepted and the widening does what that loop intends, without mutating the module-level table.
"""
import math

import torch
import torch.nn as nn

from ..extensions.chamfer_dist import ChamferDistanceL2
from .build import MODELS
from .pointwise import run_parts, run_points, run_rows
from .Transformer_utils import max_over_tokens


def _stack(widths, norm_last=False):
    """Conv1d(1x1) -> BatchNorm1d -> ReLU repeated over `widths`, the last convolution bare unless norm_last."""
    mods = []
    for i, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
        mods.append(nn.Conv1d(cin, cout, 1))
        if norm_last or i < len(widths) - 2:
            mods += [nn.BatchNorm1d(cout), nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


def _square_grid(steps, half):
    """[steps*steps, 2] lattice on [-half, half]^2; column 0 runs fastest (PCN.py:46-48)."""
    t = torch.linspace(-half, half, steps=steps, dtype=torch.float)
    return torch.stack([t.repeat(steps), t.repeat_interleave(steps)], dim=-1)


class _PointNetCompletion(nn.Module):
    """Encoder + loss shared by the three baselines."""

    def __init__(self, code_width):
        super().__init__()
        self.first_conv = _stack([3, 128, 256])
        self.second_conv = _stack([512, 512, code_width])
        self.build_loss_func()

    def build_loss_func(self):
        self.loss_func = ChamferDistanceL2()

    def get_loss(self, ret, gt, epoch=0):
        loss_coarse = self.loss_func(ret[0], gt)
        # FoldingNet and TopNet return their one output twice: same value, and summing the pair doubles the gradient
        # exactly as two evaluations would - without a second 16384^2 Chamfer pass
        loss_fine = loss_coarse if ret[1] is ret[0] else self.loss_func(ret[1], gt)
        return loss_coarse, loss_fine

    def encode(self, xyz):
        """[B,n,3] -> global code [B,code_width]: max-pool, re-inject the pooled vector per point, max-pool again."""
        feature = run_points(self.first_conv, xyz)                                # [B,n,256]
        pooled = feature.max(dim=1, keepdim=True)[0]                              # [B,1,256]
        feature = run_parts(self.second_conv, [pooled, feature])                  # cat order: pooled, per-point
        return max_over_tokens(feature.contiguous())


@MODELS.register_module()
class PCN(_PointNetCompletion):
    def __init__(self, config):
        self.number_fine = config.num_pred
        self.encoder_channel = config.encoder_channel
        self.grid_size = 4
        assert self.number_fine % self.grid_size ** 2 == 0
        self.number_coarse = self.number_fine // self.grid_size ** 2
        super().__init__(self.encoder_channel)
        self.mlp = nn.Sequential(nn.Linear(self.encoder_channel, 1024), nn.ReLU(inplace=True),
                                 nn.Linear(1024, 1024), nn.ReLU(inplace=True),
                                 nn.Linear(1024, 3 * self.number_coarse))
        self.final_conv = _stack([1024 + 3 + 2, 512, 512, 3])
        self.register_buffer('folding_seed', _square_grid(self.grid_size, 0.05), persistent=False)   # [S,2]

    def forward(self, xyz):
        B = xyz.size(0)
        code = self.encode(xyz)
        coarse = self.mlp(code).reshape(B, self.number_coarse, 3)
        # fine point (m, s) sees [code | lattice offset s | coarse point m]  (PCN.py:66-74)
        anchor = coarse.unsqueeze(2)                                              # [B,M,1,3]
        delta = run_parts(self.final_conv, [code.view(B, 1, 1, -1), self.folding_seed.view(1, 1, -1, 2), anchor])
        fine = (delta + anchor).reshape(B, self.number_fine, 3)
        return coarse.contiguous(), fine


@MODELS.register_module()
class FoldingNet(_PointNetCompletion):
    def __init__(self, config):
        self.num_pred = config.num_pred
        self.encoder_channel = config.encoder_channel
        self.grid_size = int(pow(self.num_pred, 0.5) + 0.5)
        super().__init__(self.encoder_channel)
        self.folding1 = _stack([self.encoder_channel + 2, 512, 512, 3])
        self.folding2 = _stack([self.encoder_channel + 3, 512, 512, 3])
        self.register_buffer('folding_seed', _square_grid(self.grid_size, 0.5), persistent=False)    # [N,2]

    def decoder(self, code):
        code = code.unsqueeze(1)                                                  # [B,1,C]
        fd1 = run_parts(self.folding1, [self.folding_seed.unsqueeze(0), code])    # cat order: lattice, code
        fd2 = run_parts(self.folding2, [fd1, code])
        return fd1, fd2

    def forward(self, xyz):
        fd2 = self.decoder(self.encode(xyz))[1]
        return fd2, fd2                                                           # FoldingNet.py:70


_TREE_ARCH = {2: [32, 64], 4: [4, 8, 8, 8], 6: [2, 4, 4, 4, 4, 4], 8: [2, 2, 2, 2, 2, 4, 4, 4]}


def get_arch(nlevels, npts):
    """Children per tree level for `npts` leaves: the 2048-leaf table with its last narrowest level doubled once per
    factor of two (TopNet.py:17-24, with the list/array comparison fixed)."""
    doublings = int(math.log2(npts / 2048))
    assert 2048 * 2 ** doublings == npts, f'Number of points is {npts}, expected 2048x(2^n)'
    arch = list(_TREE_ARCH[nlevels])
    for _ in range(doublings):
        narrowest = min(arch)
        arch[max(i for i, a in enumerate(arch) if a == narrowest)] *= 2
    return arch


@MODELS.register_module()
class TopNet(_PointNetCompletion):
    def __init__(self, config):
        self.node_feature = config.node_feature
        self.encoder_feature = config.get('encoder_feature', None) or config.encoder_channel
        self.nlevels = config.nlevels
        self.num_pred = config.num_pred
        self.tarch = get_arch(self.nlevels, self.num_pred)
        self.Top_in_channel = self.encoder_feature + self.node_feature
        self.Top_out_channel = self.node_feature
        super().__init__(self.encoder_feature)
        self.root_layer = nn.Sequential(
            nn.Linear(self.encoder_feature, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
            nn.Linear(256, 64), nn.BatchNorm1d(64), nn.ReLU(inplace=True),
            nn.Linear(64, self.node_feature * self.tarch[0]), nn.Tanh())
        self.leaf_layer = self.get_tree_layer(self.Top_in_channel, 3, self.tarch[-1])
        self.feature_layers = nn.ModuleList(
            self.get_tree_layer(self.Top_in_channel, self.Top_out_channel, self.tarch[d])
            for d in range(1, self.nlevels - 1))

    @staticmethod
    def get_tree_layer(in_channel, out_channel, node):
        return _stack([in_channel, in_channel // 2, in_channel // 4, in_channel // 8, out_channel * node])

    def forward(self, xyz):
        B = xyz.size(0)
        code = self.encode(xyz)
        # nodes are token-major [B, L, F]; the reference holds them as [B, F, L]
        level = run_rows(self.root_layer, code).view(B, self.node_feature, self.tarch[0]).transpose(1, 2)
        code = code.unsqueeze(1)
        for i in range(1, self.nlevels):
            leaf = i == self.nlevels - 1
            stack = self.leaf_layer if leaf else self.feature_layers[i - 1]
            width, node = (3, self.tarch[-1]) if leaf else (self.node_feature, self.tarch[i])
            y = run_parts(stack, [code, level])                                   # [B, L, width*node]
            # channel c = f*node + j of parent l is feature f of child j*L + l  (`reshape(bs, F, -1)`, TopNet.py:104-106)
            L = level.size(1)
            level = torch.tanh(y.view(B, L, width, node).permute(0, 3, 1, 2).reshape(B, node * L, width))
        out = level.contiguous()
        return out, out
