"""GDANet part-segmentation network used as a frozen feature extractor (interface and
state-dict keys of models/segmentation/GDANet_ptseg.py:7-131 of the reference).

forward(x [B,3,N], norm_plt [B,N,3], cls_label [B,16]) -> (embeddings [B,128,N], log-probs [B,N,C]).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import GDANet_util as U
from .GDANet_util import GDM, SGCAM, local_operator, local_operator_withnorm


class GDANet(nn.Module):
    def __init__(self, num_classes, device=None):
        super().__init__()
        self.device = device
        bn2, bn1 = (lambda c: nn.BatchNorm2d(c, momentum=0.1)), (lambda c: nn.BatchNorm1d(c, momentum=0.1))
        # the reference registers every BatchNorm twice (as `bnX` and inside `convX.1`); same here so
        # that checkpoints, which contain both key sets, load strictly
        self.bn1, self.bn11, self.bn12 = bn2(64), bn2(64), bn1(64)
        self.bn2, self.bn21, self.bn22 = bn2(64), bn2(64), bn1(64)
        self.bn3, self.bn31, self.bn32 = bn2(128), bn2(128), bn1(128)
        self.bn4, self.bnc = bn1(512), bn1(64)
        self.bn5, self.bn6, self.bn7 = bn1(256), bn1(256), bn1(128)

        self.conv1 = nn.Sequential(nn.Conv2d(9, 64, kernel_size=1, bias=True), self.bn1)
        self.conv11 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=True), self.bn11)
        self.conv12 = nn.Sequential(nn.Conv1d(64 * 2, 64, kernel_size=1, bias=True), self.bn12)
        self.conv2 = nn.Sequential(nn.Conv2d(67 * 2, 64, kernel_size=1, bias=True), self.bn2)
        self.conv21 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=True), self.bn21)
        self.conv22 = nn.Sequential(nn.Conv1d(64 * 2, 64, kernel_size=1, bias=True), self.bn22)
        self.conv3 = nn.Sequential(nn.Conv2d(131 * 2, 128, kernel_size=1, bias=True), self.bn3)
        self.conv31 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=True), self.bn31)
        self.conv32 = nn.Sequential(nn.Conv1d(128, 128, kernel_size=1, bias=True), self.bn32)
        self.conv4 = nn.Sequential(nn.Conv1d(256, 512, kernel_size=1, bias=True), self.bn4)
        self.convc = nn.Sequential(nn.Conv1d(16, 64, kernel_size=1, bias=True), self.bnc)
        self.conv5 = nn.Sequential(nn.Conv1d(256 + 512 + 64, 256, kernel_size=1, bias=True), self.bn5)
        self.dp1 = nn.Dropout(0.4)
        self.conv6 = nn.Sequential(nn.Conv1d(256, 256, kernel_size=1, bias=True), self.bn6)
        self.dp2 = nn.Dropout(0.4)
        self.conv7 = nn.Sequential(nn.Conv1d(256, 128, kernel_size=1, bias=True), self.bn7)
        self.conv8 = nn.Conv1d(128, num_classes, kernel_size=1, bias=True)
        self.SGCAM_1s, self.SGCAM_1g = SGCAM(64), SGCAM(64)
        self.SGCAM_2s, self.SGCAM_2g = SGCAM(64), SGCAM(64)

    @staticmethod
    def _first_conv(x, k, conv_a, norm_plt=None):
        """relu(conv_a(local_operator[_withnorm](x))): conv_a = Sequential(Conv2d 1x1, BatchNorm2d)."""
        if U.FUSED_LOCAL_CONV:
            extra = None if norm_plt is None else norm_plt.contiguous().view(x.size(0), -1, x.size(2))  # the raw view of :40-70
            return F.relu(conv_a[1](U.local_conv(x, conv_a[0], k, extra)))
        feat = local_operator(x, k=k) if norm_plt is None else local_operator_withnorm(x, norm_plt, k=k)
        return F.relu(conv_a(feat))

    def _edge_stage(self, x, k, conv_a, conv_b, norm_plt=None):
        """relu(conv_b(relu(conv_a(local_operator[_withnorm](x))))).max over the k neighbours: x [B,C,N] -> [B,Co,N]."""
        if (U.TOKEN_MAJOR_STAGE and U.FUSED_LOCAL_CONV and self.training and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and U._native.backend_for_testing() is None
                and conv_a[0].out_channels % 4 == 0 and conv_b[0].out_channels % 4 == 0):
            from ..AdaPoinTr import _BatchNormRowsLeakyReLU, _linear_tokens
            from ..Transformer_utils import max_over_tokens
            B, _, N = x.shape
            extra = None if norm_plt is None else norm_plt.contiguous().view(B, -1, N)
            y = U.local_conv(x, conv_a[0], k, extra, token_major=True)                    # [B,N,k,Ca]
            bn_a, bn_b, lin = conv_a[1], conv_b[1], conv_b[0]
            a = _BatchNormRowsLeakyReLU.apply(y.view(-1, y.size(-1)), bn_a.weight, bn_a.bias, bn_a, 0.0)   # slope 0: ReLU
            b = _linear_tokens(a, lin.weight.view(lin.out_channels, -1), lin.bias)
            c = _BatchNormRowsLeakyReLU.apply(b, bn_b.weight, bn_b.bias, bn_b, 0.0)
            f = max_over_tokens(c.view(B * N, k, lin.out_channels))                         # [B*N,Cb]
            return f.view(B, N, lin.out_channels).transpose(1, 2)
        return F.relu(conv_b(self._first_conv(x, k, conv_a, norm_plt))).max(dim=-1, keepdim=False)[0]

    def _stage(self, f, sharp, gentle, conv_merge):
        fs, fg = GDM(f, M=512)
        f = torch.cat([sharp(f, fs.transpose(2, 1)), gentle(f, fg.transpose(2, 1))], 1)
        return F.relu(conv_merge(f))

    def forward(self, x, norm_plt, cls_label):
        B, C, N = x.size()
        x1 = self._stage(self._edge_stage(x, 30, self.conv1, self.conv11, norm_plt),
                         self.SGCAM_1s, self.SGCAM_1g, self.conv12)
        x1t = torch.cat((x, x1), dim=1)
        x2 = self._stage(self._edge_stage(x1t, 30, self.conv2, self.conv21),
                         self.SGCAM_2s, self.SGCAM_2g, self.conv22)
        x2t = torch.cat((x1t, x2), dim=1)
        x3 = F.relu(self.conv32(self._edge_stage(x2t, 30, self.conv3, self.conv31)))
        xx = torch.cat((x1, x2, x3), dim=1)

        xc = F.adaptive_max_pool1d(F.relu(self.conv4(xx)), 1).view(B, -1)
        cls_label = F.relu(self.convc(cls_label.view(B, 16, 1)))
        cls = torch.cat((xc.view(B, 512, 1), cls_label), dim=1).repeat(1, 1, N)
        x = torch.cat((xx, cls), dim=1)
        x = self.dp1(F.relu(self.conv5(x)))
        x = self.dp2(F.relu(self.conv6(x)))
        x = F.relu(self.conv7(x))
        emb = x.clone()
        logp = F.log_softmax(self.conv8(x), dim=1).permute(0, 2, 1)
        return emb, logp
