"""Fusion trunk: AdaPoinTr's PCTransformer with optional image / segmentation cross-attention.

Interface of models/internal/pctransformer.py of the reference (attribute names = state-dict
keys): after the point encoder the 256 proxy tokens attend to 256 image tokens
(ResNet18[:layer3] -> Linear(196->384)+GELU) through five post-norm `nn.MultiheadAttention(384, 8)`
layers: cross(x<-img) -> self -> [skip] -> cross(x<-img) -> self -> cross(x<-skip)
(pctransformer.py:148-172).  The segmentation variant applies the same five-layer pattern with
GDANet part embeddings before the encoder output (pctransformer.py:175-199) or after the decoder
(pctransformer.py:237-278, 293-326).
"""
import torch
import torch.nn as nn

from .AdaPoinTr import PCTransformer as _PointTrunk
from .image_encoders import ResNet18
from .Transformer_utils import LayerNorm


class PCTransformer(_PointTrunk):
    def __init__(self, config):
        self.config = config
        super().__init__(config)          # point trunk (applies the trunc-normal init at its end)
        self.use_img = bool(getattr(config, 'use_img_features', False))
        self.use_seg = bool(getattr(config, 'use_seg_features', False))
        if self.use_img:
            self.im_encoder = ResNet18()
            self.img_dim = 384
            self.get_better_img_size = nn.Sequential(nn.Linear(196, self.img_dim), nn.GELU())
            self._init_fusion_layers('img')
        if self.use_seg:
            from .segmentation.GDANet_ptseg import GDANet
            self.segmentator = GDANet(50)
            self.img_dim = 384
            self.get_better_seg_size = nn.Sequential(nn.Linear(128, self.img_dim), nn.GELU())
            self._init_fusion_layers('seg')
        if self.use_img or self.use_seg:
            # the reference applies _init_weights once, after every sub-module exists
            # (pctransformer.py:108): Linear -> trunc_normal(.02), LayerNorm -> (1, 0)
            for name in ('get_better_img_size', 'get_better_seg_size'):
                if hasattr(self, name):
                    getattr(self, name).apply(self._init_weights)
            for m in self.modules():
                if isinstance(m, nn.MultiheadAttention):
                    self._init_weights(m.out_proj)

    def _init_fusion_layers(self, tag):
        names = [('cross_attn', 1), ('self_attn', 1), ('cross_attn', 2), ('self_attn', 2), ('cross_attn', 3)]
        for i, (kind, n) in enumerate(names, start=1):
            setattr(self, f'{kind}_{tag}{n}', nn.MultiheadAttention(self.img_dim, 8))
            setattr(self, f'layer_norm_{tag}{i}', LayerNorm(self.img_dim))

    def _fuse(self, x, ctx, tag):
        """x [N,B,C] tokens, ctx [M,B,C] context (sequence-first, as nn.MultiheadAttention expects)."""
        g = lambda name: getattr(self, f'{name}_{tag}' if not name.startswith('layer_norm') else name)  # noqa: E731
        ln = lambda i: getattr(self, f'layer_norm_{tag}{i}')  # noqa: E731
        out, _ = getattr(self, f'cross_attn_{tag}1')(x, ctx, ctx, need_weights=False)
        x = ln(1)(out + x)
        out, _ = getattr(self, f'self_attn_{tag}1')(x, x, x, need_weights=False)
        x = ln(2)(out + x)
        skip = x
        out, _ = getattr(self, f'cross_attn_{tag}2')(x, ctx, ctx, need_weights=False)
        x = ln(3)(out + x)
        out, _ = getattr(self, f'self_attn_{tag}2')(x, x, x, need_weights=False)
        x = ln(4)(out + x)
        out, _ = getattr(self, f'cross_attn_{tag}3')(x, skip, skip, need_weights=False)
        return ln(5)(out + x)

    def _seg_tokens(self, points, cls_vec):
        """points [B,N,3] -> GDANet part embeddings as [N,B,384] tokens."""
        from .segmentation.normals import estimate_pointcloud_normals
        normals = estimate_pointcloud_normals(points, 30, disambiguate_directions=False)
        seg_emb, _ = self.segmentator(points.transpose(1, 2), normals, cls_vec)
        return self.get_better_seg_size(seg_emb.transpose(1, 2).transpose(0, 1))

    # hooks called by the point trunk ---------------------------------------------------------
    def fuse_encoder_tokens(self, x, coor, img, cls_vec):
        if self.use_img:
            img_feat = self.get_better_img_size(self.im_encoder(img)).transpose(0, 1)   # [256,B,384]
            x = self._fuse(x.transpose(0, 1), img_feat, 'img').transpose(0, 1)
        if self.use_seg and getattr(self.config, 'seg_before_enc', False) and cls_vec is not None:
            x = self._fuse(x.transpose(0, 1), self._seg_tokens(coor, cls_vec), 'seg').transpose(0, 1)
        return x

    def fuse_decoder_tokens(self, q, coarse, denoise_length, cls_vec):
        if not (self.use_seg and getattr(self.config, 'seg_before_dec', False) and cls_vec is not None):
            return q
        n = coarse.size(1) - denoise_length            # only the real queries are enhanced
        seg = self._seg_tokens(coarse[:, :n], cls_vec)
        fused = self._fuse(q[:, :n].transpose(0, 1), seg, 'seg').transpose(0, 1)
        return fused if denoise_length == 0 else torch.cat([fused, q[:, n:]], dim=1)

    def forward(self, xyz, img=None, cls_vec=None, denoise_noise=None):
        return super().forward(xyz, denoise_noise=denoise_noise, img=img, cls_vec=cls_vec)
