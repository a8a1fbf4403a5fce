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
import torch.nn.functional as F

from .. import _native
from .AdaPoinTr import PCTransformer as _PointTrunk
from .AdaPoinTr import _linear_tokens
from .image_encoders import ResNet18
from . import Transformer_utils as TU
from .Transformer_utils import LayerNorm, add_layer_norm

# True: the fusion layers' 8 x 48-wide heads run on libptops' fp32 MFMA attention (64-wide tiles) by zero-padding the
# projection WEIGHTS per head (48 -> 64 output rows; the out-projection gets matching zero input columns), so no
# activation is ever padded or sliced.  Zero dims add nothing to q.k and produce zero output dims: same function.
PADDED_HEAD_MFMA_ATTENTION = True
_PAD = 64
# True: while a HIP graph is being captured (utils/graph_runner.py::GraphedTrainStep) the image encoder runs on a side
# stream that forks at the start of the forward and joins where its tokens are first needed, i.e. as a parallel branch
# of the graph beside the grouper and the point encoder (FPS occupies 32 of 256 CUs for 0.3 ms, kNN and the edge-conv
# stages are small launches); autograd runs its backward on that stream too.  In eager mode the same fork / join costs
# more in cross-stream events than the idle CUs are worth, so it is used under capture only.
import os as _os  # noqa: E402
OVERLAP_IMAGE_BRANCH = _os.environ.get("PTOPS_OVERLAP_IMAGE_BRANCH", "1") != "0"


def _pad_head_rows(w, parts, heads, d):
    """[parts*heads*d, ...] -> [parts*heads*64, ...] with zero rows after each head's d."""
    tail = w.shape[1:]
    w = w.view(parts, heads, d, *tail)
    return F.pad(w, (0, 0) * len(tail) + (0, _PAD - d)).reshape(parts * heads * _PAD, *tail)


def multihead_attention(mha, x, ctx):
    """`mha(x, ctx, ctx)[0]` of an `nn.MultiheadAttention`, batch-first: x [B,N,E], ctx [B,M,E] -> [B,N,E]."""
    E, H = mha.embed_dim, mha.num_heads
    d = E // H
    B, N, _ = x.shape
    M = ctx.size(1)
    if not (PADDED_HEAD_MFMA_ATTENTION and TU.USE_MFMA_ATTENTION and x.is_cuda and x.dtype == torch.float32
            and _native.backend_for_testing() is None and d <= _PAD and N % 32 == 0 and M % 32 == 0
            and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None
            and not mha.add_zero_attn and (mha.dropout == 0.0 or not mha.training)):
        xs, cs = x.transpose(0, 1), ctx.transpose(0, 1)
        if mha.batch_first:
            xs, cs = x, ctx
        out = mha(xs, cs, cs, need_weights=False)[0]
        return out if mha.batch_first else out.transpose(0, 1)
    w = _pad_head_rows(mha.in_proj_weight, 3, H, d)                     # [3*H*64, E]
    b = _pad_head_rows(mha.in_proj_bias, 3, H, d)
    scale = d ** -0.5
    x2, c2 = x.reshape(B * N, E), ctx.reshape(B * M, E)
    if x is ctx:
        qkv = _linear_tokens(x2, w, b).view(B, N, 3, H, _PAD)
        o = TU._MFMASelfAttention.apply(qkv, 0, N, scale)               # [B,N,H,64]
    else:
        q = _linear_tokens(x2, w[:H * _PAD], b[:H * _PAD]).view(B, N, H, _PAD).transpose(1, 2)
        k, v = _linear_tokens(c2, w[H * _PAD:], b[H * _PAD:]).view(B, M, 2, H, _PAD).permute(2, 0, 3, 1, 4).unbind(0)
        o = TU._MFMAAttention.apply(q, k, v, 0, M, scale)
    wo = F.pad(mha.out_proj.weight.view(E, H, d), (0, _PAD - d)).reshape(E, H * _PAD)
    return _linear_tokens(o.reshape(B * N, H * _PAD), wo, mha.out_proj.bias).view(B, N, E)


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
        """x [B,N,C] tokens, ctx [B,M,C] context: five post-norm attention layers (pctransformer.py:148-172)."""
        att = lambda kind, n, q, kv: multihead_attention(getattr(self, f'{kind}_{tag}{n}'), q, kv)  # noqa: E731
        ln = lambda i, out, res: add_layer_norm(out, res, getattr(self, f'layer_norm_{tag}{i}'))[1]  # noqa: E731
        x = x.contiguous()
        ctx = ctx.contiguous()
        x = ln(1, att('cross_attn', 1, x, ctx), x)
        x = ln(2, att('self_attn', 1, x, x), x)
        skip = x
        x = ln(3, att('cross_attn', 2, x, ctx), x)
        x = ln(4, att('self_attn', 2, x, x), x)
        return ln(5, att('cross_attn', 3, x, skip), x)

    def _seg_tokens(self, points, cls_vec):
        """points [B,N,3] -> GDANet part embeddings as [B,N,384] tokens."""
        from .segmentation.normals import estimate_pointcloud_normals
        normals = estimate_pointcloud_normals(points, 30, disambiguate_directions=False)
        seg_emb, _ = self.segmentator(points.transpose(1, 2), normals, cls_vec)
        return self.get_better_seg_size(seg_emb.transpose(1, 2))

    # hooks called by the point trunk ---------------------------------------------------------
    def _image_tokens(self, img):
        return self.get_better_img_size(self.im_encoder(img))                           # [B,256,384]

    def fuse_encoder_tokens(self, x, coor, img, cls_vec):
        if self.use_img:
            ahead = self.__dict__.pop('_img_ahead', None)
            if ahead is not None:                    # computed on the side stream since the start of the forward
                img_feat, side = ahead
                torch.cuda.current_stream().wait_stream(side)
                img_feat.record_stream(torch.cuda.current_stream())      # allocated there, read here: no early reuse
            else:
                img_feat = self._image_tokens(img)
            x = self._fuse(x, img_feat, 'img')
        if self.use_seg and getattr(self.config, 'seg_before_enc', False) and cls_vec is not None:
            x = self._fuse(x, self._seg_tokens(coor, cls_vec), 'seg')
        return x

    def _side_stream_for(self, t):
        side = self.__dict__.get('_side_stream')
        if side is None or side.device != t.device:
            side = self.__dict__['_side_stream'] = torch.cuda.Stream(device=t.device)
        return side

    def begin_decoder_side_branch(self, coarse, denoise_length, cls_vec):
        """Segmentation after the decoder (`seg_before_dec`): GDANet's part embeddings depend on the query coordinates only,
        which exist before the query MLP and the eight decoder blocks.  Under graph capture they are computed on the side
        stream as a parallel branch beside the decoder (normals + GDANet are ~1 150 small launches that leave most of the
        chip idle; the decoder's GEMM / attention launches do not), and joined in `fuse_decoder_tokens`."""
        if not (OVERLAP_IMAGE_BRANCH and self.use_seg and getattr(self.config, 'seg_before_dec', False) and cls_vec is not None
                and coarse.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        side = self._side_stream_for(coarse)
        side.wait_stream(torch.cuda.current_stream())
        coarse.record_stream(side)                   # allocated on the main stream, read (and saved for backward) on the side one
        n = coarse.size(1) - denoise_length
        with torch.cuda.stream(side):
            self.__dict__['_seg_ahead'] = (self._seg_tokens(coarse[:, :n], cls_vec), side)

    def fuse_decoder_tokens(self, q, coarse, denoise_length, cls_vec):
        if not (self.use_seg and getattr(self.config, 'seg_before_dec', False) and cls_vec is not None):
            return q
        n = coarse.size(1) - denoise_length            # only the real queries are enhanced
        ahead = self.__dict__.pop('_seg_ahead', None)
        if ahead is not None:
            seg = ahead[0]
            torch.cuda.current_stream().wait_stream(ahead[1])
            seg.record_stream(torch.cuda.current_stream())               # allocated there, read here: no early reuse
        else:
            seg = self._seg_tokens(coarse[:, :n], cls_vec)
        fused = self._fuse(q[:, :n], seg, 'seg')
        return fused if denoise_length == 0 else torch.cat([fused, q[:, n:]], dim=1)

    def forward(self, xyz, img=None, cls_vec=None, denoise_noise=None):
        if (OVERLAP_IMAGE_BRANCH and self.use_img and img is not None and img.is_cuda and img.dim() == 4
                and torch.cuda.is_current_stream_capturing()):
            main = torch.cuda.current_stream()
            side = self._side_stream_for(img)
            side.wait_stream(main)                   # fork: joins the capture
            with torch.cuda.stream(side):
                self.__dict__['_img_ahead'] = (self._image_tokens(img), side)
        try:
            return super().forward(xyz, denoise_noise=denoise_noise, img=img, cls_vec=cls_vec)
        finally:
            for key in ('_img_ahead', '_seg_ahead'):
                ahead = self.__dict__.pop(key, None)
                if ahead is not None:                # the trunk never asked for the tokens: still join the capture
                    torch.cuda.current_stream().wait_stream(ahead[1])
