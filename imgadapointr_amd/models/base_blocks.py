"""Encoder / decoder blocks, DGCNN grouper and rebuild heads of the AdaPoinTr path.

Interface (class names, constructor arguments, attribute names and therefore state-dict keys)
of models/base_blocks.py of the reference:
  SelfAttnBlockApi / CrossAttnBlockApi            (base_blocks.py:14-313)
  TransformerEncoder / TransformerDecoder          (base_blocks.py:316-367)
  PointTransformerEncoder / Decoder (+Entry)       (base_blocks.py:370-512)
  DGCNN_Grouper                                    (base_blocks.py:516-629)
  Fold, SimpleRebuildFCLayer, CycleLR              (base_blocks.py:702-783)
Block tokens: 'attn' and 'graph' are implemented (every AdaPoinTr-family YAML uses only those);
the deformable tokens 'rw_deform' / 'deform' / 'deform_graph' raise NotImplementedError.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import _native
from ..pointnet2_ops import pointnet2_utils
from .Transformer_utils import (Attention, CrossAttention, DynamicGraphAttention, LayerNorm, LayerScale, Linear, Mlp,
                                _SplitEdgeWeight, add_layer_norm,
                                knn_idx, linear_on_broadcast_concat, max_over_tokens)

_VALID_TOKENS = ('attn', 'rw_deform', 'deform', 'graph', 'deform_graph')
# the decoder's memory side as a parallel graph branch (TransformerDecoder.forward); PTOPS_OVERLAP_DECODER_MEMORY=0: off
import os as _os  # noqa: E402
OVERLAP_DECODER_MEMORY = _os.environ.get("PTOPS_OVERLAP_DECODER_MEMORY", "1") != "0"
_memory_streams = {}


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """timm.models.layers.trunc_normal_ (the reference's initialiser) via torch.nn.init."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm's DropPath; rate is 0 in every AdaPoinTr YAML)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


def _local_module(token, dim, k):
    if token == 'graph':
        return DynamicGraphAttention(dim, k=k)
    raise NotImplementedError(
        f"block token '{token}' (deformable attention) is outside the AdaPoinTr hot path; "
        "no reference YAML selects it")


def _init_linear_ln(m):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


def _residual_out(x, branch, next_norm):
    """x + branch, or (x + branch, next_norm(x + branch)) with the add folded into the following block's norm1."""
    if next_norm is None:
        return x + branch
    return add_layer_norm(x, branch, next_norm)


class SelfAttnBlockApi(nn.Module):
    """Pre-norm encoder block: [global attention ‖ local graph attention] -> MLP
    (base_blocks.py:14-109)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm, block_style='attn-deform',
                 combine_style='concat', k=10, n_group=2):
        super().__init__()
        self.combine_style = combine_style
        assert combine_style in ['concat', 'onebyone'], f'got unexpect combine_style {combine_style} for local and global attn'
        self.norm1 = norm_layer(dim)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

        tokens = block_style.split('-')
        assert 0 < len(tokens) <= 2, f'invalid block_style {block_style}'
        self.block_length = len(tokens)
        self.attn = None
        self.local_attn = None
        for token in tokens:
            assert token in _VALID_TOKENS, f'got unexpect block_token {token} for Block component'
            if token == 'attn':
                self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
            else:
                self.local_attn = _local_module(token, dim, k)
        if self.attn is not None and self.local_attn is not None:
            if combine_style == 'concat':
                self.merge_map = Linear(dim * 2, dim)
            else:
                self.norm3 = norm_layer(dim)
                self.ls3 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
                self.drop_path3 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, x, pos, idx=None, normed=None, next_norm=None):
        """`normed`: norm1(x) if the caller already has it; `next_norm`: the following block's norm1 — then the
        last residual add is folded into that norm and (x, next_norm(x)) is returned (see TransformerEncoder)."""
        if self.block_length == 2 and self.combine_style == 'onebyone':
            x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x) if normed is None else normed)))
            x = x + self.drop_path3(self.ls3(self.local_attn(self.norm3(x), pos, idx=idx)))
        else:
            norm_x = self.norm1(x) if normed is None else normed
            feats = []
            if self.attn is not None:
                feats.append(self.attn(norm_x))
            if self.local_attn is not None:
                feats.append(self.local_attn(norm_x, pos, idx=idx))
            if len(feats) != self.block_length:
                raise RuntimeError()
            f = self.merge_map(torch.cat(feats, dim=-1)) if len(feats) == 2 else feats[0]
            x, norm_x = add_layer_norm(x, self.drop_path1(self.ls1(f)), self.norm2)     # x = x + f; norm2(x)
            return _residual_out(x, self.drop_path2(self.ls2(self.mlp(norm_x))), next_norm)
        return _residual_out(x, self.drop_path2(self.ls2(self.mlp(self.norm2(x)))), next_norm)


class CrossAttnBlockApi(nn.Module):
    """Pre-norm decoder block: self attention (optionally masked, optionally fused with a local
    graph branch) -> cross attention (same options) -> MLP  (base_blocks.py:112-313)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm,
                 self_attn_block_style='attn-deform', self_attn_combine_style='concat',
                 cross_attn_block_style='attn-deform', cross_attn_combine_style='concat', k=10, n_group=2):
        super().__init__()
        self.norm2 = norm_layer(dim)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

        # self attention
        self.norm1 = norm_layer(dim)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.self_attn_combine_style = self_attn_combine_style
        assert self_attn_combine_style in ['concat', 'onebyone'], \
            f'got unexpect self_attn_combine_style {self_attn_combine_style} for local and global attn'
        tokens = self_attn_block_style.split('-')
        assert 0 < len(tokens) <= 2, f'invalid self_attn_block_style {self_attn_block_style}'
        self.self_attn_block_length = len(tokens)
        self.self_attn = None
        self.local_self_attn = None
        for token in tokens:
            assert token in _VALID_TOKENS, f'got unexpect self_attn_block_token {token} for Block component'
            if token == 'attn':
                self.self_attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
            else:
                self.local_self_attn = _local_module(token, dim, k)
        if self.self_attn is not None and self.local_self_attn is not None:
            if self_attn_combine_style == 'concat':
                self.self_attn_merge_map = Linear(dim * 2, dim)
            else:
                self.norm3 = norm_layer(dim)
                self.ls3 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
                self.drop_path3 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

        # cross attention
        self.norm_q = norm_layer(dim)
        self.norm_v = norm_layer(dim)
        self.ls4 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path4 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.cross_attn_combine_style = cross_attn_combine_style
        assert cross_attn_combine_style in ['concat', 'onebyone'], \
            f'got unexpect cross_attn_combine_style {cross_attn_combine_style} for local and global attn'
        tokens = cross_attn_block_style.split('-')
        assert 0 < len(tokens) <= 2, f'invalid cross_attn_block_style {cross_attn_block_style}'
        self.cross_attn_block_length = len(tokens)
        self.cross_attn = None
        self.local_cross_attn = None
        for token in tokens:
            assert token in ('attn', 'deform', 'graph', 'deform_graph'), \
                f'got unexpect cross_attn_block_token {token} for Block component'
            if token == 'attn':
                self.cross_attn = CrossAttention(dim, dim, num_heads=num_heads, qkv_bias=qkv_bias,
                                                 attn_drop=attn_drop, proj_drop=drop)
            else:
                self.local_cross_attn = _local_module(token, dim, k)
        if self.cross_attn is not None and self.local_cross_attn is not None:
            if cross_attn_combine_style == 'concat':
                self.cross_attn_merge_map = Linear(dim * 2, dim)
            else:
                self.norm_q_2 = norm_layer(dim)
                self.norm_v_2 = norm_layer(dim)
                self.ls5 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
                self.drop_path5 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    @staticmethod
    def denoise_mask(n, denoise_length, device):
        """[n,n] mask, 1 = blocked (base_blocks.py:224-229): the real queries (rows [:-dl]) may not attend
        to the denoise tokens (columns [-dl:]); denoise tokens see everything."""
        mask = torch.zeros(n, n, device=device)
        mask[:-denoise_length, -denoise_length:] = 1.
        return mask

    def memory_side(self, v, n_queries):
        """What this block computes from the memory `v` alone: norm_v(v) and, where the packed form applies, the k / v
        projections of its global cross attention -> (norm_v, kv or None).  None if the block does not take that path."""
        if self.cross_attn_block_length == 2 and self.cross_attn_combine_style == 'onebyone':
            return None
        norm_v = self.norm_v(v)
        kv = self.cross_attn.project_kv(norm_v, n_queries) if self.cross_attn is not None else None
        return norm_v, kv

    def forward(self, q, v, q_pos, v_pos, self_attn_idx=None, cross_attn_idx=None, denoise_length=None,
                denoise_mask=None, normed=None, next_norm=None, memory=None):
        """`denoise_mask`: the mask above, if the caller already built it (it is the same for every block).
        `normed` / `next_norm`: as in SelfAttnBlockApi.forward (norm1 of this / of the following block).
        `memory`: (norm_v, kv, event) of `memory_side`, computed ahead on another stream (TransformerDecoder.forward)."""
        if denoise_length is None:
            mask = mask_split = None
        else:
            n = q.size(1)
            mask = denoise_mask if denoise_mask is not None else self.denoise_mask(n, denoise_length, q.device)
            mask_split = (n - denoise_length, n - denoise_length)

        if self.self_attn_block_length == 2 and self.self_attn_combine_style == 'onebyone':
            q = q + self.drop_path1(self.ls1(self.self_attn(self.norm1(q) if normed is None else normed, mask=mask,
                                                            mask_split=mask_split)))
            q = q + self.drop_path3(self.ls3(self.local_self_attn(
                self.norm3(q), q_pos, idx=self_attn_idx, denoise_length=denoise_length)))
        else:
            norm_q = self.norm1(q) if normed is None else normed
            feats = []
            if self.self_attn is not None:
                feats.append(self.self_attn(norm_q, mask=mask, mask_split=mask_split))
            if self.local_self_attn is not None:
                feats.append(self.local_self_attn(norm_q, q_pos, idx=self_attn_idx, denoise_length=denoise_length))
            if len(feats) != self.self_attn_block_length:
                raise RuntimeError()
            f = self.self_attn_merge_map(torch.cat(feats, dim=-1)) if len(feats) == 2 else feats[0]
            if self.cross_attn_block_length == 2 and self.cross_attn_combine_style == 'onebyone':
                q = q + self.drop_path1(self.ls1(f))
            else:                                                                       # q = q + f; norm_q(q)
                q, norm_q = add_layer_norm(q, self.drop_path1(self.ls1(f)), self.norm_q)

        if self.cross_attn_block_length == 2 and self.cross_attn_combine_style == 'onebyone':
            q = q + self.drop_path4(self.ls4(self.cross_attn(self.norm_q(q), self.norm_v(v))))
            q = q + self.drop_path5(self.ls5(self.local_cross_attn(
                q=self.norm_q_2(q), v=self.norm_v_2(v), q_pos=q_pos, v_pos=v_pos, idx=cross_attn_idx)))
        else:
            if self.self_attn_block_length == 2 and self.self_attn_combine_style == 'onebyone':
                norm_q = self.norm_q(q)
            kv = None
            if memory is not None:                                # ready since before the query chain got here
                norm_v, kv, ready = memory
                cur = torch.cuda.current_stream()
                cur.wait_event(ready)
                norm_v.record_stream(cur)
                if kv is not None:
                    kv.record_stream(cur)
            else:
                norm_v = self.norm_v(v)
            feats = []
            if self.cross_attn is not None:
                feats.append(self.cross_attn(norm_q, norm_v, kv=kv))
            if self.local_cross_attn is not None:
                feats.append(self.local_cross_attn(q=norm_q, v=norm_v, q_pos=q_pos, v_pos=v_pos, idx=cross_attn_idx))
            if len(feats) != self.cross_attn_block_length:
                raise RuntimeError()
            f = self.cross_attn_merge_map(torch.cat(feats, dim=-1)) if len(feats) == 2 else feats[0]
            q, norm_q2 = add_layer_norm(q, self.drop_path4(self.ls4(f)), self.norm2)   # q = q + f; norm2(q)
            return _residual_out(q, self.drop_path2(self.ls2(self.mlp(norm_q2))), next_norm)

        return _residual_out(q, self.drop_path2(self.ls2(self.mlp(self.norm2(q)))), next_norm)


class TransformerEncoder(nn.Module):
    """Stack of SelfAttnBlockApi; the kNN graph is computed once per forward (base_blocks.py:316-334)."""

    def __init__(self, embed_dim=256, depth=4, num_heads=4, mlp_ratio=4., qkv_bias=False, init_values=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., act_layer=nn.GELU, norm_layer=LayerNorm,
                 block_style_list=['attn-deform'], combine_style='concat', k=10, n_group=2):
        super().__init__()
        self.k = k
        self.blocks = nn.ModuleList([
            SelfAttnBlockApi(
                dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, init_values=init_values,
                drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate,
                act_layer=act_layer, norm_layer=norm_layer, block_style=block_style_list[i],
                combine_style=combine_style, k=k, n_group=n_group)
            for i in range(depth)])

    def forward(self, x, pos):
        idx = knn_idx(self.k, pos, pos)
        normed = None
        for i, block in enumerate(self.blocks):   # a block's last residual add is folded into the next block's norm1
            nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
            out = block(x, pos, idx=idx, normed=normed, next_norm=nxt)
            x, normed = out if nxt is not None else (out, None)
        return x


class TransformerDecoder(nn.Module):
    """Stack of CrossAttnBlockApi (base_blocks.py:337-367)."""

    def __init__(self, embed_dim=256, depth=4, num_heads=4, mlp_ratio=4., qkv_bias=False, init_values=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., act_layer=nn.GELU, norm_layer=LayerNorm,
                 self_attn_block_style_list=['attn-deform'], self_attn_combine_style='concat',
                 cross_attn_block_style_list=['attn-deform'], cross_attn_combine_style='concat', k=10, n_group=2):
        super().__init__()
        self.k = k
        self.blocks = nn.ModuleList([
            CrossAttnBlockApi(
                dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, init_values=init_values,
                drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate,
                act_layer=act_layer, norm_layer=norm_layer,
                self_attn_block_style=self_attn_block_style_list[i], self_attn_combine_style=self_attn_combine_style,
                cross_attn_block_style=cross_attn_block_style_list[i], cross_attn_combine_style=cross_attn_combine_style,
                k=k, n_group=n_group)
            for i in range(depth)])

    def forward(self, q, v, q_pos, v_pos, denoise_length=None):
        self_attn_idx = knn_idx(self.k, q_pos, q_pos) if denoise_length is None else None
        cross_attn_idx = knn_idx(self.k, v_pos, q_pos)
        mask = None if denoise_length is None else CrossAttnBlockApi.denoise_mask(q.size(1), denoise_length, q.device)
        # Inside a HIP-graph capture the memory side of every block (norm_v + the k / v projections: they depend on the
        # encoder output only) becomes a parallel branch of the graph beside the query chain; each block joins through its
        # own event.  In eager mode the same fork costs more in cross-stream events than it gains.
        memory = [None] * len(self.blocks)
        if OVERLAP_DECODER_MEMORY and v.is_cuda and torch.is_grad_enabled() and torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream()
            side = _memory_streams.get(v.device.index)
            if side is None:
                side = _memory_streams[v.device.index] = torch.cuda.Stream(device=v.device)
            side.wait_stream(cur)
            v.record_stream(side)
            with torch.cuda.stream(side):
                for i, block in enumerate(self.blocks):
                    m = block.memory_side(v, q.size(1))
                    if m is not None:
                        ready = torch.cuda.Event()
                        ready.record(side)
                        memory[i] = (m[0], m[1], ready)
        else:
            side = None
        normed = None
        for i, block in enumerate(self.blocks):
            nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
            out = block(q, v, q_pos, v_pos, self_attn_idx=self_attn_idx, cross_attn_idx=cross_attn_idx,
                        denoise_length=denoise_length, denoise_mask=mask, normed=normed, next_norm=nxt, memory=memory[i])
            q, normed = out if nxt is not None else (out, None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)      # every block joined through its event; this closes the fork
        return q


class PointTransformerEncoder(nn.Module):
    """(base_blocks.py:370-432).  `self.norm` is constructed but never applied, as in the
    reference (base_blocks.py:418,430-432); it stays so that checkpoints keep their keys."""

    def __init__(self, embed_dim=256, depth=12, num_heads=4, mlp_ratio=4., qkv_bias=True, init_values=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=None, act_layer=None,
                 block_style_list=['attn-deform'], combine_style='concat', k=10, n_group=2):
        super().__init__()
        norm_layer = norm_layer or partial(LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.num_features = self.embed_dim = embed_dim
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        assert len(block_style_list) == depth
        self.blocks = TransformerEncoder(
            embed_dim=embed_dim, num_heads=num_heads, depth=depth, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            init_values=init_values, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=dpr,
            norm_layer=norm_layer, act_layer=act_layer, block_style_list=block_style_list,
            combine_style=combine_style, k=k, n_group=n_group)
        self.norm = norm_layer(embed_dim)
        self.apply(_init_linear_ln)

    def forward(self, x, pos):
        return self.blocks(x, pos)


class PointTransformerDecoder(nn.Module):
    """(base_blocks.py:435-501)."""

    def __init__(self, embed_dim=256, depth=12, num_heads=4, mlp_ratio=4., qkv_bias=True, init_values=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=None, act_layer=None,
                 self_attn_block_style_list=['attn-deform'], self_attn_combine_style='concat',
                 cross_attn_block_style_list=['attn-deform'], cross_attn_combine_style='concat', k=10, n_group=2):
        super().__init__()
        norm_layer = norm_layer or partial(LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.num_features = self.embed_dim = embed_dim
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        assert len(self_attn_block_style_list) == len(cross_attn_block_style_list) == depth
        self.blocks = TransformerDecoder(
            embed_dim=embed_dim, num_heads=num_heads, depth=depth, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            init_values=init_values, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=dpr,
            norm_layer=norm_layer, act_layer=act_layer,
            self_attn_block_style_list=self_attn_block_style_list, self_attn_combine_style=self_attn_combine_style,
            cross_attn_block_style_list=cross_attn_block_style_list, cross_attn_combine_style=cross_attn_combine_style,
            k=k, n_group=n_group)
        self.apply(_init_linear_ln)

    def forward(self, q, v, q_pos, v_pos, denoise_length=None):
        return self.blocks(q, v, q_pos, v_pos, denoise_length=denoise_length)


class PointTransformerEncoderEntry(PointTransformerEncoder):
    def __init__(self, config, **kwargs):
        super().__init__(**dict(config))


class PointTransformerDecoderEntry(PointTransformerDecoder):
    def __init__(self, config, **kwargs):
        super().__init__(**dict(config))


# ----------------------------------------------------------------------------------- grouper
class _EdgeGroup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_k, x_q, idx):
        ctx.save_for_backward(idx)
        ctx.Nk = x_k.size(2)
        return _native.edge_group_forward(x_k, x_q, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        gk, gq = _native.edge_group_backward(grad_out.contiguous(), idx, ctx.Nk)
        return gk, gq, None


# True: every edge-conv stage runs as the fused gather / statistics / select kernels of csrc/edgeconv.hip
# (exact algebra, no [B,C,Nq,k] tensors).  False: the reference's literal chain (gather+concat kernel, then
# Conv2d / GroupNorm / LeakyReLU / max on PyTorch-ROCm) — kept for parity tests.
FUSED_EDGE_CONV = True


class _FusedEdgeConv(torch.autograd.Function):
    """out[b,c,q] = max_j LeakyReLU(GroupNorm(P[b,c,idx[b,q,j]] + R[b,c,q]))  over the k neighbours j.

    P = W1.x_k and R = (W2-W1).x_q are the two halves of the stage's 1x1 convolution (computed by the
    caller with ordinary GEMMs, so autograd turns dP / dR into weight and feature gradients).  The
    GroupNorm statistics run over all (channel-in-group, q, j) edge values, exactly as in the reference
    where the norm precedes the max (base_blocks.py:527-545); because the affine + LeakyReLU are monotone
    per channel, the max over j is the max (gamma >= 0) or min (gamma < 0) of the raw edge values."""

    @staticmethod
    def forward(ctx, P, R, idx, gamma, beta, groups, eps, slope):
        P, R = P.contiguous(), R.contiguous()
        k = idx.shape[2]
        ysel, nsel, ysum, partial = _native.edge_select(P, R, idx, gamma.detach())
        out, mean_rstd = _native.edge_norm_forward(ysel, partial, gamma, beta, groups, k, eps, slope)
        ctx.save_for_backward(P, R, idx, gamma, beta, nsel, ysum, ysel, mean_rstd)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, go):
        P, R, idx, gamma, beta, nsel, ysum, ysel, mean_rstd = ctx.saved_tensors
        # dy_e = [e selected] S + A + Bc y_e per group (GroupNorm backward with a gradient that is non-zero on
        # the selected edges only); the kernels turn that into dP, dR and the per-(b,c) sums behind dgamma / dbeta
        dP, dR, sums = _native.edge_norm_backward(go.contiguous(), ysel, ysum, P, R, idx, nsel, gamma, beta, mean_rstd,
                                                  ctx.slope)
        dgb = sums.permute(0, 2, 1).sum(dim=0)           # [2, C]: contiguous rows for AccumulateGrad (no clone per half)
        return dP, dR, None, dgb[1], dgb[0], None, None, None


def _channel_first_linear(w, x):
    """w [C', C], x [B, C, N] -> [B, C', N] = w . x[b] as ONE strided-batched GEMM with the weight broadcast over the
    batch (batch stride 0): the output is contiguous channel-first as the edge kernels want it, and neither operand
    is copied.  (torch.matmul(w, x) folds the batch into the rows instead, which costs a transposed copy of x going
    in, one of the result coming out, and the same again in the backward pass.)"""
    return torch.bmm(w.unsqueeze(0).expand(x.size(0), -1, -1), x)


class DGCNN_Grouper(nn.Module):
    """Four edge-conv stages with two FPS down-samplings (base_blocks.py:516-629)."""

    def __init__(self, k=16):
        super().__init__()
        self.k = k
        self.input_trans = nn.Conv1d(3, 8, 1)
        self.layer1 = nn.Sequential(nn.Conv2d(16, 32, kernel_size=1, bias=False), nn.GroupNorm(4, 32),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.layer2 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=False), nn.GroupNorm(4, 64),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.layer3 = nn.Sequential(nn.Conv2d(128, 64, kernel_size=1, bias=False), nn.GroupNorm(4, 64),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.layer4 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=False), nn.GroupNorm(4, 128),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.num_features = 128

    @staticmethod
    def fps_downsample(coor, x, num_group, fps_idx=None):
        """coor [B,3,N], x [B,C,N] -> the num_group farthest points and their features.
        `fps_idx` (int32 [B,>=num_group]) lets the caller share one sampling run."""
        if isinstance(fps_idx, pointnet2_utils.FpsHandle):
            fps_idx = fps_idx.result()
        if fps_idx is None:
            xyz = coor.transpose(1, 2).contiguous()
            fps_idx = pointnet2_utils.furthest_point_sample(xyz, num_group)
        else:
            fps_idx = fps_idx[:, :num_group].contiguous()
        # two gathers: the reference's cat + gather + two slices (base_blocks.py:556-562) costs a concat going in and, in
        # the backward pass, a zero-fill + copy per slice and a split of the gathered gradient
        return pointnet2_utils.gather_operation(coor, fps_idx), pointnet2_utils.gather_operation(x, fps_idx)

    def get_graph_feature(self, coor_q, x_q, coor_k, x_k):
        """coor [B,3,N*], x [B,C,N*] -> [B,2C,Nq,k] = cat(x_k[idx] - x_q, x_q)  (base_blocks.py:565-588)."""
        with torch.no_grad():
            idx = knn_idx(self.k, coor_k.transpose(-1, -2).contiguous(), coor_q.transpose(-1, -2).contiguous())
        return _EdgeGroup.apply(x_k.contiguous(), x_q.contiguous(), idx)

    def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        """One stage: max over the k neighbours of layer(get_graph_feature(...)) -> [B,C',Nq]."""
        conv, norm, act = layer[0], layer[1], layer[2]
        fused = (FUSED_EDGE_CONV and x_k.is_cuda and _native.backend_for_testing() is None and self.k <= 32
                 and x_k.size(2) * 8 <= 64 * 1024 and conv.bias is None)
        if not fused:
            return layer(self.get_graph_feature(coor_q, x_q, coor_k, x_k)).max(dim=-1, keepdim=False)[0]
        with torch.no_grad():
            idx = knn_idx(self.k, coor_k.transpose(-1, -2).contiguous(), coor_q.transpose(-1, -2).contiguous())
        C = x_k.size(1)
        w1, w21 = _SplitEdgeWeight.apply(conv.weight.view(conv.out_channels, 2 * C))
        P = _channel_first_linear(w1, x_k)             # [B,C',Nk]   W1 . x_k
        R = _channel_first_linear(w21, x_q)            # [B,C',Nq]   (W2 - W1) . x_q
        return _FusedEdgeConv.apply(P, R, idx, norm.weight, norm.bias, norm.num_groups, norm.eps,
                                    act.negative_slope)

    def forward(self, x, num, fps_idx=None):
        """x [B,N,3], num e.g. [512, 256] -> coor [B,num[1],3], f [B,num[1],128]."""
        x = x.transpose(-1, -2).contiguous()
        coor = x
        # Conv1d(3 -> 8, kernel 1) as the matrix product it is: the weight gradient of aten's convolution backward is the
        # one non-deterministic kernel of the step on this stack (differs in the last bits from run to run); two batched
        # GEMMs are reproducible.  Same parameters, same values.
        it = self.input_trans
        if isinstance(it, nn.Conv1d) and it.kernel_size == (1,) and it.groups == 1 and it.stride == (1,):
            f = _channel_first_linear(it.weight.squeeze(-1), x)
            if it.bias is not None:
                f = f + it.bias.view(1, -1, 1)
        else:
            f = it(x)

        f = self.edge_conv(self.layer1, coor, f, coor, f)

        coor_q, f_q = self.fps_downsample(coor, f, num[0], fps_idx=fps_idx)
        # the second sampling needs the first one's coordinates only: start it now, on the side stream, under stages 2 and 3
        fps2 = None
        if coor_q.dtype == torch.float32:
            fps2 = pointnet2_utils.furthest_point_sample_async(coor_q.detach().transpose(1, 2).contiguous(), num[1])
        f = self.edge_conv(self.layer2, coor_q, f_q, coor, f)
        coor = coor_q.contiguous()

        f = self.edge_conv(self.layer3, coor, f, coor, f)

        coor_q, f_q = self.fps_downsample(coor, f, num[1], fps_idx=fps2)
        f = self.edge_conv(self.layer4, coor_q, f_q, coor, f)
        coor = coor_q

        return coor.transpose(-1, -2).contiguous(), f.transpose(-1, -2).contiguous()


# ------------------------------------------------------------------------------ rebuild heads
class Fold(nn.Module):
    """FoldingNet-style two-stage head (base_blocks.py:702-744); unused by the AdaPoinTr YAMLs
    (`decoder_type: fc`).  The seed grid is a buffer here instead of a `.cuda()` tensor."""

    def __init__(self, in_channel, step, hidden_dim=512):
        super().__init__()
        self.in_channel = in_channel
        self.step = step
        a = torch.linspace(-1., 1., steps=step, dtype=torch.float).view(1, step).expand(step, step).reshape(1, -1)
        b = torch.linspace(-1., 1., steps=step, dtype=torch.float).view(step, 1).expand(step, step).reshape(1, -1)
        self.register_buffer('folding_seed', torch.cat([a, b], dim=0), persistent=False)

        def head(cin):
            return nn.Sequential(
                nn.Conv1d(cin, hidden_dim, 1), nn.BatchNorm1d(hidden_dim), nn.ReLU(inplace=True),
                nn.Conv1d(hidden_dim, hidden_dim // 2, 1), nn.BatchNorm1d(hidden_dim // 2), nn.ReLU(inplace=True),
                nn.Conv1d(hidden_dim // 2, 3, 1))

        self.folding1 = head(in_channel + 2)
        self.folding2 = head(in_channel + 3)

    def forward(self, x):
        n = self.step * self.step
        bs = x.size(0)
        features = x.view(bs, self.in_channel, 1).expand(bs, self.in_channel, n)
        seed = self.folding_seed.view(1, 2, n).expand(bs, 2, n).to(x.device)
        fd1 = self.folding1(torch.cat([seed, features], dim=1))
        return self.folding2(torch.cat([fd1, features], dim=1))


class SimpleRebuildFCLayer(nn.Module):
    """cat(max over ALL tokens, token) -> Mlp -> step x 3 offsets (base_blocks.py:747-768)."""

    def __init__(self, input_dims, step, hidden_dim=512):
        super().__init__()
        self.input_dims = input_dims
        self.step = step
        self.layer = Mlp(self.input_dims, hidden_dim, step * 3)

    def forward(self, rec_feature):
        batch_size = rec_feature.size(0)
        g_feature = max_over_tokens(rec_feature.contiguous())
        # self.layer(cat([g_feature broadcast, rec_feature])) with the first Linear split
        mlp = self.layer
        h = linear_on_broadcast_concat(mlp.fc1, g_feature, rec_feature)
        rebuild_pc = mlp.drop(mlp.fc2(mlp.drop(mlp.act(h)))).reshape(batch_size, -1, self.step, 3)
        assert rebuild_pc.size(1) == rec_feature.size(1)
        return rebuild_pc


class CycleLR():
    """Triangular cyclic schedule with exponential decay (base_blocks.py:771-783)."""

    def __init__(self, step_size, max_lr, base_lr=0.001, gamma=0.995, last_epoch=-1):
        self.base_lr = base_lr
        self.max_lr = max_lr
        self.step_size = step_size
        self.gamma = gamma

    def get_lr(self, last_epoch):
        cycle = np.floor(1 + last_epoch / (2 * self.step_size))
        x = np.abs(last_epoch / self.step_size - 2 * cycle + 1)
        return self.base_lr + (self.max_lr - self.base_lr) * np.maximum(0, (1 - x)) * self.gamma ** last_epoch
