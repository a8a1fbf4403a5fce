"""Functional boundary + attention primitives of the AdaPoinTr path.

Interface of models/Transformer_utils.py of the reference:
  knn_point / square_distance / index_points            (Transformer_utils.py:19-69)
  Mlp, Attention, CrossAttention                        (Transformer_utils.py:71-159)
  DynamicGraphAttention, LayerScale                     (Transformer_utils.py:779-869)
The deformable-attention variants (Transformer_utils.py:161-777, 871-1044) are reachable only
through block tokens no YAML selects and are out of scope.

kNN, the gathers, head-dim-64 attention (forward and backward), LayerNorm (+ fused residual add), GELU backward
and the bias-gradient column sums run on libptops (csrc/knn.hip, gather.hip, attention.hip, layernorm.hip,
activation.hip, reduce.hip); the dense GEMMs stay on PyTorch-ROCm (hipBLASLt / rocBLAS MFMA kernels).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native

# DynamicGraphAttention evaluation form.  True: the exact algebraic split
#   W.cat(v_j - q, q) = W1.v_j + (W2 - W1).q,  max_j lrelu(.) = lrelu(max_j .)
# (two [B*N,C]x[C,C] GEMMs + one gather-max kernel).  False: the reference's literal form
# (gather [B,S,k,C], concat to 2C, one [B*S*k,2C]x[2C,C] GEMM, max over k) — kept for parity tests.
FUSED_GRAPH_ATTENTION = True
# Attention evaluation: F.scaled_dot_product_attention (True) or the reference's explicit
# softmax(q k^T) v chain (False).
USE_SDPA = True
# Head-dim-64 attention on libptops' fp32 MFMA kernels (csrc/attention.hip); the backward variant is chosen below.
# Falls back to SDPA for shapes the kernels do not cover (token counts that are not multiples of 32).
USE_MFMA_ATTENTION = True
# "split": the dq + dk/dv kernel pair (7 GEMMs; on the bf16 matrix pipe in three-piece fp32-accurate arithmetic from
# 128 tokens up: 352 / 176 / 117 us at the bs = 32 decoder-self / cross / encoder shapes);
# "fused": the single-pass f32-MFMA backward (5 GEMMs; 430 / 206 / 114 us);
# False: PyTorch-ROCm's aten::_efficient_attention_backward, which consumes the same (out, logsumexp) pair
def _parse_attn_bwd(value):
    """PTOPS_ATTN_BWD -> "split" | "fused" | False; anything else is an error (a typo must not pick a kernel silently, and
    '0' / 'false' must mean the aten backward, not "truthy string")."""
    v = str(value).strip().lower()
    if v in ("split", "fused"):
        return v
    if v in ("0", "false", "off", "aten", "none", ""):
        return False
    raise ValueError(f"PTOPS_ATTN_BWD={value!r}: expected split, fused or 0")


USE_MFMA_ATTENTION_BWD = _parse_attn_bwd(os.environ.get("PTOPS_ATTN_BWD", "split"))       # A/B switch
_LN2 = 0.6931471805599453
_bias_cache = {}
_zero_cache = {}


def _zero_i64(device):
    z = _zero_cache.get(device)
    if z is None:
        z = _zero_cache[device] = torch.zeros((), dtype=torch.int64, device=device)
    return z


# Linear layers whose input is cat([per-cloud vector broadcast over tokens, per-token features]):
# True evaluates W.[g; x] as W_g.g (once per cloud) + W_x.x (per token) — exact in real arithmetic,
# removes the broadcast/concat tensor and the K=1024 part of a [B*M, 1027..1411] GEMM.
SPLIT_BROADCAST_LINEAR = True


class _SplitColumns(torch.autograd.Function):
    """w [O, I] -> (w[:, :c], w[:, c:]) as views; the backward is ONE concat of the two gradients.  (Autograd's own
    slicing zero-fills and copies a full-size gradient per half and adds the two: five launches.)"""

    @staticmethod
    def forward(ctx, w, c):
        return w[:, :c], w[:, c:]

    @staticmethod
    def backward(ctx, g1, g2):
        return torch.cat([g1, g2], dim=1), None


class _SplitEdgeWeight(torch.autograd.Function):
    """w [C', 2C] = [W1 | W2] of an edge layer -> (W1, W2 - W1): W . cat(x_k[idx] - x_q, x_q) = W1 x_k[idx] +
    (W2 - W1) x_q.  One subtraction forward; backward gw = [g1 - g2 | g2] as one subtraction + one concat."""

    @staticmethod
    def forward(ctx, w):
        C = w.size(1) // 2
        return w[:, :C], w[:, C:] - w[:, :C]

    @staticmethod
    def backward(ctx, g1, g2):
        return torch.cat([g1 - g2, g2], dim=1)


def linear_on_broadcast_concat(lin, g, x):
    """lin(cat([g.unsqueeze(1).expand(-1, M, -1), x], -1)) for g [B,G], x [B,M,X], lin: Linear(G+X, O)."""
    if not SPLIT_BROADCAST_LINEAR:
        return lin(torch.cat([g.unsqueeze(1).expand(-1, x.size(1), -1), x], dim=-1))
    G = g.size(-1)
    if torch.is_grad_enabled():                                 # plain autograd path: see _native.deferred_folds
        _native.fold_mark_plain(("weight", lin.weight.data_ptr()), lin.bias.data_ptr() if lin.bias is not None else None)
    w_g, w_x = _SplitColumns.apply(lin.weight, G)
    per_cloud = F.linear(g, w_g, lin.bias)                      # [B,O]
    return F.linear(x, w_x) + per_cloud.unsqueeze(1)


# Dense layers on libptops' fp32-on-bf16-matrix-pipe GEMM (csrc/gemm_split3.hip: operands split into three bf16 pieces,
# six exact piece products accumulated in fp32 — as accurate as an fp32 FMA chain, see tests) where it beats the fp32
# MFMA library kernels: large row counts (benchmarks/gemm_split3_probe.py, profiles/round2/gemm_split3_probe_*.log).
# PTOPS_SPLIT3_GEMM=0 keeps every GEMM on hipBLASLt / rocBLAS.
USE_SPLIT3_GEMM = os.environ.get("PTOPS_SPLIT3_GEMM", "1") != "0"


def _split3_ok(a, k):
    return (USE_SPLIT3_GEMM and a.is_cuda and a.dtype == torch.float32 and k % 16 == 0 and a.dim() == 2
            and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
            and _native.backend_for_testing() is None)


def _use_split3_forward(x2, weight):
    """Where x.W^T on the split-bf16 kernel beats the library's fp32 MFMA kernels (benchmarks/gemm_split3_probe.py;
    profiles/round4/gemm_split3_probe_bs16_v1.log for the rows of the batch-16 variants): every width from 8192 rows up
    (1.03-1.7x at 8192 / 9216 rows, 1.1-1.4x at 18432), wide outputs from 4096 rows up (1.2x at N = 1024; narrower ones
    tie or lose there: 0.64x at N = 384, K = 768)."""
    M, K = x2.shape
    N = weight.size(0)
    return (_split3_ok(x2, K) and _split3_ok(weight, K) and N >= 128 and (M >= 8192 or (M >= 4096 and N >= 1024)))


def _use_split3_dgrad(g2, weight):
    M, N = g2.shape                     # contraction over N (out_features), output width K (in_features)
    K = weight.size(1)
    # the k-strided operand form loses to the library below 16384 rows except on the 1024-wide squares (1.1-1.2x at 8192 /
    # 9216 rows; 0.5-0.97x elsewhere: profiles/round4/gemm_split3_probe_bs16_v1.log)
    return (_split3_ok(g2, N) and weight.is_cuda and weight.dtype == torch.float32 and weight.stride(1) == 1
            and K >= 128 and ((M >= 16384 and (N >= 512 or K >= 1024)) or (M >= 8192 and N >= 1024 and K >= 1024)))


USE_SPLIT3_WGRAD = os.environ.get("PTOPS_SPLIT3_WGRAD", "1") != "0"


def _use_split3_wgrad(g2, x2):
    """dy^T.x on the split-bf16 kernel with the contraction split over workgroups: ahead of the library's split-K fp32
    kernels (and it delivers the bias gradient in the same pass) for long contractions and outputs of >= 48 tiles'
    worth of work (benchmarks/gemm_split3_probe.py: 18432 x 1152 x 384 154 vs 155 + 15 us, 18432 x 1024 x 1024 262 vs
    314, 65536 x 512 x 512 251 vs 306; 18432 x 384 x 384 72 vs 56 + 8 and everything at 8192 rows stay on the library)."""
    M, N = g2.shape
    K = x2.size(1)
    return (USE_SPLIT3_WGRAD and _split3_ok(g2, 16) and _split3_ok(x2, 16) and M >= 16384 and M % 16 == 0 and K % 4 == 0
            and N * K >= 393216 and min(N, K) >= 128)


def linear_forward(x, weight, bias):
    """F.linear(x, weight, bias), on the split-bf16 GEMM where that is the faster fp32-accurate kernel."""
    K = x.size(-1)
    if x.is_cuda and x.dim() >= 2 and x.is_contiguous():
        x2 = x.view(-1, K)
        if _use_split3_forward(x2, weight):
            return _native.gemm_split3(x2, weight, bias).view(*x.shape[:-1], weight.size(0))
    return F.linear(x, weight, bias)


class _LinearFn(torch.autograd.Function):
    """F.linear whose bias gradient is libptops' streaming column sum (csrc/reduce.hip) instead of the
    generic 2-5 workgroup reduction autograd would run; forward, input-gradient and weight-gradient GEMMs go to the
    split-bf16 kernels where they win (the weight-gradient kernel also returns the bias gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        _note_parameter_keys(ctx, weight, bias)
        return linear_forward(x, weight, bias)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        return _linear_backward(ctx, x, weight, grad, _take_column_sums(grad) if ctx.needs_input_grad[2] else None)


def _note_parameter_keys(ctx, weight, bias):
    """Identity of the PARAMETERS behind a layer's weight / bias for `_native.fold_returning` (a queued gradient must be
    the first and only one its parameter gets before the flush).  A leaf (or a reshaped view of a whole leaf) is its own
    address.  A derived tensor (the concatenated k/v weights of CrossAttention) names its leaves in `_ptops_keys`; one
    that does not is unknown: its address may be recycled within the pass, so its gradients are never queued."""
    def keys(t, tag):
        if t is None:
            return ()
        named = getattr(t, "_ptops_keys", None)
        if named is not None:
            return tuple((tag, k) if tag else k for k in named)
        if not t.is_leaf and t._is_view() and t._base is not None and t._base.is_leaf and t.numel() == t._base.numel() \
                and t.is_contiguous() and t._base.is_contiguous():
            t = t._base          # a reshaped view of a whole leaf (a 1x1 convolution's weight as a matrix): its gradient
                                 # travels back as a view, unread
        if t.is_leaf:
            return ((tag, t.data_ptr()) if tag else t.data_ptr(),)
        return None
    ctx.weight_keys = keys(weight, "weight")
    ctx.bias_keys = keys(bias, None)


def _linear_backward(ctx, x, weight, grad, gb):
    """(dx, dW, db) of y = x W^T + b for the gradient `grad` of y; `gb`: column sums of `grad` if a producer kernel
    already has them."""
    g2 = grad.reshape(-1, grad.size(-1))
    gx = gw = None
    if ctx.needs_input_grad[0]:
        if g2.is_contiguous() and _use_split3_dgrad(g2, weight):
            gx = _native.gemm_split3(g2, weight, None, True, False).view(x.shape)
        else:
            gx = (g2 @ weight).view(x.shape)
    if ctx.needs_input_grad[1]:
        x2 = x.reshape(-1, x.size(-1))
        if USE_SPLIT3_WGRAD and ctx.weight_keys is not None and _native.wgrad_deferrable(g2, x2):
            # nothing reads a weight gradient before the end of backward: queue it for the grouped launch
            # (_native.wgrad_defer: all layers in two launches, tiles of a layer on one XCD, bias gradient included)
            want_db = (ctx.needs_input_grad[2] and gb is None and g2.size(1) % 4 == 0 and ctx.bias_keys is not None)
            gw, db = _native.wgrad_defer(g2, x2, want_db)
            gb = db if db is not None else gb
            _native.fold_returning(*ctx.weight_keys)
        elif _use_split3_wgrad(g2, x2):
            # the bias gradient (column sums of g2) falls out of the same pass
            gw, db = _native.gemm_split3_wgrad(g2, x2, ctx.needs_input_grad[2] and gb is None and g2.size(1) % 4 == 0)
            gb = db if db is not None else gb
        else:
            gw = g2.t() @ x2
    if ctx.needs_input_grad[2] and gb is None:
        gb = _native.colsum(g2.contiguous())
    if not ctx.needs_input_grad[2]:
        gb = None
    if gb is not None:
        if ctx.bias_keys is None:
            _native.fold_flush()                   # a bias of unknown provenance: nothing queued may reach it
        else:
            _native.fold_returning(*ctx.bias_keys)  # gb may be a queued fold (include/ptops.h, deferred folds)
    return gx, gw, gb


FUSED_LINEAR_GELU = os.environ.get("PTOPS_FUSED_LINEAR_GELU", "1") != "0"


class _LinearGeluFn(torch.autograd.Function):
    """gelu(x W^T + b) for the first layer of an Mlp when that GEMM runs on the split-bf16 kernel: the GELU comes out of
    the GEMM's epilogue next to the pre-activation (one launch and one read of the pre-activation less), the backward is
    the fused GELU-backward + column-sum kernel followed by the Linear backward."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        u, g = _native.gemm_split3_gelu(x.view(-1, x.size(-1)), weight, bias)
        ctx.save_for_backward(x, weight, u)
        _note_parameter_keys(ctx, weight, bias)
        return g.view(*x.shape[:-1], weight.size(0))

    @staticmethod
    def backward(ctx, gy):
        x, weight, u = ctx.saved_tensors
        N = u.size(1)
        dpre, cs = _native.gelu_backward_colsum(u, gy.contiguous().view(-1, N))
        return _linear_backward(ctx, x, weight, dpre.view(*x.shape[:-1], N), cs if ctx.needs_input_grad[2] else None)


def linear_gelu(lin, act, x):
    """act(lin(x)) for an nn.Linear-like `lin` and an nn.GELU() `act`."""
    if (FUSED_LINEAR_GELU and FUSED_GELU_BACKWARD and isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"
            and FAST_BIAS_GRAD and x.is_cuda and x.dim() >= 2 and x.is_contiguous() and torch.is_grad_enabled()
            and (x.requires_grad or lin.weight.requires_grad) and lin.bias is not None and lin.weight.size(0) % 4 == 0
            and _use_split3_forward(x.view(-1, x.size(-1)), lin.weight)):
        return _LinearGeluFn.apply(x, lin.weight, lin.bias)
    return gelu(lin(x), act)


FAST_BIAS_GRAD = True
# Column sums a producer kernel already has for a gradient tensor (the LayerNorm backward sums the dx it writes):
# keyed by (data_ptr, shape) of that tensor and consumed by the backward of the Linear that receives it as grad.
_column_sums = {}


def _layout_key(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.storage_offset())


def _offer_column_sums(t, sums):
    # holding `t` keeps the address from being reused; its version counter detects an in-place write after the offer
    _column_sums[_layout_key(t)] = (t, sums, t._version)


def _take_column_sums(t):
    """Column sums offered for exactly this tensor: same storage address, shape, strides and offset, not written to
    since the offer.  Anything else (e.g. a transposed view of a square gradient: same address and shape, other
    strides) returns None and the caller runs its own column-sum kernel."""
    hit = _column_sums.pop(_layout_key(t), None)
    if hit is None:
        return None
    src, sums, version = hit
    if src.dtype != t.dtype or src._version != version:      # views share the version counter of their base
        return None
    return sums


def _clear_column_sums():
    _column_sums.clear()


class Linear(nn.Linear):
    """nn.Linear (same parameters / state-dict keys) with the bias-gradient reduction on libptops."""

    def forward(self, x):
        if (FAST_BIAS_GRAD and self.bias is not None and x.is_cuda and x.dtype == torch.float32
                and self.out_features % 4 == 0 and torch.is_grad_enabled() and x.dim() >= 2
                and x.numel() // x.size(-1) >= 2048 and _native.backend_for_testing() is None):
            return _LinearFn.apply(x, self.weight, self.bias)
        if not torch.is_grad_enabled():
            return linear_forward(x, self.weight, self.bias)      # validation / inference: same kernels as training
        # plain autograd path: see _native.deferred_folds
        _native.fold_mark_plain(("weight", self.weight.data_ptr()), self.bias.data_ptr() if self.bias is not None else None)
        return F.linear(x, self.weight, self.bias)


FAST_LAYER_NORM = True


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x2 = x.contiguous().view(-1, x.size(-1))
        y, mean, rstd = _native.layer_norm_forward(x2, weight, bias, eps)
        ctx.save_for_backward(x2, weight, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, weight, mean, rstd = ctx.saved_tensors
        C = x2.size(1)
        dx, dgb = _native.layer_norm_backward(gy.contiguous().view(-1, C), x2, mean, rstd, weight)
        _native.fold_returning(weight.data_ptr())
        return dx.view(gy.shape), dgb[:C], dgb[C:2 * C], None


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) on the library's streaming kernels: one pass forward,
    one pass backward (dx and the dgamma / dbeta partials together)."""

    def forward(self, x):
        if (FAST_LAYER_NORM and len(self.normalized_shape) == 1 and _native.backend_for_testing() is None
                and _native.layer_norm_supported(x, self.weight, self.bias) and x.numel() >= 1 << 16):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        if torch.is_grad_enabled() and self.weight is not None:
            _native.fold_mark_plain(self.weight.data_ptr())
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)


class _AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (x + res, LayerNorm(x + res)); the backward folds the gradient arriving at s into the
    LayerNorm backward pass, so neither the forward add nor the backward accumulate is a separate kernel."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        C = x.size(-1)
        s, y, mean, rstd = _native.add_layer_norm_forward(x.contiguous().view(-1, C), res.contiguous().view(-1, C),
                                                          weight, bias, eps)
        ctx.save_for_backward(s, weight, mean, rstd)
        ctx.set_materialize_grads(False)
        return s.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, gs, gy):
        s, weight, mean, rstd = ctx.saved_tensors
        C = s.size(1)
        if gy is None:
            return gs, gs, None, None, None
        extra = None if gs is None else gs.contiguous().view(-1, C)
        dx, dgb = _native.layer_norm_backward(gy.contiguous().view(-1, C), s, mean, rstd, weight, extra)
        dx = dx.view(gy.shape)
        # `res` usually is the output of a Linear (attention projection, merge map, MLP fc2): its bias gradient is the
        # column sum of this very dx, which the kernel has just accumulated
        _clear_column_sums()
        _offer_column_sums(dx, dgb[2 * C:])
        _native.fold_returning(weight.data_ptr())
        return dx, dx, dgb[:C], dgb[C:2 * C], None


FUSED_ADD_LAYER_NORM = True


def add_layer_norm(x, res, norm):
    """-> (x + res, norm(x + res)) for an nn.LayerNorm-like `norm` (the residual add of a block fused into the
    norm that follows it)."""
    if (FUSED_ADD_LAYER_NORM and FAST_LAYER_NORM and isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1
            and _native.backend_for_testing() is None and x.shape == res.shape and res.dtype == x.dtype
            and _native.layer_norm_supported(x, norm.weight, norm.bias) and x.numel() >= 1 << 16):
        return _AddLayerNormFn.apply(x, res, norm.weight, norm.bias, norm.eps)
    s = x + res
    return s, norm(s)


class _MaxTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out, arg = _native.max_tokens_forward(x)
        ctx.save_for_backward(arg)
        ctx.tokens = x.size(1)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return _native.max_tokens_backward(g.contiguous(), arg, ctx.tokens)


FAST_MAX_TOKENS = True


def max_over_tokens(x):
    """torch.max(x, dim=1)[0] for token features x [B,M,C] (the global-feature pools of AdaPoinTr.py:110,225 and
    base_blocks.py:759).  On the GPU one libptops kernel each way: ATen's strided reduction zeroes its semaphores with
    a memset, and a MEMSET node in a replayed HIP graph is not reliable on this stack (csrc/core.hip) — this is what
    keeps the validation forward graph-safe; the backward writes the gradient once instead of zero-fill + scatter."""
    if (FAST_MAX_TOKENS and _native.backend_for_testing() is None and x.dim() == 3 and x.is_contiguous()
            and _native.max_tokens_supported(x)):
        return _MaxTokensFn.apply(x)
    return torch.max(x, dim=1)[0]


# ------------------------------------------------------------------------------ functional
def knn_idx(nsample, xyz, new_xyz):
    """int32 [B,S,nsample] neighbour indices, ordered by (distance, index).  No gradient."""
    return _native.knn(nsample, xyz.detach().contiguous(), new_xyz.detach().contiguous())


def knn_point(nsample, xyz, new_xyz):
    """Reference signature (Transformer_utils.py:19-30): xyz [B,N,3] all points, new_xyz [B,S,3]
    queries -> int64 [B,S,nsample].  The reference returns them unsorted (topk sorted=False);
    here they come ordered by (distance, index), which every consumer accepts."""
    return knn_idx(nsample, xyz, new_xyz).long()


def square_distance(src, dst):
    """[B,N,C] x [B,M,C] -> [B,N,M] expansion-form squared distances (Transformer_utils.py:32-51).
    Kept for API parity (pure torch); the hot path never materialises this matrix."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx3):
        ctx.save_for_backward(idx3)
        ctx.N = points.size(1)
        return _native.index_points_forward(points, idx3)

    @staticmethod
    def backward(ctx, grad_out):
        (idx3,) = ctx.saved_tensors
        return _native.index_points_backward(grad_out.contiguous(), idx3, ctx.N), None


def index_points(points, idx):
    """points [B,N,C], idx [B, ...] (int32 or int64) -> [B, ..., C]  (Transformer_utils.py:53-69)."""
    B = points.size(0)
    shape = list(idx.shape)
    idx3 = idx.reshape(B, -1, 1)
    if idx3.dtype != torch.int32:
        idx3 = idx3.to(torch.int32)
    out = _IndexPoints.apply(points.contiguous(), idx3.contiguous())
    return out.reshape(shape + [points.size(-1)])


class _GatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, idx):
        out, arg = _native.gather_max_forward(p, idx)
        ctx.save_for_backward(arg)
        ctx.N = p.size(1)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)         # no zero tensor for the (integer) second output
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        if grad_out is None:
            return None, None
        return _native.gather_max_backward(grad_out.contiguous(), arg, ctx.N), None


class _GatherMaxAct(torch.autograd.Function):
    """LeakyReLU(max_j p[idx] + add, slope) in one launch each way (csrc/gather.hip); values and gradients are those of
    the three separate ops (same float operations in the same order; the scatter is the deterministic tile)."""

    @staticmethod
    def forward(ctx, p, idx, add, slope):
        out, arg = _native.gather_max_act_forward(p, idx, add, slope)
        ctx.save_for_backward(out, arg)
        ctx.N, ctx.slope = p.size(1), slope
        return out

    @staticmethod
    def backward(ctx, grad_out):
        out, arg = ctx.saved_tensors
        gp, ga = _native.gather_max_act_backward(grad_out.contiguous(), out, arg, ctx.slope, ctx.N)
        return gp, None, ga, None


FUSED_GATHER_MAX_ACT = os.environ.get("PTOPS_FUSED_GATHER_MAX_ACT", "1") != "0"


def gather_max_add_leaky_relu(p, idx, add, slope):
    """leaky_relu(gather_max(p, idx) + add, slope)."""
    if (FUSED_GATHER_MAX_ACT and p.is_cuda and p.dtype == torch.float32 and add.dtype == torch.float32 and p.size(1) <= 1024
            and p.size(0) <= 65535 and _native.backend_for_testing() is None and add.shape[:2] == idx.shape[:2]):
        return _GatherMaxAct.apply(p.contiguous(), idx.contiguous(), add.contiguous(), float(slope))
    return F.leaky_relu(gather_max(p, idx) + add, slope)


def gather_max(p, idx):
    """p [B,N,C], idx int32 [B,S,k] -> max over the k gathered rows, [B,S,C]."""
    return _GatherMax.apply(p.contiguous(), idx.contiguous())[0]


# --------------------------------------------------------------------------------- modules
FUSED_GELU_BACKWARD = True


class _GeluFn(torch.autograd.Function):
    """nn.GELU() whose backward also yields the column sums of its result: the bias gradient of the Linear that fed it."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return F.gelu(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N = x.size(-1)
        dpre, cs = _native.gelu_backward_colsum(x.contiguous().view(-1, N), gy.contiguous().view(-1, N))
        dpre = dpre.view(x.shape)
        _clear_column_sums()
        _offer_column_sums(dpre, cs)
        return dpre


def gelu(x, act):
    """act(x) for an nn.GELU() `act`, on the fused-backward path when it applies."""
    if (FUSED_GELU_BACKWARD and isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" and x.is_cuda
            and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled() and x.size(-1) % 4 == 0
            and x.numel() // x.size(-1) >= 2048 and _native.backend_for_testing() is None):
        return _GeluFn.apply(x)
    if isinstance(act, nn.GELU):
        return F.gelu(x, approximate=getattr(act, "approximate", "none"))
    return act(x)


class GELU(nn.GELU):
    """nn.GELU() (no parameters, same state dict) that takes the fused backward when its input is large enough."""

    def forward(self, x):
        return gelu(x, self)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(linear_gelu(self.fc1, self.act, x))))


def _block_bias(nq, nk, split_q, limit_k, device):
    """Additive [nq,nk] bias of the block mask (0 / -inf), cached per shape and device."""
    key = (nq, nk, split_q, limit_k, str(device))
    b = _bias_cache.get(key)
    if b is None:
        b = torch.zeros(nq, nk, device=device)
        b[:split_q, limit_k:] = float('-inf')
        _bias_cache[key] = b
    return b


class _MFMAAttention(torch.autograd.Function):
    """out[B,N,h,d] = softmax(q k^T scale + block mask) v; q/k/v are [B,h,N,64] (strided views)."""

    @staticmethod
    def forward(ctx, q, k, v, split_q, limit_k, scale):
        out, lse2 = _native.attention_forward(q, k, v, split_q, limit_k, scale)
        ctx.save_for_backward(q, k, v, out, lse2)
        ctx.mask, ctx.scale = (split_q, limit_k), scale
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, out, lse2 = ctx.saved_tensors
        split_q, limit_k = ctx.mask
        B, H, Nq, _ = q.shape
        Nk = k.size(2)
        if USE_MFMA_ATTENTION_BWD:
            # gradients in the token-major layout the projections produce ([B,N,h,d] viewed as [B,h,N,d])
            gq = torch.empty((B, Nq, H, 64), dtype=q.dtype, device=q.device).transpose(1, 2)
            gkv = torch.empty((B, Nk, 2, H, 64), dtype=q.dtype, device=q.device)
            gk, gv = gkv.permute(2, 0, 3, 1, 4).unbind(0)
            _native.attention_backward(q, k, v, out, grad_out.contiguous(), lse2, split_q, limit_k, ctx.scale, gq, gk, gv,
                                       fused=(USE_MFMA_ATTENTION_BWD == "fused"))
            return gq, gk, gv, None, None, None
        bias = None
        if split_q > 0 and limit_k < Nk:
            bias = _block_bias(Nq, Nk, split_q, limit_k, q.device).expand(B, H, Nq, Nk)
        zero = _zero_i64(q.device)   # philox seed / offset: unused without dropout
        gq, gk, gv, _ = torch.ops.aten._efficient_attention_backward(
            grad_out.contiguous(), q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), bias, out, None, None,
            Nq, Nk, lse2 * _LN2, 0.0, zero, zero, 0, False, scale=ctx.scale)
        return gq.transpose(1, 2), gk.transpose(1, 2), gv.transpose(1, 2), None, None, None


class _MFMASelfAttention(torch.autograd.Function):
    """Self attention on the PACKED projection qkv [B,N,3,h,64]: the backward writes dq/dk/dv straight into one
    [B,N,3,h,64] allocation (strided views handed to the MFMA backward kernel; with USE_MFMA_ATTENTION_BWD off,
    aten::_efficient_attention_backward's shared-storage output) and returns it as the gradient of qkv, so autograd
    does not have to zero-fill and scatter three slices per layer."""

    @staticmethod
    def forward(ctx, qkv, split_q, limit_k, scale):
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        out, lse2 = _native.attention_forward(q, k, v, split_q, limit_k, scale)
        ctx.save_for_backward(qkv, out, lse2)
        ctx.mask, ctx.scale = (split_q, limit_k), scale
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qkv, out, lse2 = ctx.saved_tensors
        split_q, limit_k = ctx.mask
        B, N, _, H, D = qkv.shape
        if USE_MFMA_ATTENTION_BWD:
            g = torch.empty_like(qkv)
            q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
            gq, gk, gv = g.permute(2, 0, 3, 1, 4).unbind(0)
            _native.attention_backward(q, k, v, out, grad_out.contiguous(), lse2, split_q, limit_k, ctx.scale, gq, gk, gv,
                                       fused=(USE_MFMA_ATTENTION_BWD == "fused"))
            return g, None, None, None
        q, k, v = qkv.unbind(2)                               # [B,N,h,d] views, the layout the op expects
        bias = None
        if split_q > 0 and limit_k < N:
            bias = _block_bias(N, N, split_q, limit_k, qkv.device).expand(B, H, N, N)
        zero = _zero_i64(qkv.device)
        gq, gk, gv, _ = torch.ops.aten._efficient_attention_backward(
            grad_out.contiguous(), q, k, v, bias, out, None, None, N, N, lse2 * _LN2, 0.0, zero, zero, 0, False,
            scale=ctx.scale, shared_storage_dqdkdv=True)
        packed_strides = (N * 3 * H * D, 3 * H * D, D, 1)
        if (gq.stride() == packed_strides and gk.data_ptr() == gq.data_ptr() + 4 * H * D
                and gv.data_ptr() == gq.data_ptr() + 8 * H * D):
            g = torch.as_strided(gq, (B, N, 3, H, D), (N * 3 * H * D, 3 * H * D, H * D, D, 1))
        else:                                                 # the op did not share storage: assemble
            g = torch.stack([gq, gk, gv], dim=2)
        return g, None, None, None


class _MFMAKVAttention(torch.autograd.Function):
    """Cross attention on a PACKED key / value projection kv [B,M,2,h,64] (one GEMM for k_map and v_map): the backward
    writes dk / dv straight into one [B,M,2,h,64] allocation and returns it as the gradient of kv - contiguous
    [B*M, 2*C] rows for the projection's backward, where two separate projections received two strided slices of it
    (a copy each for the bias column sums, two GEMMs each way instead of one)."""

    @staticmethod
    def forward(ctx, q, kv, scale):
        k, v = kv.permute(2, 0, 3, 1, 4).unbind(0)
        out, lse2 = _native.attention_forward(q, k, v, 0, k.size(2), scale)
        ctx.save_for_backward(q, kv, out, lse2)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, kv, out, lse2 = ctx.saved_tensors
        B, H, Nq, _ = q.shape
        k, v = kv.permute(2, 0, 3, 1, 4).unbind(0)
        gq = torch.empty((B, Nq, H, 64), dtype=q.dtype, device=q.device).transpose(1, 2)
        gkv = torch.empty_like(kv)
        gk, gv = gkv.permute(2, 0, 3, 1, 4).unbind(0)
        _native.attention_backward(q, k, v, out, grad_out.contiguous(), lse2, 0, k.size(2), ctx.scale, gq, gk, gv,
                                   fused=(USE_MFMA_ATTENTION_BWD == "fused"))
        return gq, gkv, None


# CrossAttention: k_map and v_map (same input, separate parameters in the reference and in the state dict) as ONE
# [2C, C] projection on the concatenated weights; PTOPS_FUSED_KV=0 keeps the two GEMMs
FUSED_KV_PROJECTION = os.environ.get("PTOPS_FUSED_KV", "1") != "0"


def _attend(q, k, v, scale, mask, attn_drop, mask_split=None):
    """q [B,h,N,d], k/v [B,h,M,d]; mask bool [N,M] with True = blocked (reference convention);
    mask_split = (split_q, limit_k) when the mask is the block mask "queries < split_q do not see
    keys >= limit_k".  Returns [B,N,h,d]."""
    no_drop = attn_drop.p == 0.0 or not attn_drop.training
    if USE_MFMA_ATTENTION and no_drop and (mask is None or mask_split is not None):
        split_q, limit_k = mask_split if mask_split is not None else (0, k.size(2))
        if _native.backend_for_testing() is None and _native.attention_supported(q, k, split_q, limit_k):
            return _MFMAAttention.apply(q, k, v, split_q, limit_k, scale)
    return _attend_torch(q, k, v, scale, mask, attn_drop).transpose(1, 2)


def _attend_torch(q, k, v, scale, mask, attn_drop):
    if USE_SDPA and (attn_drop.p == 0.0 or not attn_drop.training):
        keep = None if mask is None else ~mask
        return F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=scale)
    attn = (q @ k.transpose(-2, -1)) * scale
    if mask is not None:
        attn = attn.masked_fill(mask, -torch.finfo(attn.dtype).max)
    attn = attn_drop(attn.softmax(dim=-1))
    return attn @ v


class Attention(nn.Module):
    """Multi-head self attention with an optional [N,N] block mask (Transformer_utils.py:89-122)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, mask=None, mask_split=None):
        """mask: [N,N], > 0 = blocked (reference semantics).  mask_split = (split_q, limit_k), optional:
        tells that `mask` is the block mask "queries < split_q do not see keys >= limit_k"."""
        B, N, C = x.shape
        packed = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        qkv = packed.permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if (USE_MFMA_ATTENTION and (mask is None or mask_split is not None) and _native.backend_for_testing() is None
                and (self.attn_drop.p == 0.0 or not self.training)):
            split_q, limit_k = mask_split if mask_split is not None else (0, N)
            if _native.attention_supported(q, k, split_q, limit_k):
                x = _MFMASelfAttention.apply(packed, split_q, limit_k, self.scale).reshape(B, N, C)
                return self.proj_drop(self.proj(x))
        if mask is not None:
            mask = mask > 0  # 1 = blocked
        x = _attend(q, k, v, self.scale, mask, self.attn_drop, mask_split).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class CrossAttention(nn.Module):
    """Multi-head cross attention, separate q/k/v maps (Transformer_utils.py:124-159)."""

    def __init__(self, dim, out_dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.dim = dim
        self.out_dim = out_dim
        head_dim = out_dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.q_map = Linear(dim, out_dim, bias=qkv_bias)
        self.k_map = Linear(dim, out_dim, bias=qkv_bias)
        self.v_map = Linear(dim, out_dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(out_dim, out_dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def project_kv(self, k, n_queries):
        """k_map and v_map of the keys / values as ONE projection on the concatenated weights -> packed [B,M,2,h,64] for
        `_MFMAKVAttention`, or None where that form does not apply (the caller then runs the two maps).  A separate step so
        that a decoder can run it for all its blocks ahead of the query chain (the memory does not depend on the queries)."""
        B, NK, _ = k.shape
        C, h = self.out_dim, self.num_heads
        if not (FUSED_KV_PROJECTION and USE_MFMA_ATTENTION and USE_MFMA_ATTENTION_BWD and C // h == 64 and k.is_cuda
                and k.dtype == torch.float32 and _native.backend_for_testing() is None and torch.is_grad_enabled()
                and self.k_map.bias is not None and self.v_map.bias is not None and B * NK >= 2048
                and (self.attn_drop.p == 0.0 or not self.training) and n_queries % 32 == 0 and NK % 32 == 0):
            return None
        wkv = torch.cat([self.k_map.weight, self.v_map.weight], dim=0)            # [2C, C]: 1.2 MB, once per call
        bkv = torch.cat([self.k_map.bias, self.v_map.bias], dim=0)
        wkv._ptops_keys = (self.k_map.weight.data_ptr(), self.v_map.weight.data_ptr())
        bkv._ptops_keys = (self.k_map.bias.data_ptr(), self.v_map.bias.data_ptr())
        # the weight / bias gradients travel on through cat's backward as two dim-0 views each, which AccumulateGrad
        # adopts without reading (dense, strides of the parameter, sole owner): they may stay queued (deferred
        # folds / grouped weight gradients; test_cross_attention_fused_kv_projection runs this with poisoned outputs)
        return _LinearFn.apply(k.reshape(B * NK, -1), wkv, bkv).view(B, NK, 2, h, C // h)

    def forward(self, q, v, kv=None):
        B, N, _ = q.shape
        C = self.out_dim
        k = v
        NK = k.size(1)
        h = self.num_heads
        q = self.q_map(q).view(B, N, h, C // h).permute(0, 2, 1, 3)
        if kv is None:
            kv = self.project_kv(k, N)
        if kv is not None:
            x = _MFMAKVAttention.apply(q, kv, self.scale).reshape(B, N, C)
            return self.proj_drop(self.proj(x))
        k = self.k_map(k).view(B, NK, h, C // h).permute(0, 2, 1, 3)
        v = self.v_map(v).view(B, NK, h, C // h).permute(0, 2, 1, 3)
        x = _attend(q, k, v, self.scale, None, self.attn_drop).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class DynamicGraphAttention(nn.Module):
    """max_k LeakyReLU(W . cat(v[idx] - q, q))  (Transformer_utils.py:779-860).

    Parameters are laid out exactly as in the reference (`knn_map.0.weight` [C,2C], `.bias`)."""

    def __init__(self, dim, k=10):
        super().__init__()
        self.dim = dim
        self.k = k
        self.knn_map = nn.Sequential(nn.Linear(dim * 2, dim), nn.LeakyReLU(negative_slope=0.2))

    def _aggregate(self, q, v, idx):
        """q [B,S,C], v [B,N,C], idx int32 [B,S,k] -> [B,S,C]."""
        C = self.dim
        lin = self.knn_map[0]
        if FUSED_GRAPH_ATTENTION:
            w1, w21 = _SplitEdgeWeight.apply(lin.weight)
            p = F.linear(v, w1)                    # per source token
            qt = F.linear(q, w21, lin.bias)        # per query token
            return gather_max_add_leaky_relu(p, idx, qt, self.knn_map[1].negative_slope)
        local_v = index_points(v, idx)             # [B,S,k,C]
        qe = q.unsqueeze(-2).expand(-1, -1, idx.size(-1), -1)
        feature = torch.cat((local_v - qe, qe), dim=-1)
        return self.knn_map(feature).max(-2)[0]

    def forward(self, q, q_pos, v=None, v_pos=None, idx=None, denoise_length=None):
        if denoise_length is None:
            if v is None:
                v = q
            if v_pos is None:
                v_pos = q_pos
            assert v_pos.dim() == 3 and v_pos.size(-1) == 3, f'unexpected v_pos shape {tuple(v_pos.shape)}'
            assert q_pos.dim() == 3 and q_pos.size(-1) == 3, f'unexpected q_pos shape {tuple(q_pos.shape)}'
            assert q.size(-1) == v.size(-1) == self.dim
            if idx is None:
                idx = knn_idx(self.k, v_pos, q_pos)
            assert idx.size(-1) == self.k
            if idx.dtype != torch.int32:
                idx = idx.to(torch.int32)
            return self._aggregate(q, v, idx)
        # Denoise task (self attention only, Transformer_utils.py:821-859): real queries see only
        # real queries, denoise queries see everything; both index the same token array because
        # the real queries are its prefix.
        assert idx is None, 'online index calculation is required when denoise_length is set'
        assert v is None and v_pos is None, 'denoise_length is only defined for self attention'
        dl = denoise_length
        idx_r = knn_idx(self.k, q_pos[:, :-dl], q_pos[:, :-dl])
        idx_n = knn_idx(self.k, q_pos, q_pos[:, -dl:])
        return self._aggregate(q, q, torch.cat([idx_r, idx_n], dim=1))


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x.mul_(self.gamma) if self.inplace else x * self.gamma
