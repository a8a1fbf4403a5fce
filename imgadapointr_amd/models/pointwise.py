"""Shared-weight per-point MLPs (`nn.Sequential` of 1x1 `Conv1d` / `Linear`, `BatchNorm1d`, activation) executed
token-major on [rows, C].

The reference's Chamfer-only baselines (models/external/PCN.py, FoldingNet.py, TopNet.py) run these stacks
channel-first on `[B, C, N]` and feed them concatenations of a per-cloud code broadcast over all N points with a few
per-point channels (`cat([g.expand(B,1024,N), seed, xyz])`, PCN.py:72-74; FoldingNet.py:73-78; TopNet.py:97-102).
Here the modules keep the reference's parameter shapes and state-dict keys, but

 * every 1x1 convolution is a GEMM on rows (no transposes, no `[B, C, N]` copies),
 * the first convolution of a stack that consumes such a concatenation is split by input columns: the broadcast part
   is one `[B, G] x [G, O]` product per cloud, the per-point part a `[rows, X] x [X, O]` product with X <= 8, and the
   concatenation (2.1 GB at B = 32 for PCN's final stack) is never formed,
 * training-mode BatchNorm + ReLU over rows is libptops' column-moment kernel pair (csrc/batchnorm.hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native
from .AdaPoinTr import _BatchNormRowsLeakyReLU, _linear_tokens

# False: the literal form - materialise the concatenation and run the stack channel-first through the modules
# themselves, as the reference does (kept as the referee for tests and benchmarks/baselines_bench.py)
TOKEN_MAJOR = True


def _weight2d(layer):
    if isinstance(layer, nn.Conv1d):
        assert layer.kernel_size == (1,) and layer.stride == (1,) and layer.groups == 1, 'point-wise convolutions only'
        return layer.weight.squeeze(-1)
    assert isinstance(layer, nn.Linear)
    return layer.weight


def _norm_act(bn, act, h):
    """act(bn(h)) for h [rows, C]; `act` may be None."""
    slope = 0.0 if isinstance(act, nn.ReLU) else getattr(act, 'negative_slope', None)
    if (bn.training and h.is_cuda and h.dtype == torch.float32 and _native.backend_for_testing() is None
            and bn.affine and bn.track_running_stats and slope is not None and h.size(1) % 4 == 0 and h.size(0) >= 64):
        return _BatchNormRowsLeakyReLU.apply(h.contiguous(), bn.weight, bn.bias, bn, slope)
    if h.is_cuda or not bn.training:
        h = bn(h)                                                       # BatchNorm1d accepts [rows, C]
    else:
        # host path (tests): ATen's CPU batch statistics over [rows, C] accumulate serially in fp32 and are ~8x less
        # accurate than its [1, C, rows] path; deep train-mode stacks amplify that (tests/parity_checks.py)
        h = bn(h.t().unsqueeze(0)).squeeze(0).t()
    if act is None:
        return h
    return F.relu(h) if isinstance(act, nn.ReLU) else act(h)           # out-of-place: h may be saved by bn


def first_layer_on_parts(layer, parts):
    """`layer(cat(parts, channel))` without the concatenation.  `parts` lists, in the channel order of the
    reference's `torch.cat`, tensors shaped [d0, d1, ..., C_i] whose leading dims are broadcastable against each other
    (size-1 dims mark what the reference expands).  Returns the broadcast sum of the per-part products, bias folded
    into the smallest part."""
    w = _weight2d(layer)
    assert sum(p.size(-1) for p in parts) == w.size(1)
    order = sorted(range(len(parts)), key=lambda i: parts[i].numel())
    out, col = None, 0
    cols = []
    for p in parts:
        cols.append((col, col + p.size(-1)))
        col += p.size(-1)
    for n, i in enumerate(order):
        a, b = cols[i]
        term = F.linear(parts[i], w[:, a:b], layer.bias if n == 0 else None)
        out = term if out is None else out + term
    return out


def run_rows(seq, h, start=0):
    """Applies seq[start:] to h [rows, C]."""
    if not TOKEN_MAJOR and start == 0:
        return seq(h)
    layers = list(seq)
    i = start
    while i < len(layers):
        m = layers[i]
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            h = _linear_tokens(h, _weight2d(m), m.bias)
        elif isinstance(m, nn.BatchNorm1d):
            act = layers[i + 1] if i + 1 < len(layers) and isinstance(layers[i + 1], (nn.ReLU, nn.LeakyReLU)) else None
            h = _norm_act(m, act, h)
            i += act is not None
        elif isinstance(m, nn.ReLU):
            h = F.relu(h)
        else:
            h = m(h)
        i += 1
    return h


def run_points(seq, x):
    """seq over x [B, N, C] -> [B, N, C']."""
    B, N, C = x.shape
    if not TOKEN_MAJOR:
        return seq(x.transpose(1, 2)).transpose(1, 2)
    return run_rows(seq, x.reshape(B * N, C)).view(B, N, -1)


def run_parts(seq, parts):
    """seq over the (virtual) channel concatenation of `parts` -> [d0, d1, ..., C']."""
    if not TOKEN_MAJOR:
        lead = torch.broadcast_shapes(*[p.shape[:-1] for p in parts])
        x = torch.cat([p.expand(*lead, p.size(-1)) for p in parts], dim=-1)
        y = seq(x.reshape(lead[0], -1, x.size(-1)).transpose(1, 2))              # [B, C, N], as the reference
        return y.transpose(1, 2).reshape(*lead, -1)
    h = first_layer_on_parts(seq[0], parts)
    lead = h.shape[:-1]
    return run_rows(seq, h.reshape(-1, h.size(-1)), start=1).view(*lead, -1)
