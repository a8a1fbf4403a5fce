"""Helpers shared by tests/golden/make_golden.py and the tests: deterministic weights and the
synthetic inputs of SURVEY §8d.  Pure numpy/torch; knows nothing about the reference."""
import zlib

import numpy as np
import torch


def seeded_state_dict(sd, tag, std=0.02):
    """A state_dict with the keys/shapes/dtypes of `sd`, filled as a pure function of
    (tag, key, shape).  1-D '*.weight' (norm scales) are 1 + 0.1 n with mixed small deviations,
    running_var is positive, integer buffers are zero, everything else is std * n."""
    out = {}
    for key in sd:
        ref = sd[key]
        g = torch.Generator().manual_seed(zlib.crc32(f"{tag}/{key}".encode()) & 0x7FFFFFFF)
        if not ref.is_floating_point():
            out[key] = torch.zeros_like(ref)
            continue
        n = torch.randn(ref.shape, generator=g, dtype=torch.float32)
        if key.endswith("running_var"):
            t = 1.0 + 0.1 * n.abs()
        elif key.endswith("running_mean"):
            t = 0.05 * n
        elif ref.dim() == 1 and key.endswith("weight"):
            t = 1.0 + 0.1 * n
        elif "query_ranking" in key:
            # wide scores: the top-512-of-768 query selection (argsort) must not hinge on 1e-7 noise
            # (std 0.3 gives pre-sigmoid scores of order 1: spread out, not saturated)
            t = (0.3 if key.endswith("weight") else 0.1) * n
        else:
            t = std * n
        out[key] = t.to(ref.dtype)
    return out


def synthetic_batch(B, n_partial, n_gt, seed=0, structured=False):
    """(partial [B,n_partial,3], gt [B,n_gt,3]) float32.

    Plain: uniform in [-0.5, 0.5)^3.  Structured: points on sphere shells of random radius, the
    last 25 % of the partial rows zeroed (the zero padding RandomSamplePoints produces,
    datasets/data_transforms.py:55-67 of the reference)."""
    rng = np.random.default_rng(seed)
    if not structured:
        return (rng.random((B, n_partial, 3), dtype=np.float32) - 0.5,
                rng.random((B, n_gt, 3), dtype=np.float32) - 0.5)

    def shell(n):
        v = rng.standard_normal((B, n, 3)).astype(np.float32)
        v /= np.linalg.norm(v, axis=-1, keepdims=True) + 1e-9
        return (v * (0.25 + 0.2 * rng.random((B, 1, 1), dtype=np.float32))).astype(np.float32)

    partial, gt = shell(n_partial), shell(n_gt)
    partial[:, n_partial - n_partial // 4:, :] = 0.0
    return partial, gt


def demo_partial(pcd_path, n=2048):
    """ASCII .pcd -> [n,3] float32: the points in file order, zero-padded (or truncated) to n."""
    pts = []
    with open(pcd_path) as fh:
        data = False
        for line in fh:
            if data:
                vals = line.split()
                if len(vals) >= 3:
                    pts.append([float(vals[0]), float(vals[1]), float(vals[2])])
            elif line.startswith("DATA"):
                assert "ascii" in line
                data = True
    a = np.asarray(pts, dtype=np.float32)[:n]
    out = np.zeros((n, 3), np.float32)
    out[:len(a)] = a
    return out


def synthetic_image(B, seed=0):
    """[B,3,224,224] float32 in [0,1): the value range torchvision's ToTensor produces."""
    return np.random.default_rng(seed).random((B, 3, 224, 224), dtype=np.float32)
