"""Micro-benchmarks of the libptops kernels at the AdaPoinTr hot-path shapes.

Usage: python benchmarks/ops_bench.py [--B 32] [--iters 20]
Times each op with events on torch's current stream (the stream the kernels launch on) and
prints ms/launch plus the derived pair-evals/s (Chamfer, kNN) or GB/s (movers).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B = a.B
    torch.manual_seed(0)
    dev = "cuda"
    rows = []

    for n, m in [(2048, 2048), (512, 16384), (16384, 16384)]:
        x = torch.rand(B, n, 3, device=dev) - 0.5
        y = torch.rand(B, m, 3, device=dev) - 0.5
        ms = timeit(lambda: nat.chamfer_forward(x, y), a.iters)
        pairs = 2.0 * B * n * m
        rows.append(dict(op="chamfer_fwd", shape=f"{B}x{n}x{m}", ms=ms, gpairs_s=pairs / ms / 1e6,
                         valu_tflops=pairs * 8 / ms / 1e9, alg_gbs=20.0 * B * (n + m) / ms / 1e6))
        d1, d2, i1, i2 = nat.chamfer_forward(x, y)
        g1, g2 = torch.randn_like(d1), torch.randn_like(d2)
        ms = timeit(lambda: nat.chamfer_backward(x, y, i1, i2, g1, g2), a.iters)
        rows.append(dict(op="chamfer_bwd", shape=f"{B}x{n}x{m}", ms=ms, alg_gbs=32.0 * B * (n + m) / ms / 1e6))

    for N, npnt in [(2048, 512), (512, 256), (2048, 256), (2048, 64)]:
        x = torch.rand(B, N, 3, device=dev) - 0.5
        ms = timeit(lambda: nat.furthest_point_sample(x, npnt), a.iters)
        rows.append(dict(op="fps", shape=f"{B}x{N}->{npnt}", ms=ms, us_per_iter=ms * 1e3 / max(npnt - 1, 1)))

    for N, S, k in [(2048, 2048, 16), (2048, 512, 16), (512, 512, 16), (512, 256, 16), (256, 256, 8),
                    (256, 576, 8), (512, 512, 8), (576, 64, 8), (16384, 64, 32)]:
        c = torch.rand(B, N, 3, device=dev) - 0.5
        q = torch.rand(B, S, 3, device=dev) - 0.5
        ms = timeit(lambda: nat.knn(k, c, q), a.iters)
        rows.append(dict(op="knn", shape=f"{B}x{S}x{N} k{k}", ms=ms, gpairs_s=B * S * N / ms / 1e6))

    for C, Nk, Nq in [(8, 2048, 2048), (32, 2048, 512), (64, 512, 512), (64, 512, 256)]:
        xk = torch.randn(B, C, Nk, device=dev)
        xq = torch.randn(B, C, Nq, device=dev)
        idx = torch.randint(0, Nk, (B, Nq, 16), device=dev, dtype=torch.int32)
        ms = timeit(lambda: nat.edge_group_forward(xk, xq, idx), a.iters)
        byts = 4.0 * B * 2 * C * Nq * 16
        rows.append(dict(op="edge_group_fwd", shape=f"{B}x{C}x{Nk}x{Nq}", ms=ms, alg_gbs=byts / ms / 1e6))
        go = torch.randn(B, 2 * C, Nq, 16, device=dev)
        ms = timeit(lambda: nat.edge_group_backward(go, idx, Nk), a.iters)
        rows.append(dict(op="edge_group_bwd", shape=f"{B}x{C}x{Nk}x{Nq}", ms=ms, alg_gbs=byts / ms / 1e6))

    for N, S in [(256, 256), (576, 576), (256, 576)]:
        p = torch.randn(B, N, 384, device=dev)
        idx = torch.randint(0, N, (B, S, 8), device=dev, dtype=torch.int32)
        ms = timeit(lambda: nat.gather_max_forward(p, idx), a.iters)
        rows.append(dict(op="gather_max_fwd", shape=f"{B}x{N}x{S}", ms=ms,
                         alg_gbs=4.0 * B * S * 384 * (8 + 2) / ms / 1e6))
        out, arg = nat.gather_max_forward(p, idx)
        go = torch.randn_like(out)
        ms = timeit(lambda: nat.gather_max_backward(go, arg, N), a.iters)
        rows.append(dict(op="gather_max_bwd", shape=f"{B}x{N}x{S}", ms=ms))
        ms = timeit(lambda: nat.index_points_forward(p, idx), a.iters)
        rows.append(dict(op="index_points_fwd", shape=f"{B}x{N}x{S}x8x384", ms=ms,
                         alg_gbs=4.0 * B * S * 8 * 384 * 2 / ms / 1e6))

    for r in rows:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
