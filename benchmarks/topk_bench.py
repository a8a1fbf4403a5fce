"""Row-wise top-k of GDANet's feature-space kNN: libptops (csrc/topk.hip) vs ATen, [16,512,512] rows, k = 30 / 64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imgadapointr_amd import _native  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    torch.manual_seed(0)
    for B, N, k in ((16, 512, 30), (16, 512, 64), (32, 512, 30), (16, 1024, 30), (16, 2048, 64)):
        x = torch.randn(B, N, N, device="cuda")
        want = x.topk(k, dim=-1)[1]
        got = _native.topk_rows(x, k)
        same = bool(torch.equal(want, got))
        print(f"[{B},{N},{N}] k={k}: aten {timed(lambda: x.topk(k, dim=-1)):7.1f} us   libptops "
              f"{timed(lambda: _native.topk_rows(x, k)):7.1f} us   identical={same}", flush=True)


if __name__ == "__main__":
    main()
