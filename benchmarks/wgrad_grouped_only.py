"""The grouped weight-gradient launch of one bs = 32 AdaPoinTr step on synthetic operands (the 93 shapes of
profiles/round3/wgrad_group_probe_v2.log), a few times: target of the rocprofv3 counter passes. argv: iterations"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native  # noqa: E402

SHAPES = [(8, 18432, 1152, 384, 1), (9, 18432, 384, 768, 0), (8, 18432, 768, 384, 0), (14, 18432, 384, 384, 0),
          (10, 18432, 384, 384, 1), (6, 8192, 1152, 384, 1), (1, 18432, 1024, 1024, 1), (8, 8192, 768, 384, 1),
          (1, 18432, 1024, 1024, 0), (6, 8192, 768, 384, 0), (6, 8192, 384, 768, 0), (1, 8192, 1024, 1024, 1),
          (1, 18432, 1024, 384, 1), (1, 18432, 384, 1024, 1), (5, 8192, 384, 384, 0), (1, 18432, 384, 768, 1),
          (1, 8192, 1024, 384, 0), (1, 8192, 384, 768, 1), (1, 8192, 384, 512, 1), (1, 8192, 384, 384, 1),
          (1, 18432, 96, 512, 1), (1, 8192, 512, 128, 0), (1, 8192, 384, 128, 1)]


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    _native.load()
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    # operands shared between layers of one shape (the counters do not care; 93 private pairs would be 5 GB)
    pools = {}
    probs = []
    for count, M, N, K, wb in SHAPES:
        dy = pools.setdefault(("dy", M, N), torch.randn(M, N, device=dev))
        x = pools.setdefault(("x", M, K), torch.randn(M, K, device=dev))
        probs += [(dy, x, bool(wb))] * count
    print(len(probs), "problems,", sum(2e-9 * dy.size(0) * dy.size(1) * x.size(1) for dy, x, _ in probs), "GFLOP")
    for _ in range(it):
        outs = [_native.wgrad_defer(dy, x, wb) for dy, x, wb in probs]
        _native.wgrad_flush()
        del outs
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
