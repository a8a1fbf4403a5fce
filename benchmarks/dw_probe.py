"""dW = dY^T X two ways (TunableOp on): g.t() @ x  vs  (x.t() @ g).t().contiguous()."""
import torch
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(50); torch.cuda.tunable.set_max_tuning_iterations(50)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N, K in [(18432, 384, 384), (18432, 1152, 384), (18432, 768, 384), (18432, 384, 768), (8192, 384, 384), (8192, 768, 384), (18432, 1024, 1024)]:
    g = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    a = t(lambda: g.t() @ x)
    b = t(lambda: (x.t() @ g).t().contiguous())
    c = t(lambda: x.t() @ g)
    fl = 2.0 * M * N * K
    print(f"M={M} out={N} in={K}: g.t()@x {a:7.1f} us ({fl/a/1e6:6.1f} TF)   (x.t()@g).t().contiguous() {b:7.1f} us   x.t()@g alone {c:7.1f} us ({fl/c/1e6:6.1f} TF)")
