"""Where does the split-bf16 attention forward spend its time?  Runs the kernel with parts of its work removed
(PTOPS_ATTN_FWD_ABL bits: 1 = K / V staged without the three-piece split, 2 = P without it, 4 = no softmax arithmetic,
8 = no S^T MFMAs, 16 = no O^T MFMAs, 32 = no loads / staging after the first tile; results are wrong by construction) at the
three training shapes (B = 32, 6 heads) and prints microseconds per launch.

    make -C imgadapointr_amd/csrc clean && make -C imgadapointr_amd/csrc -j8 TUNING=1      # the variants are tuning-build only
    python benchmarks/attention_ablation.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgadapointr_amd import _native as nat  # noqa: E402

nat.load()
B, H, d = 32, 6, 64
torch.manual_seed(0)
shapes = ((576, 576, 512, 512), (576, 256, 0, 256), (256, 256, 0, 256))
cases = [(0, "full kernel"), (1, "K/V staging without the split"), (2, "P without the split"), (3, "neither split"),
         (4, "no softmax arithmetic"), (7, "no splits, no softmax"), (8, "no S^T MFMAs"), (16, "no O^T MFMAs"),
         (24, "no MFMAs at all"), (32, "no loads / staging after tile 0"), (35, "no staging, no splits"),
         (39, "no staging, no splits, no softmax (MFMAs + fragment reads only)"), (63, "barriers and fragment reads only")]
for Nq, Nk, split, limit in shapes:
    q = torch.randn(B, H, Nq, d, device="cuda")
    k = torch.randn(B, H, Nk, d, device="cuda")
    v = torch.randn(B, H, Nk, d, device="cuda")
    print(f"shape {Nq} x {Nk} (split {split}, limit {limit})")
    for abl, label in cases:
        nat.set_option("PTOPS_ATTN_FWD_ABL", abl)
        for _ in range(3):
            nat.attention_forward(q, k, v, split, limit, d ** -0.5)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            nat.attention_forward(q, k, v, split, limit, d ** -0.5)
        e.record()
        torch.cuda.synchronize()
        print(f"  ABL {abl:2d}  {s.elapsed_time(e) / 20 * 1e3:7.1f} us   {label}")
    nat.set_option("PTOPS_ATTN_FWD_ABL", 0)
