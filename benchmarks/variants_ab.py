"""A/B of the switchable fast paths on the batch-16 model variants (BASELINE configs[2], [4]): one fresh process per
(variant, setting) on ONE box, so the rows are comparable.  Finds which round-2 change cost the variants their round-1
times (29.9 -> 34.9 ms, 42.7 -> 50.3 ms).

    python benchmarks/variants_ab.py [img|seg|ada ...] [--only=substring,substring of the setting names]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SETTINGS = [
    ("default", {}),
    ("attn fwd f32", {"PTOPS_ATTN_FWD_F32": "1"}),
    ("attn bwd f32 pair", {"PTOPS_ATTN_BWD_F32": "1"}),
    ("attn bwd fused f32", {"PTOPS_ATTN_BWD": "fused"}),
    ("attn fwd f32 + bwd fused f32", {"PTOPS_ATTN_FWD_F32": "1", "PTOPS_ATTN_BWD": "fused"}),
    ("no deferred folds", {"PTOPS_DEFER_FOLDS": "0"}),
    ("stock clip + AdamW", {"PTOPS_FUSED_ADAMW": "0"}),
    ("library GEMMs", {"PTOPS_SPLIT3_GEMM": "0"}),
    ("library weight gradients", {"PTOPS_SPLIT3_WGRAD": "0"}),
    ("no Linear+GELU fusion", {"PTOPS_FUSED_LINEAR_GELU": "0"}),
    ("per-layer weight gradients (round 3: grouped)", {"PTOPS_GROUPED_WGRAD": "0"}),
    ("k_map / v_map as two projections", {"PTOPS_FUSED_KV": "0"}),
    ("literal Chamfer reductions", {"PTOPS_FUSED_CHAMFER_REDUCTIONS": "0"}),
    ("all round-2 switches off", {"PTOPS_ATTN_FWD_F32": "1", "PTOPS_ATTN_BWD": "fused", "PTOPS_DEFER_FOLDS": "0",
                                  "PTOPS_FUSED_ADAMW": "0", "PTOPS_SPLIT3_GEMM": "0", "PTOPS_SPLIT3_WGRAD": "0",
                                  "PTOPS_FUSED_LINEAR_GELU": "0"}),
]
KEYS = {"img": "ImgResNetEnc", "seg": "ImgEncSegDec", "ada": "AdaPoinTr (PCN)"}


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("--only=")] or ["img", "seg"]
    only = [a[7:].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    settings = [s for s in SETTINGS if not only or any(k in s[0] for k in only[0])]
    for w in which:
        for name, env in settings:
            e = dict(os.environ)
            e.update(env)
            r = subprocess.run([sys.executable, os.path.join(HERE, "model_variants_bench.py"), KEYS[w]], env=e,
                               capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if "train step" in ln]
            print(f"[{name:32s}] {line[0] if line else 'FAILED: ' + r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
