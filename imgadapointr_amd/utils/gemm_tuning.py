"""Per-shape GEMM kernel selection for the dense layers (PyTorch-ROCm TunableOp).

The transformer / MLP GEMMs of the path stay on the vendor libraries (hipBLASLt, rocBLAS); which
of their fp32 MFMA kernels runs for a given (M, N, K, layout) is chosen from
`imgadapointr_amd/tuning/gemm_gfx950.csv`, measured on an MI355X for exactly the shapes of the
AdaPoinTr train step at bs=32 (benchmarks/step_variants.py --tunable): 66.2 -> 54.6 ms/step.
Shapes missing from the file are tuned on first use when `tune_missing` is set.  The results are
identical fp32 GEMMs; only the kernel choice changes.
"""
import os
import shutil
import tempfile

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "gemm_gfx950.csv")


def enable_tuned_gemm(path=None, tune_missing=True, max_tuning_ms=30, max_tuning_iters=20):
    """Turn TunableOp on with the shipped selections.  Returns a short description for logs."""
    if not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return "off (no GPU / no TunableOp)"
    path = path or SHIPPED
    tun = torch.cuda.tunable
    # work on a private copy: TunableOp rewrites its file when it tunes something new
    work = os.path.join(tempfile.gettempdir(), f"ptops_tunableop_{os.getpid()}.csv")
    have = os.path.exists(path)
    if have:
        shutil.copyfile(path, work)
    tun.enable(True)
    tun.set_filename(work)
    tun.tuning_enable(bool(tune_missing))
    tun.set_max_tuning_duration(int(max_tuning_ms))
    tun.set_max_tuning_iterations(int(max_tuning_iters))
    if hasattr(tun, "write_file_on_exit"):
        tun.write_file_on_exit(False)
    return f"tunableop({'shipped selections' if have else 'no file'}{', tuning missing shapes' if tune_missing else ''})"
