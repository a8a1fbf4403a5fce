#!/bin/bash
# Round-4 counter passes (run on the GPU box from the repo root): Chamfer MFMA search, FPS, attention.
# One rocprofv3 pass per counter group; --pmc only with --kernel-trace (gpurun refuses other trace domains with it).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/gpurun_out"
bash "$root/benchmarks/pmc_passes.sh" pmc4_chamfer benchmarks/chamfer_only.py 4 16384 16384 mfma
python3 "$root/benchmarks/pmc_chamfer_mfma.py" "$root"/gpurun_out/pmc4_chamfer_0 "$root"/gpurun_out/pmc4_chamfer_1 "$root"/gpurun_out/pmc4_chamfer_2 "$root"/gpurun_out/pmc4_chamfer_3 > "$root/gpurun_out/chamfer_mfma_pmc_r4.json"
bash "$root/benchmarks/pmc_passes.sh" pmc4_fps benchmarks/primitives_only.py fps 5
python3 "$root/benchmarks/pmc_kernels.py" "FPS launches of one bs=32 step (2048->512, 512->256), round-4 kernel" fps_kernel "$root"/gpurun_out/pmc4_fps_0 "$root"/gpurun_out/pmc4_fps_1 "$root"/gpurun_out/pmc4_fps_2 "$root"/gpurun_out/pmc4_fps_3 > "$root/gpurun_out/fps_pmc_r4.json"
