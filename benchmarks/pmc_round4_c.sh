#!/bin/bash
# Late round 4: counters of the Chamfer MFMA search with the in-kernel split (no prep pass).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/gpurun_out"
bash "$root/benchmarks/pmc_passes.sh" pmc4c_chamfer benchmarks/chamfer_only.py 4 16384 16384 mfma
python3 "$root/benchmarks/pmc_chamfer_mfma.py" "$root"/gpurun_out/pmc4c_chamfer_0 "$root"/gpurun_out/pmc4c_chamfer_1 "$root"/gpurun_out/pmc4c_chamfer_2 "$root"/gpurun_out/pmc4c_chamfer_3 > "$root/gpurun_out/chamfer_mfma_pmc_r4c.json"
