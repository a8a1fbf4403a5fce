#!/bin/bash
# rocprofv3 counter passes of one target program, one pass per counter group (SQ/GRBM, SQ LDS, FETCH_SIZE, WRITE_SIZE) as
# MI355X_MICROARCH.md prescribes; --pmc only together with --kernel-trace.  Usage:
#   benchmarks/pmc_passes.sh <out-prefix under gpurun_out/> <program args after python3...>
# e.g. benchmarks/pmc_passes.sh pmc_fps benchmarks/primitives_only.py fps 5
set -e
prefix=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/gpurun_out/${prefix}_$i" -o p -- python3 "$root/$1" "${@:2}" > "$root/gpurun_out/${prefix}_$i.log" 2>&1
  i=$((i+1))
done
