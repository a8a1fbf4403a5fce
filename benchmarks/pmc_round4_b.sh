#!/bin/bash
# Round-4 counter passes, part 2 (run on the GPU box from the repo root): the attention kernels as they are now, kNN with the
# fused DPP chains.  One rocprofv3 pass per counter group; --pmc only with --kernel-trace.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
g="$root/gpurun_out"
bash "$root/benchmarks/pmc_passes.sh" pmc4_attn benchmarks/attention_pmc_driver.py
python3 "$root/benchmarks/pmc_attention.py" "$g/pmc4_attn_0/p_counter_collection.csv" "$g/pmc4_attn_0/p_kernel_trace.csv" \
    "$g/pmc4_attn_2/p_counter_collection.csv" "$g/pmc4_attn_3/p_counter_collection.csv" > "$g/attention_pmc_r4.json"
bash "$root/benchmarks/pmc_passes.sh" pmc4_knn benchmarks/primitives_only.py knn 5
python3 "$root/benchmarks/pmc_kernels.py" "kNN launches of one bs=32 step, round-4 kernel (fused DPP min chains, 32 queries per workgroup on the large calls)" knn_kernel "$g/pmc4_knn_0" "$g/pmc4_knn_1" "$g/pmc4_knn_2" "$g/pmc4_knn_3" > "$g/knn_pmc_r4.json"
