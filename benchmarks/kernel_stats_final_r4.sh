#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command and of the seg variant on the final round-4 tree.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for m in adapointr seg; do
  rm -rf /tmp/ks_$m
  (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$m -o ks -- \
      python3 $R/bench.py --model $m --steps 10 --warmup 5 --no-variants --no-cpu-baseline > $R/gpurun_out/ks_${m}_bench.log 2>&1)
  f=$(find /tmp/ks_$m -name 'ks_kernel_stats.csv' | head -n 1)
  head -n 60 "$f" > $R/gpurun_out/bench_${m}_kernel_stats_r4_v2.csv
  t=$(find /tmp/ks_$m -name 'ks_kernel_trace.csv' | head -n 1)
  python3 $R/benchmarks/trace_breakdown.py "$t" --steps 6 --top 12 > $R/gpurun_out/step_breakdown_${m}_r4_v3.txt || true
  echo "$m done"
done
