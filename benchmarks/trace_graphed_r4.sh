#!/bin/bash
# Kernel trace of the GRAPHED img / seg train step + coverage analysis (benchmarks/trace_breakdown.py --timeline).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for m in img seg; do
  rm -rf /tmp/prof_$m
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$m -o step -- \
      python3 $R/bench.py --model $m --graph on --steps 12 --warmup 4 --no-variants --no-cpu-baseline \
      > $R/gpurun_out/trace_graphed_${m}_bench.log 2>&1)
  f=$(find /tmp/prof_$m -name 'step_kernel_trace.csv' | head -n 1)
  python3 $R/benchmarks/trace_breakdown.py "$f" --steps 8 --timeline --top 12 > $R/gpurun_out/timeline_graphed_${m}_r4_v1.txt
  echo "$m done"
done
