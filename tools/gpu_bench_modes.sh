#!/bin/bash
# Runs on the GPU box (via gpurun): whole-step bench in both GEMM precisions, sampling bench, rocprof kernel stats.
# usage: tools/gpu_bench_modes.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for P in f32 bf16x3; do
  timeout 600 python3 bench.py --steps 10 --warmup 3 --precision $P --cpu-baseline-seconds 0 > $OUT/${TAG}_train_$P.log 2>&1
  tail -1 $OUT/${TAG}_train_$P.log | cut -c1-400
done
for P in f32 bf16x3; do
  timeout 900 python3 bench.py --workload sample --steps 2 --warmup 1 --precision $P > $OUT/${TAG}_sample_$P.log 2>&1
  tail -1 $OUT/${TAG}_sample_$P.log | cut -c1-600
done
for P in f32 bf16x3; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_train_$P -o run -- python3 bench.py --no-gpu-state --steps 5 --warmup 2 --precision $P --cpu-baseline-seconds 0 > $OUT/${TAG}_prof_train_$P.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_sample -o run -- python3 bench.py --no-gpu-state --workload sample --steps 1 --warmup 1 --precision f32 > $OUT/${TAG}_prof_sample.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT | tail -20
