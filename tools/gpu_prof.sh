#!/bin/bash
# usage: tools/gpu_prof.sh <tag> [bench args...]   -> gpurun_out/<tag>_stats.md (+ the rocpd db)
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python3 bench.py --no-gpu-state --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 "$@" > $OUT/${TAG}_prof.log 2>&1
grep '^{' $OUT/${TAG}_prof.log | tail -1 > $OUT/${TAG}_bench.json
python3 tools/rocpd_stats.py $OUT/${TAG}_prof/run_results.db 40 > $OUT/${TAG}_stats.md
cat $OUT/${TAG}_stats.md
