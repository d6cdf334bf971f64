#!/bin/bash
# one steady step's two-stream timeline only (the first part of tools/r3_profiles.sh).  usage: tools/timeline2.sh <tag> [env ...]
set -u
TAG=${1:-tl}; shift || true
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o run -- python3 bench.py $NOBASE --steps 12 > $O/prof2.log 2>&1; echo "prof2 rc=$?"
python3 tools/step_timeline.py $O/prof2/run_results.db > $O/step_timeline_2stream.txt 2>&1
rm -rf $O/prof2
