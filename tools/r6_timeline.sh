#!/bin/bash
# one steady step's kernel timeline + kernel stats under rocprofv3 (two streams, as the step runs).
# usage: tools/r6_timeline.sh <outdir> <name> [bench.py args ...]     environment switches are inherited
set -u
O=gpurun_out/$1; mkdir -p $O
NAME=$2; shift 2
export TMPDIR=/tmp
NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$NAME -o run -- python3 bench.py $NOBASE --steps 12 "$@" > $O/prof_$NAME.log 2>&1; echo "prof $NAME rc=$?"
python3 tools/rocpd_stats.py $O/prof_$NAME/run_results.db 45 > $O/kernel_stats_$NAME.md 2>&1
python3 tools/step_timeline.py $O/prof_$NAME/run_results.db > $O/step_timeline_$NAME.txt 2>&1
grep '^{' $O/prof_$NAME.log | tail -1 > $O/bench_under_rocprof_$NAME.json
rm -rf $O/prof_$NAME
