#!/bin/bash
# round 5: the sampler's static part (window encoders + the static cond_transform columns) seen alone: ONE run, kernel trace
set -u
O=gpurun_out/${1:-r5static}; mkdir -p $O
export TMPDIR=/tmp
NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0"
LFI_SAMPLE_RUNS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/profs -o run -- python3 bench.py --workload sample $NOBASE --steps 3 --warmup 2 > $O/profs.log 2>&1; echo "prof rc=$?"
python3 tools/rocpd_stats.py $O/profs/run_results.db 24 > $O/sampler_one_run_kernel_stats.md 2>&1
rm -rf $O/profs
cut -c1-200 $O/sampler_one_run_kernel_stats.md
