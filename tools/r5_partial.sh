#!/bin/bash
# round 5: the sampler's static part on a stream that owns part of every XCD (LFI_SAMPLE_STATIC_CUS = CUs per XCD) beside the chain
set -u
O=gpurun_out/${1:-r5partial}; mkdir -p $O
one() {  # cus [runs]  (no runs: the engine's default)
  local runs=${2:-}
  env LFI_SAMPLE_STATIC_CUS=$1 ${runs:+LFI_SAMPLE_RUNS=$runs} timeout -k 10 300 python bench.py --workload sample --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 > $O/out.json 2>$O/err.txt; rc=$?
  python -c "
import sys,json
d=json.loads(open('$O/out.json').read().strip().splitlines()[-1]); print('rc $rc static CUs/XCD $1, runs ${runs:-default}:', round(d['ms_per_step'],2), 'ms; static in front', d['kernel_timing'].get('sample_static',{}).get('ms'), 'rest', d['kernel_timing'].get('sample_graph',{}).get('ms'))" || { tail -5 $O/err.txt; exit 1; }
}
shift
for cfg in "$@"; do one ${cfg%%:*} ${cfg##*:} || exit 1; done
