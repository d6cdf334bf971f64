#!/bin/bash
# round 5: the sampler's per-frame sequence with the window split and the chain-state reset inside the conditioning kernel
# (default) against round 4's three extra nodes per frame (LFI_SAMPLE_XFRAG=1 keeps the fragment kernel; the memset went with the fusion)
set -u
O=gpurun_out/${1:-r5sample}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sampl or inference or generate" > $O/pytest_sampling.log 2>&1; rc=$?
echo "sampling tests rc=$rc"; tail -4 $O/pytest_sampling.log
[ $rc -ne 0 ] && exit $rc
for x in 1 0 1 0; do   # (1 = the default: the reverse chain leaves the next frame's window fragments; 0 = a fragment launch per frame)
  LFI_SAMPLE_XF_CHAIN=$x timeout -k 10 300 python bench.py --workload sample > $O/bench_sample_xfchain_$x.json 2> $O/bench_sample_xfchain_$x.err || exit 1
  python -c "
import json
d=json.loads(open('$O/bench_sample_xfchain_$x.json').read().strip().splitlines()[-1]); print('XF_CHAIN=$x', round(d['ms_per_step'],2), 'ms', d['kernel_timing'].get('sample_graph'))"
done
