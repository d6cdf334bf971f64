#!/bin/bash
# round 4: A/B of the smallest encoder's BPTT on the second stream, then the whole GPU test suite with per-test durations.  usage: tools/r4_full.sh <tag>
set -u
O=gpurun_out/${1:-r4l}; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
for i in 1 2 3; do
  LFI_ENC_BWD_SMALL_ON_SIDE=0 timeout -k 10 200 python bench.py --quick > $O/bench_inline_$i.json 2> $O/bench_inline_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_side_$i.json 2> $O/bench_side_$i.err
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_*.json
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
