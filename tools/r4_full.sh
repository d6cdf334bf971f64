#!/bin/bash
# round 4: the whole GPU test suite with per-test durations.  usage: tools/r4_full.sh <tag>
set -u
O=gpurun_out/${1:-r4k}; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
