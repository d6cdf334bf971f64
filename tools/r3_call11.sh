#!/bin/bash
set -u
O=gpurun_out/c11; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -4 $O/pytest_all.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 40 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
