#!/bin/bash
# round 3, GPU call 5: auto 2-product backward + hipGraph step: new tests, bench, then the whole GPU suite
set -u
O=gpurun_out/c5; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 600 python -m pytest tests/test_gpu_headline_parity.py -x -q -m gpu -k "graphed or headline_config" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"
tail -5 $O/pytest_new.log
timeout -k 10 400 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"
LFI_STEP_GRAPH=0 timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 > $O/bench_nograph.json 2> $O/bench_nograph.err; echo "nograph rc=$?"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -5 $O/pytest_all.log
