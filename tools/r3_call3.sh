#!/bin/bash
# round 3, GPU call 3: generalized planes GEMM unit tests + sampling precision split (chain only) + bench sanity
set -u
O=gpurun_out/c3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "planes or column_sums or gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gemm.log
LFI_PIPE_X3=0 timeout -k 10 300 python tools/sample_precision_probe.py > $O/sample_precision_chain_f32.md 2> $O/sample_precision.err; echo "probe rc=$?"
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --strong-anchor-batch 0 > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"
