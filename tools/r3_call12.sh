#!/bin/bash
set -u
O=gpurun_out/c12; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"
tail -4 $O/pytest_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "headline_config or planes_chain or full_model_k16 or graphed or strong_scaling or bf16x3 or fused_training or every_shipped" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
tail -4 $O/pytest_parity.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0 --steps 40 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0 --steps 40 > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc=$?"
