#!/bin/bash
set -u
O=gpurun_out/c10; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "pipeline_walk or abandoned or train_forward_backward or fused_training or headline_config or planes_chain or graphed or config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 40 > $O/bench_pf.json 2> $O/bench_pf.err; echo "bench rc=$?"
LFI_PIPE_PREFETCH=0 timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 40 > $O/bench_nopf.json 2> $O/bench_nopf.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 40 > $O/bench_pf2.json 2> $O/bench_pf2.err; echo "bench rc=$?"
timeout -k 10 200 python tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1; tail -12 $O/pipe_stamps.txt
