#!/bin/bash
# round 3, GPU call 9: bf16 gradient stash of the encoders' BPTT (two-product mode): unit + parity tests, bench
set -u
O=gpurun_out/c9; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "a_bf16 or gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"
tail -4 $O/pytest_gemm.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_a_gpu_dp.py -x -q -m gpu -k "headline_config or planes_chain or full_model_k16 or graphed or strong_scaling or full_size or dp or rccl or pipeline_walk" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
tail -4 $O/pytest_parity.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o run -- python3 bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 12 > $O/prof2.log 2>&1; echo "prof2 rc=$?"
python3 tools/step_timeline.py $O/prof2/run_results.db > $O/step_timeline_2stream.txt 2>&1
rm -rf $O/prof2
