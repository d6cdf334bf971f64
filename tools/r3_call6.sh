#!/bin/bash
# round 3, GPU call 6: tile variant + A-lo elision + hi-only planes: unit tests, parity, bench A/B, 1-stream timeline
set -u
O=gpurun_out/c6; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "planes" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"
tail -4 $O/pytest_gemm.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "headline_config or planes_chain or bf16x3 or full_model_k16 or graphed" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
tail -4 $O/pytest_parity.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
export TMPDIR=/tmp
LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --steps 8 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 tools/step_timeline.py $O/prof/run_results.db > $O/step_timeline_1stream.txt 2>&1; echo "timeline rc=$?"
rm -rf $O/prof
