#!/bin/bash
# round 3, GPU call 4: the planes chain end to end: parity (fixtures + headline), A/B timing, timeline
set -u
O=gpurun_out/c4; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "not config4 and not config3 and not strong_scaling" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_chain.json 2> $O/bench_chain.err; echo "chain rc=$?"
LFI_PCHAIN=0 timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_nochain.json 2> $O/bench_nochain.err; echo "nochain rc=$?"
timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_chain2.json 2> $O/bench_chain2.err; echo "chain2 rc=$?"
export TMPDIR=/tmp
LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 8 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 tools/step_timeline.py $O/prof/run_results.db > $O/step_timeline_1stream.txt 2>&1; echo "timeline rc=$?"
python3 tools/rocpd_stats.py $O/prof/run_results.db 40 > $O/kernel_stats.md 2>&1
rm -rf $O/prof
