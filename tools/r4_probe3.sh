#!/bin/bash
# round 4, GPU call 5: two-row-tile encoder kernels (R64) - correctness, isolated times, whole-step A/B; kernel trace of the sampler
set -u
O=gpurun_out/r4e; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "two_row_tile" > $O/pytest_r64.log 2>&1; echo "pytest r64 rc=$?"; tail -3 $O/pytest_r64.log
for m in p2_face p2_speech; do
  for r in 1 0; do
    LFI_ENC_R64=$r timeout -k 10 120 python tools/enc_probe.py --mod $m > $O/enc_probe_${m}_r64_$r.txt 2>&1; echo "probe $m r64=$r rc=$?"; grep -E "fwd|bwd" $O/enc_probe_${m}_r64_$r.txt
  done
done
for i in 1 2; do
  LFI_ENC_R64=0 timeout -k 10 200 python bench.py --quick > $O/bench_r64off_$i.json 2> $O/bench_r64off_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_r64on_$i.json 2> $O/bench_r64on_$i.err
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_*.json
timeout -k 10 400 python -m pytest tests/test_gpu_headline_parity.py -q -m gpu -x -k "headline_config or strong_scaling" > $O/pytest_headline.log 2>&1; echo "pytest headline rc=$?"; tail -3 $O/pytest_headline.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_sample -o run -- python3 bench.py --workload sample --steps 3 --cpu-baseline-seconds 0 > $O/prof_sample.log 2>&1; echo "prof rc=$?"
python3 tools/rocpd_stats.py $O/prof_sample/run_results.db 30 > $O/sample_kernel_stats.md 2>&1; rm -rf $O/prof_sample; head -45 $O/sample_kernel_stats.md
