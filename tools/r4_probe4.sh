#!/bin/bash
# round 4, GPU call 6: eight-wave encoder kernels (W8) - correctness, isolated times, whole-step A/B; sampler with the state-first reverse cell
set -u
O=gpurun_out/r4f; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tilings" > $O/pytest_w8.log 2>&1; echo "pytest w8 rc=$?"; tail -5 $O/pytest_w8.log
for m in p2_face p2_speech; do
  LFI_ENC_W8=1 timeout -k 10 120 python tools/enc_probe.py --mod $m > $O/enc_probe_${m}_w8.txt 2>&1; echo "probe $m w8 rc=$?"; grep -E "fwd|bwd" $O/enc_probe_${m}_w8.txt
done
for i in 1 2; do
  LFI_ENC_W8=0 LFI_ENC_R64=0 timeout -k 10 200 python bench.py --quick > $O/bench_wide_$i.json 2> $O/bench_wide_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_w8_$i.json 2> $O/bench_w8_$i.err
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_*.json
timeout -k 10 500 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_parity.py -q -m gpu -x -k "headline_config or k16_sampling or sampler_leaves or inference_matches or invert" > $O/pytest_misc.log 2>&1; echo "pytest misc rc=$?"; tail -3 $O/pytest_misc.log
for r in 1 4; do
  LFI_SAMPLE_RUNS=$r timeout -k 10 200 python bench.py --workload sample --cpu-baseline-seconds 0 > $O/bench_sample_runs$r.json 2> $O/bench_sample_runs$r.err; echo "sample runs=$r rc=$?"
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_sample_*.json
timeout -k 10 200 python tools/rev_stamps.py 1024 96 > $O/rev_stamps.txt 2>&1; echo "stamps rc=$?"; grep -v "^/opt" $O/rev_stamps.txt | head -40
