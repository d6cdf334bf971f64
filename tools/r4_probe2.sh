#!/bin/bash
# round 4, GPU call 4: encoder ingredient-removal probes (variant libraries), sampler in runs (A/B against one run), reverse-chain stamps
set -u
O=gpurun_out/r4d; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
for v in base fixedw noxp nokloop nogate fixedw_nogate_noxp; do
  if [ $v = base ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$v.so; fi
  timeout -k 10 120 python tools/enc_probe.py --mod p2_face > $O/enc_probe_$v.txt 2>&1; echo "probe $v rc=$?"; grep -E "fwd|bwd" $O/enc_probe_$v.txt
done
unset LFI_LIB_PATH
timeout -k 10 600 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_deep_parity.py::test_deep_flow_sampling_and_invert_against_oracle -q -m gpu -x -k "trajectory or pipeline_walk or inference or sampl or config3 or deep_flow_sampling" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for r in 1 4; do
  LFI_SAMPLE_RUNS=$r timeout -k 10 200 python bench.py --workload sample --cpu-baseline-seconds 0 > $O/bench_sample_runs$r.json 2> $O/bench_sample_runs$r.err; echo "sample runs=$r rc=$?"
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_sample_*.json
timeout -k 10 200 python tools/rev_stamps.py 1024 96 > $O/rev_stamps.txt 2>&1; echo "stamps rc=$?"; tail -40 $O/rev_stamps.txt
