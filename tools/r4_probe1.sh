#!/bin/bash
# round 4, GPU call 3: encoder ingredient-removal probes, fp16-stash A/B of the whole step, the new parity tests
set -u
O=gpurun_out/r4c; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
for v in base fixedw noxp fixedw_noxp; do
  if [ $v = base ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$v.so; fi
  timeout -k 10 120 python tools/enc_probe.py --mod p2_face > $O/enc_probe_$v.txt 2>&1; echo "probe $v rc=$?"
done
unset LFI_LIB_PATH
timeout -k 10 120 python tools/enc_probe.py --mod p2_speech > $O/enc_probe_base_p2_speech.txt 2>&1
for i in 1 2; do
  LFI_ENC_STASH_F16=0 timeout -k 10 200 python bench.py --quick > $O/bench_f32stash_$i.json 2> $O/bench_f32stash_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_f16stash_$i.json 2> $O/bench_f16stash_$i.err
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_*.json
timeout -k 10 700 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_deep_parity.py::test_deep_flow_sampling_and_invert_against_oracle -q -m gpu -k "trajectory or invert or pipeline_walk or full_size or config4 or mimic or deep" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
