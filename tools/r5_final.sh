#!/bin/bash
# round 5: the final tree's artefacts (everything DESIGN.md section 10 / README.md cite).   usage: tools/r5_final.sh <tag> [part ...]
# parts: bench (the default line with its sub-records, sample, deep, bench.py --gpus 2 started as the driver types it, on gloo),
# prof (kernel stats, timelines, PMC of the training step; kernel stats of a sampling call), probes (encoder forward kernels by
# LFI_ENC_T16 mode, walk / reverse-chain stamps), tests (smoke, pytest -m gpu with the parity report, the same suite with every
# round-4 / round-5 kernel switch off); default: all
set -u
TAG=${1:-r5final}; shift || true
PARTS=${*:-bench prof probes tests}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout -k 10 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
  timeout -k 10 300 python bench.py --workload sample > $O/bench_sample.json 2> $O/bench_sample.err; echo "sample rc=$?"
  timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
  LFI_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 gloo (self-launched) rc=$?"
fi
if has prof; then
  bash tools/r3_profiles.sh $TAG
  NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/profs -o run -- python3 bench.py --workload sample $NOBASE --steps 3 --warmup 2 > $O/profs.log 2>&1; echo "prof sample rc=$?"
  python3 tools/rocpd_stats.py $O/profs/run_results.db 16 > $O/sampler_kernel_stats.md 2>&1
  rm -rf $O/profs
fi
if has probes; then
  timeout -k 10 300 python tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1; echo "pipe stamps rc=$?"
  for m in p2_face p2_speech; do for t in 0 2 0 2; do
    LFI_ENC_T16=$t timeout -k 10 120 python tools/enc_probe.py --mod $m >> $O/enc_probe_${m}_t16_$t.txt 2>&1
  done; done
  timeout -k 10 120 python tools/coresidency_probe.py > $O/coresidency.txt 2>&1
fi
if has tests; then
  timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  export LFI_PARITY_REPORT=$O/parity.txt
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
  unset LFI_PARITY_REPORT
  LFI_PGEMM_16=0 LFI_PGEMM_16T=0 LFI_PGEMM_DIRECT=0 LFI_ENC_M16=0 LFI_ENC_T16=0 LFI_ENC_SCATTER16=0 LFI_SAMPLE_FUSED=0 LFI_FLOW_G16=0 LFI_SAMPLE_XF_CHAIN=0 LFI_SAMPLE_STATIC_CUS=0 LFI_SAMPLE_WFRAG16=0 \
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_switches_off.log 2>&1; echo "pytest (switches off) rc=$?"; tail -3 $O/pytest_gpu_switches_off.log
fi
