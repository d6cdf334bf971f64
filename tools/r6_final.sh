#!/bin/bash
# round 6: the final tree's artefacts (everything DESIGN.md section 11 / README.md cite).   usage: tools/r6_final.sh <tag> [part ...]
# parts: bench (the default line with its sub-records, sample, deep, bench.py --gpus 2 started as the driver types it, on gloo),
# prof (kernel stats, timelines, PMC of the training step; kernel stats of a sampling call; configs[4]'s kernel stats, timeline, PMC),
# tests (smoke, pytest -m gpu with the parity report, the same suite with every kernel switch of rounds 4 - 6 off); default: all
set -u
TAG=${1:-r6final}; shift || true
PARTS=${*:-bench prof tests}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout -k 10 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
  timeout -k 10 300 python bench.py --workload sample > $O/bench_sample.json 2> $O/bench_sample.err; echo "sample rc=$?"
  timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
  LFI_BENCH_DP_GRAPH=1 LFI_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --graph-steps 5 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 gloo (self-launched) rc=$?"
  bash tools/r6_ab.sh $TAG/ab_thin 2 "LFI_FLOW_WGRAD_FUSED=0" "-" > $O/ab_thin.txt 2>&1; cat $O/ab_thin.txt
fi
if has prof; then
  bash tools/r3_profiles.sh $TAG
  NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/profs -o run -- python3 bench.py --workload sample $NOBASE --steps 3 --warmup 2 > $O/profs.log 2>&1; echo "prof sample rc=$?"
  python3 tools/rocpd_stats.py $O/profs/run_results.db 16 > $O/sampler_kernel_stats.md 2>&1
  rm -rf $O/profs
  bash tools/r6_deep_prof.sh $TAG
fi
if has tests; then
  timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  export LFI_PARITY_REPORT=$O/parity.txt
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
  unset LFI_PARITY_REPORT
  LFI_PGEMM_16=0 LFI_PGEMM_16T=0 LFI_PGEMM_DIRECT=0 LFI_ENC_M16=0 LFI_ENC_T16=0 LFI_ENC_SCATTER16=0 LFI_SAMPLE_FUSED=0 LFI_FLOW_G16=0 LFI_SAMPLE_XF_CHAIN=0 LFI_SAMPLE_STATIC_CUS=0 LFI_SAMPLE_WFRAG16=0 LFI_FLOW_WGRAD_FUSED=0 \
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_switches_off.log 2>&1; echo "pytest (switches off) rc=$?"; tail -3 $O/pytest_gpu_switches_off.log
fi
