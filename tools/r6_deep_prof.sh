#!/bin/bash
# round 6: the first kernel-level record of configs[4] (K=32 x L=3, T=512, B=128): rocprofv3 kernel stats, one step's timeline,
# PMC passes (HBM traffic, MFMA busy, LDS conflicts) of the same command, and the walk's phase stamps at this shape.
# usage: tools/r6_deep_prof.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-r6deep}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
NOBASE="--workload deep --quick --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 bench.py $NOBASE --steps 6 --warmup 2 > $O/prof.log 2>&1; echo "deep prof rc=$?"
python3 tools/rocpd_stats.py $O/prof/run_results.db 45 > $O/deep_kernel_stats.md 2>&1
python3 tools/step_timeline.py $O/prof/run_results.db > $O/deep_step_timeline.txt 2>&1
grep '^{' $O/prof.log | tail -1 > $O/deep_bench_under_rocprof.json
rm -rf $O/prof
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -- python3 bench.py $NOBASE --steps 2 --warmup 1 > $O/pmc_p$i.log 2>&1; echo "deep pmc $i rc=$?"
done
python3 tools/pmc_summary.py $O/pmc $O/deep_pmc_traffic.json > $O/deep_pmc_summary.md 2>&1
rm -rf $O/pmc
