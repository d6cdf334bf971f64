#!/bin/bash
# round 3: every artefact the DESIGN / bench line cite, from ONE tree.  usage: tools/r3_profiles.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-r3}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
NOBASE="--no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
# 1. kernel statistics + one steady step's timeline, two streams (as the step runs) and one stream (every duration its own)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o run -- python3 bench.py $NOBASE --steps 12 > $O/prof2.log 2>&1; echo "prof2 rc=$?"
python3 tools/rocpd_stats.py $O/prof2/run_results.db 45 > $O/kernel_stats.md 2>&1
python3 tools/step_timeline.py $O/prof2/run_results.db > $O/step_timeline_2stream.txt 2>&1
grep '^{' $O/prof2.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/prof2
LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o run -- python3 bench.py $NOBASE --steps 8 > $O/prof1.log 2>&1; echo "prof1 rc=$?"
python3 tools/step_timeline.py $O/prof1/run_results.db > $O/step_timeline_1stream.txt 2>&1
rm -rf $O/prof1
# 2. PMC passes (HBM traffic, L2 hit, MFMA busy, LDS conflicts, effective clock), one counter group per pass
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -- python3 bench.py $NOBASE --steps 2 --warmup 2 > $O/pmc_p$i.log 2>&1; echo "pmc $i rc=$?"
done
python3 tools/pmc_summary.py $O/pmc $O/pmc_traffic_bf16x3.json > $O/pmc_summary.md
rm -rf $O/pmc
