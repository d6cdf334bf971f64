#!/bin/bash
# round 5: PMC passes (one counter group per pass) over one-run sampling calls: the per-frame conditioning kernel and the reverse chain.
# usage: tools/pmc_sample.sh <tag>
set -u
TAG=${1:-r5pmcsample}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG}; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  LFI_SAMPLE_RUNS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --workload sample --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --steps 1 --warmup 2 > $OUT/p$i.log 2>&1; echo "pmc $i rc=$?"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.md
rm -rf $OUT/p[0-9]*
grep -E "^\| kernel|sc_cond|rev_chain" $OUT/summary.md | cut -c1-400
