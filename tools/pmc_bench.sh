#!/bin/bash
# HBM traffic and pipe-utilisation counters of bench.py's kernels. Each counter group gets its own rocprofv3 pass
# (--kernel-trace + --pmc only; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  usage: tools/pmc_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --steps 2 --warmup 2 "$@" > $OUT/p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT $OUT/traffic.json > $OUT/summary.md
find $OUT -name "*.csv" -size +8M -delete
head -60 $OUT/summary.md
