#!/bin/bash
# round 5: PMC passes (one counter group per pass) over tools/enc_probe.py --fwd-only for the two forward window-encoder kernels:
# LFI_ENC_T16=0 (round 4's 64-window kernel) and =2 (epilogue under the matrix phase).  usage: tools/pmc_enc.sh <tag>
set -u
TAG=${1:-r5pmcenc}
export TMPDIR=/tmp
for t in 0 2; do
  OUT=$PWD/gpurun_out/${TAG}/t16_$t; mkdir -p $OUT
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    LFI_ENC_T16=$t timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/enc_probe.py --mod p2_face --fwd-only --reps 4 > $OUT/p$i.log 2>&1; echo "t16=$t pmc $i rc=$?"
  done
  python3 tools/pmc_summary.py $OUT > $OUT/summary.md
  rm -rf $OUT/p[0-9]*
  grep -E "^\| kernel|enc_gru_fwd" $OUT/summary.md | cut -c1-260
done
