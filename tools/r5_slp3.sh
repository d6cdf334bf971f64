#!/bin/bash
# round 5, the SLP chase (csrc/Makefile): the accumulator-layout fused BPTT kernel (enc_gru_bwd_fused_kernel<true>, forced with
# LFI_ENC_WIDE_BWD=0) in three builds - the tree's (no SLP), SLP on (16 spilled VGPRs), SLP on with launch_bounds(256, 1) (no spills).
#   LFI_SLP=1 tools/build_variant.sh slp lfi_encoder.hip;  LFI_SLP=1 tools/build_variant.sh slp_occ1 lfi_encoder.hip -DLFI_ENC_FUSED_BWD_OCC=1
set -u
O=gpurun_out/${1:-r5slp3}; mkdir -p $O
for lib in tree slp slp_occ1; do
  if [ $lib = tree ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$lib.so; fi
  rm -f $O/determinism_fusedbwd_$lib.txt
  for b in 256 40; do
    LFI_ENC_WIDE=0 LFI_ENC_WIDE_BWD=0 timeout -k 10 120 python tools/determinism_probe.py bf16x3 $b eval 4 2>&1 | grep -v amdgpu.ids | tail -14 >> $O/determinism_fusedbwd_$lib.txt
  done
  echo "== $lib"; cat $O/determinism_fusedbwd_$lib.txt
done
