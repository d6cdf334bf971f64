#!/bin/bash
# round 5, VERDICT r4 next #8: is lfi_encoder.hip built WITH the SLP vectoriser (build/var/liblfi_slp.so: LFI_SLP=1 tools/build_variant.sh slp
# lfi_encoder.hip) still not reproducible run to run, and in which kernel? Headline and ragged batch, eval and train (dropout masks),
# default kernels and the accumulator-layout fused ones (LFI_ENC_WIDE=0), 6 passes each.
set -u
O=gpurun_out/${1:-r5slp}; mkdir -p $O
export TMPDIR=/tmp
for lib in tree slp; do
  if [ $lib = tree ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_slp.so; fi
  for wide in 1 0; do for b in 256 40; do for mode in eval train; do
    LFI_ENC_WIDE=$wide timeout -k 10 120 python tools/determinism_probe.py bf16x3 $b $mode 6 2>&1 | grep -v amdgpu.ids >> $O/determinism_$lib.txt || exit 1
  done; done; done
  echo "== $lib"; cat $O/determinism_$lib.txt
done
