#!/bin/bash
# round 5: timing-only ingredient-removal builds of enc_gru_fwd_t16_kernel (tools/build_variant.sh t16_<name> ...), p2_face shape
set -u
TAG=${1:-r5var}; shift || true
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$v.so; fi
  timeout -k 10 120 python tools/enc_probe.py --mod p2_face --fwd-only > $O/enc_probe_$v.txt 2>&1 || { echo "$v failed"; tail -3 $O/enc_probe_$v.txt; exit 1; }
  echo "== $v: $(grep 'fwd' $O/enc_probe_$v.txt | awk '{printf "%s ", $NF}')"
done
