#!/bin/bash
# Ingredient-removal timing of gemm_planes_256_kernel: tools/build_variants.sh builds, this runs the probe once per variant.
for v in base nomma nodma noread noepi nodmaepi; do
  if [ $v = base ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/variants/liblfi_$v.so; fi
  echo "== $v"
  timeout -k 10 120 python tools/gemm_probe.py --reps 20 2>&1 | grep "planes\|on planes"
done
