#!/bin/bash
# Builds A/B variants of liblfi_hip.so that differ only in lfi_encoder.hip macros: tools/build_enc_variant.sh name "-DX=1 ..." ...
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
make -C $ROOT/lets_face_it_amd/csrc -j4 >/dev/null
mkdir -p $ROOT/build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -I$ROOT/include $flags -c $ROOT/lets_face_it_amd/csrc/lfi_encoder.hip -o $ROOT/build/variants/enc_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/liblfi_$name.so $ROOT/build/variants/enc_$name.o \
     $ROOT/build/csrc/lfi_core.o $ROOT/build/csrc/lfi_gemm.o $ROOT/build/csrc/lfi_flow.o $ROOT/build/csrc/lfi_data.o
  echo built $name
done
