#!/bin/bash
# Build a library variant for same-box A/Bs:  tools/build_variant.sh <name> <file.hip>[,<file2.hip>...] [-Dmacro ...]
# -> build/var/liblfi_<name>.so = the tree's objects with the named files recompiled under the extra flags (picked with LFI_LIB_PATH).
set -eu
NAME=$1; SRCS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C $ROOT/lets_face_it_amd/csrc -j4 > /dev/null
mkdir -p $ROOT/build/var
for SRC in ${SRCS//,/ }; do
  FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$ROOT/include -Wall -Wno-unused-function"
  # (LFI_SLP=1: with the SLP vectoriser, which the Makefile switches off for this file - the determinism chase of round 5)
  [ "$SRC" = "lfi_encoder.hip" ] && [ -z "${LFI_SLP:-}" ] && FLAGS="$FLAGS -fno-slp-vectorize"
  hipcc $FLAGS "$@" -c $ROOT/lets_face_it_amd/csrc/$SRC -o $ROOT/build/var/${NAME}_${SRC%.hip}.o
done
OBJS=""
for f in lfi_core lfi_gemm lfi_pgemm lfi_encoder lfi_flow lfi_data lfi_sample lfi_wgrad; do
  case ",$SRCS," in
    *",$f.hip,"*) OBJS="$OBJS $ROOT/build/var/${NAME}_$f.o" ;;
    *) OBJS="$OBJS $ROOT/build/csrc/$f.o" ;;
  esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/var/liblfi_$NAME.so $OBJS
echo "built build/var/liblfi_$NAME.so"
