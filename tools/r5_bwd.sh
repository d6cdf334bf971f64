#!/bin/bash
# round 5: the 64-window BPTT kernel with the third gate image (no d n * r register copy, three barriers) and pipelined stash batches,
# against round 4's form (build/var/liblfi_bwd_r4.so: -DLFI_ENC_BWD_NOPIPE -DLFI_ENC_BWD_NOZ) and the image change alone (liblfi_bwd_z.so)
set -u
O=gpurun_out/${1:-r5bwd}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "window_encoder_tilings" > $O/pytest_tilings.log 2>&1; rc=$?
echo "tilings rc=$rc"; tail -3 $O/pytest_tilings.log
[ $rc -ne 0 ] && exit $rc
for lib in tree bwd_r4 bwd_z tree bwd_r4; do
  if [ $lib = tree ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$lib.so; fi
  for m in p2_face p2_speech; do
    timeout -k 10 120 python tools/enc_probe.py --mod $m > $O/enc_probe_${m}_$lib.txt 2>&1 || exit 1
    echo "== $lib $m: $(grep 'bwd 2 products, fp16' $O/enc_probe_${m}_$lib.txt | awk '{print $NF}')"
  done
done
for lib in tree bwd_r4 tree bwd_r4; do
  if [ $lib = tree ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$lib.so; fi
  timeout -k 10 200 python bench.py --quick > $O/bench_$lib.json 2> $O/bench_$lib.err || exit 1
  python -c "
import json
d=json.loads(open('$O/bench_$lib.json').read().strip().splitlines()[-1]); print('$lib ms_per_step', round(d['ms_per_step'],3))"
done
