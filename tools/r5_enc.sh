#!/bin/bash
# round 5: the window-encoder forward kernel with the epilogue under the matrix phase (LFI_ENC_T16) against round 4's (LFI_ENC_T16=0):
# the bitwise tiling test, then both kernels alone through the C ABI (tools/enc_probe.py), then the whole step.   usage: tools/r5_enc.sh <tag>
set -u
TAG=${1:-r5enc}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "window_encoder_tilings" > $O/pytest_tilings.log 2>&1; rc=$?
echo "tilings rc=$rc"; tail -5 $O/pytest_tilings.log
[ $rc -ne 0 ] && exit $rc
for m in p2_face p2_speech; do for t in 1 0 1 0; do
  LFI_ENC_T16=$t timeout -k 10 120 python tools/enc_probe.py --mod $m > $O/enc_probe_${m}_t16_$t.txt 2>&1 || exit 1
  echo "== $m T16=$t"; grep -i "fwd" $O/enc_probe_${m}_t16_$t.txt | head -4
done; done
for t in 1 0 1 0; do
  LFI_ENC_T16=$t timeout -k 10 200 python bench.py --quick > $O/bench_t16_$t.json 2> $O/bench_t16_$t.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/bench_t16_$t.json").read().strip().splitlines()[-1])
print("T16=$t ms_per_step", d["ms_per_step"])
PY
done
