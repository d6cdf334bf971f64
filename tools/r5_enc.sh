#!/bin/bash
# round 5: forward window-encoder kernels A/B by LFI_ENC_T16 mode (0 = round 4's 64-window kernel, 2 = epilogue under the matrix
# phase): the bitwise tiling test, both big encoders alone through the C ABI
# (tools/enc_probe.py), then the whole step.   usage: tools/r5_enc.sh <tag> [modes...]
set -u
TAG=${1:-r5enc}; shift || true
MODES=${*:-0 2 0 2}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "window_encoder_tilings" > $O/pytest_tilings.log 2>&1; rc=$?
echo "tilings rc=$rc"; tail -5 $O/pytest_tilings.log
[ $rc -ne 0 ] && exit $rc
for m in p2_face p2_speech; do for t in $MODES; do
  LFI_ENC_T16=$t timeout -k 10 120 python tools/enc_probe.py --mod $m --fwd-only > $O/enc_probe_${m}_t16_$t.txt 2>&1 || exit 1
  echo "== $m T16=$t: $(grep 'fwd' $O/enc_probe_${m}_t16_$t.txt | awk '{printf "%s ", $NF}')"
done; done
for t in $MODES; do
  LFI_ENC_T16=$t timeout -k 10 200 python bench.py --quick > $O/bench_t16_$t.json 2> $O/bench_t16_$t.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/bench_t16_$t.json").read().strip().splitlines()[-1])
print("T16=$t ms_per_step", d["ms_per_step"])
PY
done
