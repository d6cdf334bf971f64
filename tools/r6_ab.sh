#!/bin/bash
# same-box A/B of environment switches, arms alternating:  tools/r6_ab.sh <outdir> <runs> "<VAR=value ...>" ["<VAR=value ...>" ...]
# (an arm given as "-" is the default environment). Prints ms_per_step of every run, arm by arm.
set -u
O=gpurun_out/$1; mkdir -p $O
N=$2; shift 2
export TMPDIR=/tmp
for i in $(seq 1 $N); do
  a=0
  for arm in "$@"; do
    a=$((a+1))
    if [ "$arm" = "-" ]; then
      timeout -k 10 200 python bench.py --quick > $O/bench_arm${a}_$i.json 2> $O/bench_arm${a}_$i.err
    else
      env $arm timeout -k 10 200 python bench.py --quick > $O/bench_arm${a}_$i.json 2> $O/bench_arm${a}_$i.err
    fi
  done
done
python3 - "$O" "$@" <<'PY'
import json, glob, sys
arms = sys.argv[2:]
for a, arm in enumerate(arms, 1):
    ms = []
    for f in sorted(glob.glob(sys.argv[1] + "/bench_arm%d_*.json" % a)):
        try:
            ms.append(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"])
        except Exception as e:
            ms.append(float("nan"))
    print("arm %d [%s]: %s  mean %.3f" % (a, arm, " ".join("%.3f" % m for m in ms), sum(ms) / max(len(ms), 1)))
PY
