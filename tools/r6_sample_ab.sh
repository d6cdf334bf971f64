#!/bin/bash
# same-box A/B of environment switches on the sampling workload (configs[3]):  tools/r6_sample_ab.sh <outdir> <runs> "<VAR=v ...>" ...
set -u
O=gpurun_out/$1; mkdir -p $O
N=$2; shift 2
export TMPDIR=/tmp
for i in $(seq 1 $N); do
  a=0
  for arm in "$@"; do
    a=$((a+1))
    if [ "$arm" = "-" ]; then
      timeout -k 10 200 python bench.py --workload sample --quick --no-gpu-state > $O/sample_arm${a}_$i.json 2> $O/sample_arm${a}_$i.err
    else
      env $arm timeout -k 10 200 python bench.py --workload sample --quick --no-gpu-state > $O/sample_arm${a}_$i.json 2> $O/sample_arm${a}_$i.err
    fi
  done
done
python3 - "$O" "$@" <<'PY'
import json, glob, sys
arms = sys.argv[2:]
for a, arm in enumerate(arms, 1):
    ms, kt = [], []
    for f in sorted(glob.glob(sys.argv[1] + "/sample_arm%d_*.json" % a)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            ms.append(d["ms_per_step"]); kt.append(d["roofline"]["ms_per_launch"])
        except Exception as e:
            ms.append(float("nan"))
    print("arm %d [%s]: call ms %s  mean %.3f ; chain (per-frame graphs) ms %s" % (a, arm, " ".join("%.3f" % m for m in ms), sum(ms) / max(len(ms), 1), " ".join("%.3f" % m for m in kt)))
PY
