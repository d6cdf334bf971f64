#!/bin/bash
# same-box A/B of one environment switch:  tools/r4_ab_env.sh <outdir> <VAR=value for arm A> [runs]   (arm B: the default)
set -u
O=gpurun_out/$1; mkdir -p $O
A=$2; N=${3:-3}
export TMPDIR=/tmp
for i in $(seq 1 $N); do
  env $A timeout -k 10 200 python bench.py --quick > $O/bench_A_$i.json 2> $O/bench_A_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_B_$i.json 2> $O/bench_B_$i.err
done
python3 - "$O" "$A" <<'PY'
import json, glob, sys
print("arm A:", sys.argv[2], "  arm B: default")
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    kt = d.get("kernel_timing", {})
    print(f.split("/")[-1], "%.3f ms" % d["ms_per_step"], " ".join("%s %.4f" % (k.replace("gemm_", ""), v["ms"]) for k, v in kt.items() if k.startswith("gemm")))
PY
