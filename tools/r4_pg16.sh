#!/bin/bash
# same-box A/B: planes GEMMs on v_mfma_f32_32x32x16_bf16 (tree) against the timing-only 16x16x32 build (garbage results)
set -u
O=gpurun_out/${1:-r4pg}; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --quick > $O/bench_base_$i.json 2> $O/bench_base_$i.err
  LFI_LIB_PATH=build/var/liblfi_pg16.so timeout -k 10 200 python bench.py --quick > $O/bench_pg16_$i.json 2> $O/bench_pg16_$i.err
done
python3 - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    kt = d.get("kernel_timing", {})
    print(f.split("/")[-1], "%.3f ms" % d["ms_per_step"], " ".join("%s %.4f" % (k.replace("gemm_", ""), v["ms"]) for k, v in kt.items() if k.startswith("gemm")))
PY
