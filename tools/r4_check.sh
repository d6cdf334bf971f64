#!/bin/bash
set -u
O=gpurun_out/${1:-r4m}; mkdir -p $O
export TMPDIR=/tmp
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout -k 10 300 python tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1; echo "stamps rc=$?"; grep -v "^/opt" $O/pipe_stamps.txt
for i in 1 2 3; do
  LFI_ENC_FWD_SMALL_ON_SIDE=0 timeout -k 10 200 python bench.py --quick > $O/bench_fwdinline_$i.json 2> $O/bench_fwdinline_$i.err
  timeout -k 10 200 python bench.py --quick > $O/bench_default_$i.json 2> $O/bench_default_$i.err
done
grep -o "\"ms_per_step\": [0-9.]*" $O/bench_*.json
LFI_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_lds -- python3 bench.py --quick --steps 2 --warmup 2 > $O/pmc_lds.log 2>&1; echo "pmc rc=$?"
python3 - <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/%s/pmc_lds/**/*counter_collection.csv" % (sys.argv[1] if len(sys.argv) > 1 else "r4m"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:12]:
    a = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-62s conflict %5.1f %% of LDS cycles" % (k, 100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0) / a if a else 0.0))
PY
rm -rf $O/pmc_lds
