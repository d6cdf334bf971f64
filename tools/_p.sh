timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sampl or inference or generate" 2>&1 | tail -3 || exit 1
for lib in new prev new prev; do
  if [ $lib = prev ]; then export LFI_LIB_PATH=$PWD/build/var/liblfi_prev.so; else unset LFI_LIB_PATH; fi
  timeout -k 10 300 python bench.py --workload sample --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib:', round(d['ms_per_step'],2), 'ms; static in front', d['kernel_timing'].get('sample_static',{}).get('ms'), 'rest', d['kernel_timing'].get('sample_graph',{}).get('ms'))"
  LFI_SAMPLE_RUNS=1 timeout -k 10 300 python bench.py --workload sample --no-gpu-state --no-more-workloads --cpu-baseline-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib (one run):', round(d['ms_per_step'],2), 'ms; static', d['kernel_timing'].get('sample_static',{}).get('ms'), 'chain', d['kernel_timing'].get('sample_graph',{}).get('ms'))"
done
