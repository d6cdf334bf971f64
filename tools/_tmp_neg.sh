timeout -k 10 600 python -m pytest tests/test_a_gpu_dp.py -x -q 2>&1 | tail -2
for r in 1 2 3; do for v in 1 0; do
  LFI_SYNC_OLD=$v python bench.py --steps 60 --warmup 8 --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('old=$v', round(d['ms_per_step'],3), d['final_loss'])"
done; done
