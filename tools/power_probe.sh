#!/bin/bash
# Socket power and shader clock sampled while bench.py's training step runs (is the step power-limited?): tools/power_probe.sh
python bench.py --steps 1500 --warmup 10 --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 > /dev/null 2>&1 &
pid=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '
  echo
  sleep 1
done
wait $pid
