#!/bin/bash
# round 6: which leg of tools/dp_gloo_check.py --graph moves when the eager / graph comparison fails? (checksums of both legs, N runs)
O=gpurun_out/${1:-r6dp}; mkdir -p $O
for i in $(seq 1 ${2:-12}); do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+i)) tools/dp_gloo_check.py --graph > $O/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep 'rank 0 checksums' $O/run_$i.log)"
done
