#!/bin/bash
# round 3: the final tree's artefacts (everything DESIGN.md / README.md cite): profiles, PMC, the three bench workloads, the GPU
# test log with its parity report, the smoke test, the 2-rank rehearsal of bench.py --gpus 2.   usage: tools/r3_final.sh <tag>
set -u
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
bash tools/r3_profiles.sh $TAG
timeout -k 10 500 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
timeout -k 10 300 python bench.py --workload sample > $O/bench_sample.json 2> $O/bench_sample.err; echo "sample rc=$?"
timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
LFI_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 gloo rc=$?"
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1; echo "stamps rc=$?"
