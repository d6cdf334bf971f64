#!/bin/bash
# round 3: the final tree's artefacts (everything DESIGN.md / README.md cite): profiles, PMC, the three bench workloads, the GPU
# test log with its parity report.   usage: tools/r3_final.sh <tag>
set -u
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
bash tools/r3_profiles.sh $TAG
timeout -k 10 500 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
timeout -k 10 300 python bench.py --workload sample > $O/bench_sample.json 2> $O/bench_sample.err; echo "sample rc=$?"
timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
