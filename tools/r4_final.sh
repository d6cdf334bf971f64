#!/bin/bash
# round 4: the final tree's artefacts (everything DESIGN.md / README.md cite).   usage: tools/r4_final.sh <tag> [part ...]
# parts: prof (kernel stats, timelines, PMC), bench (the default line with its sub-records, sample, deep), probes (stamps, encoder
# tilings, planes-GEMM tile order), tests (pytest -m gpu with the parity report, smoke, 2-rank gloo rehearsal); default: all
set -u
TAG=${1:-r4final}; shift || true
PARTS=${*:-prof bench probes tests}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout -k 10 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
  timeout -k 10 300 python bench.py --workload sample > $O/bench_sample.json 2> $O/bench_sample.err; echo "sample rc=$?"
  timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
fi
if has prof; then bash tools/r3_profiles.sh $TAG; fi
if has probes; then
  timeout -k 10 300 python tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1; echo "pipe stamps rc=$?"
  timeout -k 10 300 python tools/rev_stamps.py 1024 96 > $O/rev_stamps.txt 2>&1; echo "rev stamps rc=$?"
  for m in p2_face p2_speech; do for r in 1 0; do
    LFI_ENC_R64=$r timeout -k 10 120 python tools/enc_probe.py --mod $m > $O/enc_probe_${m}_r64_$r.txt 2>&1
  done; done
  for gm in 8 14; do
    LFI_PGEMM_GM=$gm timeout -k 10 200 python bench.py --quick > $O/bench_gm$gm.json 2> $O/bench_gm$gm.err
  done
fi
if has tests; then
  timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  LFI_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 gloo rc=$?"
  export LFI_PARITY_REPORT=$O/parity.txt
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 $O/pytest_gpu.log
fi
