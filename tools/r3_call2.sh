#!/bin/bash
# round 3, GPU call 2: sampling precision probe, 2-product backward timing, loss-curve A/B, deep workload, new tests
set -u
O=gpurun_out/c2; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 300 python tools/sample_precision_probe.py > $O/sample_precision.md 2> $O/sample_precision.err; echo "probe rc=$?"
timeout -k 10 200 python tools/loss_curve_ab.py > $O/loss_curve_ab.md 2> $O/loss_curve_ab.err; echo "ab rc=$?"
LFI_PASS_SKIP="dpre=1,cond_wgrad=1,cond_dgrad=1,flow_pgrads=1,enc_dwih=1,enc_dwhh=1" timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_bwd2.json 2> $O/bench_bwd2.err; echo "bwd2 rc=$?"
timeout -k 10 200 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_x3.json 2> $O/bench_x3.err; echo "x3 rc=$?"
timeout -k 10 400 python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err; echo "deep rc=$?"
timeout -k 10 400 python bench.py > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
timeout -k 10 400 python -m pytest tests/test_gpu_headline_parity.py tests/test_data_module.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
