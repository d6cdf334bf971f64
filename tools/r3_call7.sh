#!/bin/bash
# round 3, GPU call 7: two-product BPTT recurrence: sweep at B=256 with the new class, parity, bench, loss-curve A/B
set -u
O=gpurun_out/c7; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 500 python tools/precision_sweep.py --batch 256 --seq-len 80 --bwd-only > $O/precision_sweep_b256.md 2> $O/sweep.err; echo "sweep rc=$?"
grep -E "enc_bptt|EVERY" $O/precision_sweep_b256.md
timeout -k 10 300 python tools/loss_curve_ab.py > $O/loss_curve_ab.md 2> $O/ab.err; echo "ab rc=$?"
tail -4 $O/loss_curve_ab.md
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "headline_config or planes_chain or full_model_k16 or graphed or strong_scaling" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
tail -4 $O/pytest_parity.log
timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --steps 40 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
