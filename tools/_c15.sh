set -u
O=gpurun_out/c15; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or walk or abort" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
NOBASE="--cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 bench.py $NOBASE --steps 12 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 tools/rocpd_stats.py $O/prof/run_results.db 14 > $O/kernel_stats.md 2>&1
rm -rf $O/prof
grep '^{' $O/prof.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
grep flow_pipe $O/kernel_stats.md
timeout -k 10 300 python tools/pipe_stamps.py > $O/stamps.txt 2>&1; echo "stamps rc=$?"
