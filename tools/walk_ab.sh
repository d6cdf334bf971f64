# A/B of kernel variants in ONE gpurun call: build each variant of a .hip file to build/var/liblfi_<variant>.so (it travels to the box; the
# library is picked by LFI_LIB_PATH), then   tools/walk_ab.sh <outtag> <variant> [<variant> ...]   prints, per variant, the walk tests'
# outcome (SKIPT=1 skips them: ingredient-removal builds compute garbage), the bench step time and the flow walks' rocprofv3 kernel times
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
NOBASE="--cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
for v in "$@"; do
  export LFI_LIB_PATH=$PWD/build/var/liblfi_$v.so
  [ -n "${SKIPT:-}" ] || timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or walk or abort" > $O/t_$v.log 2>&1; echo "$v tests rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o run -- python3 bench.py --no-gpu-state $NOBASE --steps 12 > $O/prof_$v.log 2>&1 || { echo "$v prof failed"; exit 1; }
  python3 tools/rocpd_stats.py $O/prof_$v/run_results.db 14 > $O/stats_$v.md 2>&1
  rm -rf $O/prof_$v
  echo "$v step $(grep '^{' $O/prof_$v.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
  grep flow_pipe $O/stats_$v.md
done
