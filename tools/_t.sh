set -u
O=gpurun_out/c16; mkdir -p $O
export LFI_PARITY_REPORT=$O/parity.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
