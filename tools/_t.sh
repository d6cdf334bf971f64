set -u
O=gpurun_out/c20; mkdir -p $O
NOBASE="--cpu-baseline-seconds 0 --torch-gpu-baseline-seconds 0 --strong-anchor-batch 0 --graph-steps 0 --three-products-steps 0"
for i in 1 2 3; do
for v in 1 0; do
LFI_BENCH_ALL_TAGS=$v timeout -k 10 300 python bench.py $NOBASE > $O/b$v.json 2> $O/b$v.err; echo "alltags=$v rc=$? $(python3 -c "import json;j=json.loads([l for l in open('$O/b$v.json') if l.startswith('{')][-1]);print(j['ms_per_step'], j['roofline']['ms_per_launch'], len(j['kernel_timing']))")"
done; done
