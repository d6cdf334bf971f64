#!/bin/bash
# round 6: where does the f32-mode sampler's 17 % over the plain-fp32 floor come from? The K = 16 sampling
# test (batch 8 x 56 frames vs the fp64 oracle, both engine modes) and the sampling bench under library variants built with
# -DLFI_EXACT_GATES (lfi_common.h): 1 = libm expf / tanhf + true division, 2 = hardware exp2 + one Newton step on the reciprocal;
# "flow" = the flow kernels only (reverse cells), "all" = window encoders and the per-frame conditioning kernel too.
# usage: tools/r6_exact_gates.sh <tag>   (variants built beforehand: tools/build_variant.sh eg1flow lfi_flow.hip -DLFI_EXACT_GATES=1 ...)
set -u
O=gpurun_out/${1:-r6eg}; mkdir -p $O
export TMPDIR=/tmp
for v in tree eg1flow eg1all eg2flow eg2all; do
  if [ "$v" = "tree" ]; then unset LFI_LIB_PATH; else export LFI_LIB_PATH=$PWD/build/var/liblfi_$v.so; fi
  LFI_PARITY_REPORT=$O/parity_$v.txt timeout -k 10 300 python -m pytest tests/test_gpu_headline_parity.py -q -m gpu -k test_k16_sampling > $O/pytest_$v.log 2>&1; echo "$v pytest rc=$?"
  for i in 1 2; do
    timeout -k 10 200 python bench.py --workload sample --quick --no-gpu-state > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
  done
done
unset LFI_LIB_PATH
python3 - $O <<'PY'
import json, sys, glob, re
O = sys.argv[1]
print("| library | f32 engine: max err, rms | bf16x3 engine: max err, rms | fp32 torch floor: max, rms | max err / floor (f32, bf16x3) | rms / floor rms (f32, bf16x3) | sampling call ms (2 runs) | chain kernel ms per launch |")
print("|---|---|---|---|---|---|---|---|")
for v in ("tree", "eg1flow", "eg1all", "eg2flow", "eg2all"):
    errs, rms, floor, frms = {}, {}, None, None
    try:
        for line in open("%s/parity_%s.txt" % (O, v)):
            m = re.search(r"\((f32|bf16x3)\): max abs err vs fp64 oracle ([0-9.e+-]+) .*?plain fp32 torch on the CPU: ([0-9.e+-]+).*?rms err ([0-9.e+-]+), plain fp32 torch rms ([0-9.e+-]+)", line)
            if m:
                errs[m.group(1)] = float(m.group(2)); floor = float(m.group(3)); rms[m.group(1)] = float(m.group(4)); frms = float(m.group(5))
    except OSError:
        pass
    ms, chain = [], []
    for f in sorted(glob.glob("%s/bench_%s_*.json" % (O, v))):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            ms.append(d["ms_per_step"]); chain.append(d["roofline"]["ms_per_launch"])
        except Exception:
            pass
    if floor:
        nan = float("nan")
        print("| %s | %.3e, %.3e | %.3e, %.3e | %.3e, %.3e | %.3f, %.3f | %.3f, %.3f | %s | %s |" % (
              v, errs.get("f32", nan), rms.get("f32", nan), errs.get("bf16x3", nan), rms.get("bf16x3", nan), floor, frms,
              errs.get("f32", nan) / floor, errs.get("bf16x3", nan) / floor, rms.get("f32", nan) / frms, rms.get("bf16x3", nan) / frms,
              " ".join("%.2f" % m for m in ms), " ".join("%.4f" % c for c in chain)))
PY
