#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one .hip file (works without a GPU):
   python tools/kernel_resources.py lets_face_it_amd/csrc/lfi_encoder.hip [name filter] [-- extra hipcc flags]
Compiles with -Rpass-analysis=kernel-resource-usage and prints one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include")]
    if os.path.basename(src) == "lfi_encoder.hip":
        flags.append("-fno-slp-vectorize")     # as the Makefile builds it
    cmd = ["hipcc"] + flags + extra + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = None, []
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)
        if flt and flt not in name:
            continue
        print("%-64s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s" % (
            name[:64], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))


if __name__ == "__main__":
    main()
