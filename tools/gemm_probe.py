"""Isolated timing of the three cond_transform GEMM shapes (and any --shape M,N,K,akc,bkc) through the C ABI.

    python tools/gemm_probe.py [--reps 20]
Used under rocprofv3 (--kernel-trace --stats, or --pmc ...) to study the GEMM kernel alone.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shape", action="append", default=[])
    args = ap.parse_args()
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    dev = torch.device("cuda:0")
    eng = GlowEngine(ModelSpec(Namespace(**Fixture("tiny").hp)), dev)
    shapes = [("cond_fwd", 14336, 8192, 1530, 1, 1), ("cond_wgrad", 8192, 1530, 14336, 0, 0),
              ("cond_dgrad", 14336, 1280, 8192, 1, 0), ("enc_step", 14336, 768, 256, 1, 1)]
    for sh in args.shape:
        M, N, K, a, b = [int(v) for v in sh.split(",")]
        shapes.append(("user", M, N, K, a, b))
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, akc, bkc in shapes:
        A = torch.randn((M, K) if akc else (K, M), generator=g).to(dev)
        Bm = torch.randn((N, K) if bkc else (K, N), generator=g).to(dev)
        Cm = torch.empty(M, N, device=dev)
        run = lambda: eng.gemm(M, N, K, A, K if akc else M, akc, Bm, K if bkc else N, bkc, Cm, N)  # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / args.reps
        print("%-11s M=%d N=%d K=%d akc=%d bkc=%d: %.3f ms  %.1f TFLOP/s" % (name, M, N, K, akc, bkc, ms, 2.0 * M * N * K / ms / 1e9))


if __name__ == "__main__":
    main()
