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
    ap.add_argument("--precision", type=int, default=1, help="0 exact f32 MFMA, 1 bf16x3")
    args = ap.parse_args()
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    dev = torch.device("cuda:0")
    eng = GlowEngine(ModelSpec(Namespace(**Fixture("tiny").hp)), dev)
    eng.precision = args.precision
    shapes = [("cond_fwd", 14336, 8192, 890, 1, 1), ("cond_wgrad", 8192, 890, 14336, 0, 0),
              ("cond_dgrad", 14336, 640, 8192, 1, 0), ("enc_step", 14336, 768, 256, 1, 1),
              ("enc_bwd", 14336, 256, 768, 1, 0)]
    for sh in args.shape:
        M, N, K, a, b = [int(v) for v in sh.split(",")]
        shapes.append(("user", M, N, K, a, b))
    g = torch.Generator().manual_seed(0)
    r16 = lambda v: (v + 15) // 16 * 16  # noqa: E731
    for name, M, N, K, akc, bkc in shapes:
        lda, ldb = (r16(K) if akc else r16(M)), (r16(K) if bkc else r16(N))  # padded rows, as the engine lays them out
        A = torch.randn((M, lda) if akc else (K, lda), generator=g).to(dev)
        Bm = torch.randn((N, ldb) if bkc else (K, ldb), generator=g).to(dev)
        Cm = torch.empty(M, N, device=dev)
        run = lambda: eng.gemm(M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, N)  # noqa: E731
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
        if os.environ.get("LFI_GEMM_STAMPS"):   # -DY2_STAMPS build: per-phase s_memtime sums of waves 0 and 4 of workgroup 8
            c = Cm.view(-1)[:16].cpu().tolist()
            for w, v in ((0, c[:8]), (4, c[8:16])):
                n = max(v[6], 1.0)
                print("    wave %d, cycles per k-tile: top->stage(late) %.0f | stage(late) %.0f | mfma+frag reads %.0f | stage(early) %.0f | "
                      "barrier %.0f | (epilogue total %.0f), k-tiles %d" % (w, v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5], n))

    # the same k-contiguous products on pre-split planes (lfi_planes_from_f32 + lfi_gemm_planes), split time reported apart
    def timeit(run):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            run()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / args.reps

    for name, M, N, K in (("cond_fwd", 14336, 8192, 890), ("enc_step", 14336, 768, 256)):
        ld = r16(K)
        A = torch.randn(M, ld, generator=g).to(dev)
        Bm = torch.randn(N, ld, generator=g).to(dev)
        if os.environ.get("LFI_PROBE_ZEROS"):   # power check: all-zero operands toggle nothing (MI355X_MICROARCH.md, DVFS give-back)
            A.zero_()
            Bm.zero_()
        Cm = torch.empty(M, N, device=dev)
        bias = torch.randn(N, generator=g).to(dev)
        t_sa = timeit(lambda: eng.planes("probe.pa", A, ld, M, K))
        t_sb = timeit(lambda: eng.planes("probe.pb", Bm, ld, N, K))
        Ap, nka = eng.planes("probe.pa", A, ld, M, K)
        Bp, nkb = eng.planes("probe.pb", Bm, ld, N, K)
        stamps = torch.zeros(64, device=dev)
        if os.environ.get("LFI_PG_STAMPS"):   # -DPG_STAMPS build: per-phase s_memtime sums of waves 0 and 13 of workgroup 300
            eng.L.lfi_debug_set_stamps(stamps.data_ptr())
        ms = timeit(lambda: eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, Cm, N, bias=bias, act=1))
        if os.environ.get("LFI_PG_STAMPS"):
            eng.L.lfi_debug_set_stamps(None)
            v = stamps.cpu().tolist()
            for w, o in ((0, v[:8]), (13, v[8:16])):
                n = max(o[4], 1.0)
                print("    wave %2d: cycles per k-tile: issue+mfma %.0f | vmcnt wait %.0f | lgkm wait %.0f | barrier %.0f | total %.0f; "
                      "clock %.2f GHz" % (w, o[0] / n, o[1] / n, o[2] / n, o[3] / n, o[5] / n, o[5] / max(o[6], 1.0) * 0.1))
        print("%-11s planes M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s   (split A %.3f ms, split B %.3f ms)"
              % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, t_sa, t_sb))

    # the two batched (one product per flow step) short-K shapes of the training step, with each tile shape pinned
    Ks, F, D, G3 = 16, 14336, 512, 384
    c = torch.randn(F, Ks * D, generator=g).to(dev)
    dgi = torch.randn(Ks, F, G3, generator=g).to(dev)
    wc = torch.randn(Ks, G3, D, generator=g).to(dev)
    gic = torch.empty(Ks, F, G3, device=dev)
    bias = torch.randn(Ks, G3, generator=g).to(dev)
    for pin, label in ((0, "plan"), (0x10, "256 x 256"), (0x20, "128 x 128")):
        eng.precision = args.precision | pin
        runs = (
            ("flow_gic  (c W_c^T + b)", 2.0 * F * G3 * D * Ks,
             lambda: eng.gemm(F, G3, D, c, Ks * D, 1, wc, D, 1, gic, G3, bias=bias, batch=Ks, sA=D, sB=G3 * D, sC=F * G3, sBias=G3)),
            ("flow_dpre (dgi W_c * leaky'(c), in place)", 2.0 * F * G3 * D * Ks,
             lambda: eng.gemm(F, D, G3, dgi, G3, 1, wc, D, 0, c, Ks * D, act=2, slope=0.01, G=c, ldg=Ks * D, batch=Ks,
                              sA=F * G3, sB=G3 * D, sC=D, sG=D)),
        )
        for name, flops, run in runs:
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
            print("%-45s tiles %-9s: %.3f ms  %.1f TFLOP/s" % (name, label, ms, flops / ms / 1e9))
    eng.precision = args.precision
    cp, nkc = eng.planes("probe.pc", c, Ks * D, F, Ks * D)
    wp, nkw = eng.planes("probe.pw", wc.view(Ks * G3, D), D, Ks * G3, D)
    t_split = timeit(lambda: eng.planes("probe.pc", c, Ks * D, F, Ks * D))
    ms = timeit(lambda: eng.gemm_planes(F, G3, D, cp, nkc, wp, nkw, gic, G3, bias=bias, batch=Ks, a_stride=(D // 16) * 1024,
                                        b_stride=(G3 // 32) * nkw * 1024, sC=F * G3, sBias=G3))
    print("flow_gic on planes: %.3f ms  %.1f TFLOP/s   (split of c, F x Ks D: %.3f ms)" % (ms, 2.0 * F * G3 * D * Ks / ms / 1e9, t_split))


if __name__ == "__main__":
    main()
