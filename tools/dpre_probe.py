"""Variants of the batched dpre product (F x D x G per flow step, in place over c) to locate its cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402


def main():
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    dev = torch.device("cuda:0")
    eng = GlowEngine(ModelSpec(Namespace(**Fixture("tiny").hp)), dev)
    eng.precision = 1
    F, D, G, Ks = 14336, 512, 384, 16
    KD = Ks * D
    g = torch.Generator().manual_seed(0)
    dgi = torch.randn(Ks, F, G, generator=g).to(dev)
    wc = torch.randn(Ks, G, D, generator=g).to(dev)
    c = torch.randn(F, KD, generator=g).to(dev)
    c2 = torch.randn(Ks, F, D, generator=g).to(dev)
    gic = torch.empty(Ks, F, G, device=dev)

    def timeit(name, fn, flops):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("%-40s %.3f ms  %.1f TFLOP/s" % (name, ms, flops / ms / 1e9))

    fl = 2.0 * F * D * G * Ks
    timeit("dpre as in the engine (in place, act 2)", lambda: eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, c, KD, act=2, G=c, ldg=KD, batch=Ks, sA=F * G, sB=G * D, sC=D, sG=D), fl)
    timeit("dpre, no activation", lambda: eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, c, KD, batch=Ks, sA=F * G, sB=G * D, sC=D), fl)
    timeit("dpre, compact output [Ks][F][D]", lambda: eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, c2, D, batch=Ks, sA=F * G, sB=G * D, sC=F * D), fl)
    timeit("dpre, compact, act 2 from compact G", lambda: eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, c2, D, act=2, G=c2, ldg=D, batch=Ks, sA=F * G, sB=G * D, sC=F * D, sG=F * D), fl)
    timeit("gic as in the engine", lambda: eng.gemm(F, G, D, c, KD, 1, wc, D, 1, gic, G, batch=Ks, sA=D, sB=G * D, sC=F * G), fl)
    wct = wc.transpose(1, 2).contiguous()  # [Ks][D][G]: k-contiguous B for dpre
    timeit("dpre, B k-contiguous (wc^T), in place act 2", lambda: eng.gemm(F, D, G, dgi, G, 1, wct, G, 1, c, KD, act=2, G=c, ldg=KD, batch=Ks, sA=F * G, sB=G * D, sC=D, sG=D), fl)
    eng.precision = 0
    timeit("dpre f32 exact (in place, act 2)", lambda: eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, c, KD, act=2, G=c, ldg=KD, batch=Ks, sA=F * G, sB=G * D, sC=D, sG=D), fl)


if __name__ == "__main__":
    main()
