#!/usr/bin/env python3
"""VERDICT r4 next #4, feasibility probe: does MFMA-bound work run BESIDE the persistent flow walk on the same chip?
   python tools/coresidency_probe.py          (GPU box; headline shape: batch 256, T 80, K 16)
Times, with HIP events, (a) the forward walk alone (lfi_flow_seq_fwd on one stream), (b) the two conditioning GEMMs alone
(cond_transform forward + gic: gemm_planes16_kernel x 2), (c) both at once on two streams (the walk reads a private copy of gic:
the pair is independent, only the clock is shared). If the walk left room on its CUs, (c) would approach max(a, b); if the two
time-slice, (c) = a + b. Static side of the same question: tools/kernel_resources.py (VGPRs per wave of both kernels)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from argparse import Namespace
    from lets_face_it_amd.engine import _stream
    from lets_face_it_amd._lib import check
    from lets_face_it_amd.glow.models import SeqGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    hp = load_hparams_file(os.path.join(os.path.dirname(__file__), "..", "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    dev = torch.device("cuda:0")
    m = SeqGlow(Namespace(**hp)).to(dev)
    m.glow.set_actnorm_init(True)
    m.train()
    g = torch.Generator().manual_seed(0)
    batch = {k: torch.randn(256, 80, d, generator=g).to(dev) for k, d in
             (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m(batch)[1].sum().backward()
    torch.cuda.synchronize()
    eng = m.engine if hasattr(m, "engine") else m._engine
    ctx = eng._last
    s, L = eng.spec, eng.L
    F = ctx.F
    gic_w = ctx.gic.clone()
    x = batch["p1_face"]
    z = torch.empty(ctx.N, ctx.B, s.C, device=dev)
    nll = torch.empty(ctx.N, ctx.B, device=dev)
    p = eng._flow_params()
    reps = 10

    def walk():
        check(L.lfi_flow_seq_fwd(C.byref(ctx.dims), C.byref(p), eng.prep.data_ptr(), x.data_ptr(), ctx.T, s.start,
                                 gic_w.data_ptr(), ctx.stash.data_ptr(), z.data_ptr(), nll.data_ptr(), _stream()), "walk")

    def gemms():
        eng._project(ctx.cond, F, ctx.chain, True)

    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fa, fb):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s0.wait_event(e0); s1.wait_event(e0)
        for _ in range(reps):
            if fa:
                with torch.cuda.stream(s0):
                    fa()
            if fb:
                with torch.cuda.stream(s1):
                    fb()
        d0, d1 = torch.cuda.Event(), torch.cuda.Event()
        d0.record(s0); d1.record(s1)
        torch.cuda.current_stream().wait_event(d0); torch.cuda.current_stream().wait_event(d1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(2):
        timed(walk, gemms)
    a, b, c = timed(walk, None), timed(None, gemms), timed(walk, gemms)
    print("forward walk alone (flow_pipe_fwd_kernel, 256 workgroups x 512 threads):    %.3f ms" % a)
    print("cond_transform forward + gic alone (gemm_planes16_kernel x 2 + planes split): %.3f ms" % b)
    print("both at once on two streams:                                                 %.3f ms   (sum %.3f, max %.3f)" % (c, a + b, max(a, b)))
    print("pair costs %.3f ms less than the sum" % (a + b - c))


if __name__ == "__main__":
    main()
