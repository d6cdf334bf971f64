#!/usr/bin/env python3
"""Do the two big window encoders (p2_face: hist 24, p2_speech: hist 16; hid 256) finish sooner when their recurrences run AT THE SAME
TIME as two-workgroups-per-CU kernels (the 32-window kernels, LFI_ENC_R64=0: a p2_face and a p2_speech workgroup share a CU, one's
memory stalls under the other's MFMAs, at twice the weight stream per window) than one after the other as one-workgroup-per-CU kernels
(the 64-window ones, the default)?   python tools/enc_pair_probe.py        (GPU box; headline shape, through the C ABI)
Prints ms for forward (fp16 gate stash) and BPTT (two products, fp16 stash): sequential on one stream / concurrent on two streams, for
both kernel families."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lets_face_it_amd import _lib  # noqa: E402
from lets_face_it_amd._lib import EncDesc, check  # noqa: E402


class Enc:
    def __init__(self, L, dev, hist, hid, seed, B=256, T=80, start=24):
        self.L, self.hist, self.hid = L, hist, hid
        N = T - start
        F = N * B
        g = torch.Generator(device="cpu").manual_seed(seed)
        rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
        self.xp, self.whh, self.b_ih, self.b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
        self.mask = ((torch.rand(F, hist, generator=g) < 0.5).float() * 2).to(dev)
        self.ldc = 896
        self.cond = torch.zeros(F, self.ldc, device=dev)
        self.dcond = rnd(F, self.ldc)
        self.gates = torch.zeros(hist * F * 4 * hid, device=dev)
        self.hseq = torch.zeros(hist * F * hid, device=dev)
        self.dgi = torch.zeros(hist * F * hid, device=dev)
        self.dgh = torch.zeros(hist * F * 3 * hid, device=dev)
        self.d = EncDesc(B, T, N, start, hist, hid, self.ldc, 256, 1, 0, 0, 1, 1)
        self.work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(self.d))), 1), device=dev)
        self.part = torch.zeros(4 * F // 32 * 4 * hid, device=dev)

    def fwd(self):
        st = torch.cuda.current_stream().cuda_stream
        check(self.L.lfi_encode_windows_fwd(C.byref(self.d), self.xp.data_ptr(), self.whh.data_ptr(), self.b_ih.data_ptr(), self.b_hh.data_ptr(),
                                            self.mask.data_ptr(), self.cond.data_ptr(), self.gates.data_ptr(), self.hseq.data_ptr(),
                                            self.work.data_ptr(), st), "fwd")

    def bwd(self):
        st = torch.cuda.current_stream().cuda_stream
        check(self.L.lfi_encode_windows_bwd(C.byref(self.d), self.dcond.data_ptr(), self.ldc, self.whh.data_ptr(), self.gates.data_ptr(),
                                            self.hseq.data_ptr(), self.dgi.data_ptr(), self.dgh.data_ptr(), self.part.data_ptr(),
                                            self.work.data_ptr(), st), "bwd")


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    face, speech = Enc(L, dev, 24, 256, 1), Enc(L, dev, 16, 256, 2)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 10

    def timed(what, concurrent):
        def once():
            if concurrent:
                with torch.cuda.stream(s0):
                    getattr(face, what)()
                with torch.cuda.stream(s1):
                    getattr(speech, what)()
            else:
                with torch.cuda.stream(s0):
                    getattr(face, what)()
                    getattr(speech, what)()
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s0.wait_event(e0); s1.wait_event(e0)
        for _ in range(reps):
            once()
            if concurrent:          # the pair forms one unit of work: the next pair starts when both are done (as in the step)
                d1 = torch.cuda.Event(); d1.record(s1); s0.wait_event(d1)
                d0 = torch.cuda.Event(); d0.record(s0); s1.wait_event(d0)
        d0, d1 = torch.cuda.Event(), torch.cuda.Event()
        d0.record(s0); d1.record(s1)
        torch.cuda.current_stream().wait_event(d0); torch.cuda.current_stream().wait_event(d1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for fam, env in (("64-window kernels, one workgroup per CU (default)", {"LFI_ENC_R64": "1"}),
                     ("32-window kernels, two workgroups per CU (LFI_ENC_R64=0)", {"LFI_ENC_R64": "0"})):
        os.environ.update(env)
        face.fwd(); speech.fwd()
        torch.cuda.synchronize()
        print(fam)
        for what in ("fwd", "bwd"):
            a, b = timed(what, False), timed(what, True)
            print("   %s p2_face + p2_speech: one after the other %.3f ms, at the same time on two streams %.3f ms" % (what, a, b), flush=True)


if __name__ == "__main__":
    main()
