#!/usr/bin/env python3
"""Repeat-launch determinism of the window encoders' fallback BPTT kernel, through the C ABI: what LFI_ENC_WIDE_BWD=0 selects, launched N
times on the same inputs, outputs compared bit for bit.
   LFI_ENC_WIDE=0 LFI_ENC_WIDE_BWD=0 [LFI_LIB_PATH=build/var/liblfi_slp.so] python tools/enc_bwd_repro.py [--mod p1_speech] [--reps 20]
Until round 5 that was enc_gru_bwd_fused_kernel<true> (bf16x3), which differed launch to launch in an SLP-vectorised build
(profiles/round5_enc_bwd_repro.txt: 19 of 19). Round 6 deleted it: the switch now selects the exact-f32 accumulator-layout kernel, and
this tool is the check that the fallback is bit-identical in the tree's build and in the SLP build (LFI_SLP=1 tools/build_variant.sh
slp lfi_encoder.hip) alike - profiles/round6_enc_bwd_repro.txt."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lets_face_it_amd import _lib  # noqa: E402
from lets_face_it_amd._lib import EncDesc, check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mod", default="p1_speech")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    hist, hid = {"p2_face": (24, 256), "p2_speech": (16, 256), "p1_speech": (2, 128)}[a.mod]
    B, T, start = a.batch, 80, 24
    N = T - start
    F = N * B
    dev = torch.device("cuda:0")
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
    xp, whh, b_ih, b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
    ldc = 896
    cond, dcond = torch.zeros(F, ldc, device=dev), rnd(F, ldc)
    gates, hseq = torch.zeros(hist * F * 4 * hid, device=dev), torch.zeros(hist * F * hid, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    d = EncDesc(B, T, N, start, hist, hid, ldc, 256, 1, 0, 0, 0, 0)
    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(d))), 1), device=dev)
    prow = int(L.lfi_encode_windows_bias_rows(C.byref(d)))
    part = torch.zeros(max(prow * 4 * hid, 1), device=dev)
    check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), None, cond.data_ptr(),
                                   gates.data_ptr(), hseq.data_ptr(), work.data_ptr(), st), "fwd")
    outs = []
    for _ in range(a.reps):
        dgi, dgh = torch.zeros(hist * F * hid, device=dev), torch.zeros(hist * F * 3 * hid, device=dev)
        check(L.lfi_encode_windows_bwd(C.byref(d), dcond.data_ptr(), ldc, whh.data_ptr(), gates.data_ptr(), hseq.data_ptr(),
                                       dgi.data_ptr(), dgh.data_ptr(), part.data_ptr(), work.data_ptr(), st), "bwd")
        torch.cuda.synchronize()
        outs.append((dgi.view(torch.int32).clone(), dgh.view(torch.int32).clone()))
    bad = [(int((o[0] != outs[0][0]).sum()), int((o[1] != outs[0][1]).sum())) for o in outs[1:]]
    nbad = sum(1 for x in bad if x != (0, 0))
    print("library %s, %s (hist %d, hid %d, %d windows): %d of %d repeat launches differ from the first; unequal (dgi, dgh) elements per launch: %s"
          % (_lib.LIB_PATH, a.mod, hist, hid, F, nbad, a.reps - 1, bad[:8]))
    if nbad:
        i = next(k for k, x in enumerate(bad) if x != (0, 0)) + 1
        idx = (outs[i][1] != outs[0][1]).nonzero().flatten()
        if idx.numel():
            G3 = 3 * hid
            e = idx[:2048]
            srow, col = e // G3, e % G3
            print("   first differing dgh elements: (step*F + window, column) %s ..." % list(zip(srow[:6].tolist(), col[:6].tolist())))
            print("   gate blocks touched (0 = d r, 1 = d z, 2 = d n * r): %s; values first launch %s, launch %d %s" % (
                sorted(set((col // hid).tolist())), outs[0][1].view(torch.float32)[e[:4]].tolist(), i, outs[i][1].view(torch.float32)[e[:4]].tolist()))


if __name__ == "__main__":
    main()
