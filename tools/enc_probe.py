#!/usr/bin/env python3
"""Times the window encoders' fused recurrence kernels alone, through the C ABI, at the headline shape (B=256, T=80, 56 timesteps):
   python tools/enc_probe.py [--mod p2_face|p2_speech] [--reps 20]
per configuration (gate stash fp32 / fp16 / none, forward and BPTT) the mean launch time over --reps launches (HIP events on the
launch stream). With an ingredient-removal library (LFI_LIB_PATH=build/var/liblfi_<variant>.so, tools/build_variant.sh) the results
are garbage and only the times mean anything."""
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
    ap.add_argument("--mod", default="p2_face")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--no-mask", action="store_true", help="no feature-dropout mask (validation / sampling)")
    ap.add_argument("--frames", type=int, default=80, help="T: frames per sequence (windows = (T - 24) * batch)")
    a = ap.parse_args()
    hist, hid = {"p2_face": (24, 256), "p2_speech": (16, 256), "p1_speech": (2, 128)}[a.mod]
    B, T, start = a.batch, a.frames, 24
    N = T - start
    F = N * B
    dev = torch.device("cuda:0")
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
    xp, whh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2
    b_ih, b_hh = rnd(3 * hid), rnd(3 * hid)
    mask = ((torch.rand(F, hist, generator=g) < 0.5).float() * 2).to(dev)
    ldc = 896
    cond = torch.zeros(F, ldc, device=dev)
    dcond = rnd(F, ldc)
    gates = torch.zeros(hist * F * 4 * hid, device=dev)
    hseq = torch.zeros(hist * F * hid, device=dev)
    dgi = torch.zeros(hist * F * hid, device=dev)
    dgh = torch.zeros(hist * F * 3 * hid, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def desc(two, s16):
        return EncDesc(B, T, N, start, hist, hid, ldc, 256, 1, 0, 0, two, s16)

    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(desc(0, 0)))), 1), device=dev)
    prow = int(L.lfi_encode_windows_bias_rows(C.byref(desc(0, 0))))
    part = torch.zeros(max(prow * 4 * hid, 1), device=dev)

    def fwd(s16, stash=True):
        d = desc(1, s16)
        check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), None if a.no_mask else mask.data_ptr(),
                                       cond.data_ptr(), gates.data_ptr() if stash else None, hseq.data_ptr(), work.data_ptr(), st), "fwd")

    def bwd(two, s16):
        d = desc(two, s16)
        check(L.lfi_encode_windows_bwd(C.byref(d), dcond.data_ptr(), ldc, whh.data_ptr(), gates.data_ptr(), hseq.data_ptr(),
                                       dgi.data_ptr(), dgh.data_ptr(), part.data_ptr(), work.data_ptr(), st), "bwd")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    print("library: %s" % _lib.LIB_PATH)
    print("%s: hist %d, hid %d, %d windows; ms per launch (incl. the ~6 us weight-fragment kernel)" % (a.mod, hist, hid, F))
    ok16 = bool(L.lfi_encode_windows_stash_f16_ok(C.byref(desc(1, 1))))
    print("  fwd, no stash           %.4f" % timed(lambda: fwd(0, stash=False)))
    print("  fwd, fp32 gate stash    %.4f" % timed(lambda: fwd(0)))
    if ok16:
        print("  fwd, fp16 gate stash    %.4f" % timed(lambda: fwd(1)))
    if a.fwd_only:
        return
    fwd(0)
    print("  bwd 3 products, fp32 stash / fp32 grads   %.4f" % timed(lambda: bwd(0, 0)))
    print("  bwd 2 products, fp32 stash / bf16 grads   %.4f" % timed(lambda: bwd(1, 0)))
    if ok16:
        fwd(1)
        print("  bwd 2 products, fp16 stash / bf16 grads   %.4f" % timed(lambda: bwd(1, 1)))


if __name__ == "__main__":
    main()
