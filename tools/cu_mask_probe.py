#!/usr/bin/env python3
"""Does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine a kernel to its CUs on this card, and which bits are which?
Times one window-encoder forward (1024 sequences x 80 frames: 896 workgroups of one CU each) on streams with 256, 192, 128, 64 CUs
enabled, as the low bits of the mask or as every other / every fourth bit. Time ~ 1 / CUs says the mask holds."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lets_face_it_amd import _lib  # noqa: E402
from lets_face_it_amd._lib import EncDesc, check  # noqa: E402


def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % rc)
    return st


def main():
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    hip = C.CDLL("libamdhip64.so")
    L = _lib.lib()
    hist, hid, B, T, start = 24, 256, 1024, 80, 24
    N = T - start
    F = N * B
    g = torch.Generator(device="cpu").manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
    xp, whh, b_ih, b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
    cond = torch.zeros(F, 896, device=dev)
    hseq = torch.zeros(hist * F * hid, device=dev)
    d = EncDesc(B, T, N, start, hist, hid, 896, 256, 1, 0, 0, 1, 0)
    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(d))), 1), device=dev)
    torch.cuda.synchronize()

    def run(stream_ptr, reps=5):
        ext = torch.cuda.ExternalStream(stream_ptr, device=dev)
        with torch.cuda.stream(ext):
            for _ in range(2):
                check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), None,
                                               cond.data_ptr(), None, hseq.data_ptr(), work.data_ptr(), stream_ptr), "fwd")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(reps):
                check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), None,
                                               cond.data_ptr(), None, hseq.data_ptr(), work.data_ptr(), stream_ptr), "fwd")
            e1.record(ext)
        ext.synchronize()
        return e0.elapsed_time(e1) / reps

    full = (1 << 256) - 1
    cases = [("all 256 bits", full), ("low 192 bits", (1 << 192) - 1), ("low 128 bits", (1 << 128) - 1), ("low 64 bits", (1 << 64) - 1),
             ("every other bit (128)", sum(1 << i for i in range(0, 256, 2))), ("every fourth bit (64)", sum(1 << i for i in range(0, 256, 4))),
             ("bits 0-15 of each 32 (128)", sum(1 << i for i in range(256) if i % 32 < 16))]
    print("window-encoder forward, 896 workgroups; ms per launch on a CU-masked stream")
    for name, bits in cases:
        st = masked_stream(hip, bits)
        print("  %-28s %.3f" % (name, run(st.value)))
        hip.hipStreamDestroy(st)


if __name__ == "__main__":
    main()
