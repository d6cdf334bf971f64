"""Timing + bitwise check of lfi_encode_windows_scatter's two kernels (LFI_ENC_SCATTER16=0: 8-byte loads, one frame row per
workgroup; 1: 16-byte loads, several rows per workgroup) at the benchmark's three encoder shapes, alone on the card.
Usage: python tools/scatter_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lets_face_it_amd import _lib  # noqa: E402
from lets_face_it_amd._lib import EncDesc, check  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0, 1]
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B, T, start = 256, 80, 24
    N = T - start
    F = N * B
    g = torch.Generator().manual_seed(5)
    for mod, (hist, hid) in {"p2_face": (24, 256), "p2_speech": (16, 256), "p1_speech": (2, 128)}.items():
        d = EncDesc(B, T, N, start, hist, hid, 896, 256, 1, 0, 0, 1, 1)
        assert L.lfi_encode_windows_grad_stash_bf16(C.byref(d))
        dgh = (torch.randn(hist * F * 3 * hid, generator=g) * 0.1).to(torch.bfloat16).to(dev)
        dgi = (torch.randn(hist * F * hid, generator=g) * 0.1).to(torch.bfloat16).to(dev)
        mask = ((torch.rand(F, hist, generator=g) < 0.5).float() * 2).to(dev)
        ref = None
        for v in variants:
            os.environ["LFI_ENC_SCATTER16"] = str(v)
            dxp = torch.full((B * T, 3 * hid), float("nan"), device=dev)
            run = lambda: check(L.lfi_encode_windows_scatter(C.byref(d), dgi.data_ptr(), dgh.data_ptr(), mask.data_ptr(),  # noqa: E731
                                                             dxp.data_ptr(), st), "scatter")
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 20
            if ref is None:
                ref = dxp.clone()
            nbytes = (dgh.numel() * 2 // 3 + dgi.numel()) * 2 + dxp.numel() * 4
            print("%-10s v%d  %7.1f us  %5.2f TB/s  bitwise=%s" % (mod, v, us, nbytes / us / 1e6, bool(torch.equal(ref, dxp))), flush=True)


if __name__ == "__main__":
    main()
