"""Phase stamps of the wide forward window-encoder kernel (library built with -DLFI_ENC_STAMPS: tools/build_variant.sh stamps
lfi_encoder.hip -DLFI_ENC_STAMPS; LFI_LIB_PATH=build/var/liblfi_stamps.so LFI_ENC_R64=0 python tools/enc_stamps.py):
s_memtime of wave 0 of workgroup 0 at the phase boundaries of steps 4..11."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lets_face_it_amd import _lib  # noqa: E402
from lets_face_it_amd._lib import EncDesc, check  # noqa: E402


def main():
    hist, hid, B, T, start = 24, 256, 256, 80, 24
    N = T - start
    F = N * B
    dev = torch.device("cuda:0")
    L = _lib.lib()
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
    xp, whh, b_ih, b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
    mask = ((torch.rand(F, hist, generator=g) < 0.5).float() * 2).to(dev)
    cond = torch.zeros(F, 896, device=dev)
    gates = torch.zeros(hist * F * 4 * hid, device=dev)
    hseq = torch.zeros(hist * F * hid, device=dev)
    d = EncDesc(B, T, N, start, hist, hid, 896, 256, 1, 0, 0, 0, 0)
    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(d))), 1), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for stash in (False, True):
        stamps = torch.zeros(8192 + 64, dtype=torch.int64, device=dev)
        for rep in range(3):
            if rep == 2:
                L.lfi_debug_set_stamps(stamps.data_ptr())
            check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), mask.data_ptr(),
                                           cond.data_ptr(), gates.data_ptr() if stash else None, hseq.data_ptr(), work.data_ptr(), st), "fwd")
        torch.cuda.synchronize()
        L.lfi_debug_set_stamps(None)
        t = stamps.cpu()[128:128 + 64].view(8, 8)[:, :6].double()
        ph = t[:, 1:] - t[:, :-1]
        back = t[1:, 0] - t[:-1, 5]
        print("wide forward kernel, p2_face shape, %s: wave 0 of workgroup 0, mean over steps 4..11 (s_memtime ticks)" % ("fp32 gate stash" if stash else "no stash"))
        for nm, v in zip(["k loop (h W_hh^T, 288 MFMAs)", "epilogue: loads, r / z transposes, gate math", "wait at the mid-step barrier",
                          "epilogue: n gate, h, stash stores, state images", "wait at the end-of-step barrier"], ph.mean(0).tolist()):
            print("   %-52s %8.0f ticks" % (nm, v))
        print("   %-52s %8.0f ticks" % ("loop back edge", float(back.mean())))
        print("   %-52s %8.0f ticks" % ("step total", float((t[1:, 0] - t[:-1, 0]).mean())))


if __name__ == "__main__":
    main()
