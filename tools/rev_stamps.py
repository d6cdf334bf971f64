"""Phase timing of the sampler's per-frame reverse chain (flow_rev_chain_kernel, s_memtime stamps of workgroup (LFI_STAMP_K, tile 0)):
   python tools/rev_stamps.py [batch] [frames]
Eager launches (LFI_NO_GRAPH=1: the stamp pointer is a kernel argument, a captured graph would keep the one of its capture)."""
import os
import sys

os.environ["LFI_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

NAMES = ["weight / gic / state loads issued", "wait for step k+1's tile", "R0 stage tile + h_prev (+ W^-1 issue)", "P2 recurrent cell",
         "P3 LinearZeros", "R3 coupling inverse", "R4 W^-1 + actnorm^-1 + stores", "drain + publish"]


def main():
    from argparse import Namespace
    from lets_face_it_amd import _lib
    from lets_face_it_amd.glow.models import SeqGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    hp = load_hparams_file(os.path.join(os.path.dirname(__file__), "..", "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    dev = torch.device("cuda:0")
    m = SeqGlow(Namespace(**hp)).to(dev)
    m.glow.set_actnorm_init(True)
    m.eval()
    Ks = m.spec.Ks
    g = torch.Generator().manual_seed(0)
    T = 24 + frames
    data = {"p1_face": torch.zeros(B, T, 50, device=dev)}
    for k, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[k] = torch.randn(B, T, d, generator=g).to(dev)
    noise = torch.randn(frames, B, 50, generator=g).to(dev)
    m.inference(T, data, noise=noise)
    for kst in (Ks - 1, Ks // 2, 0):
        os.environ["LFI_STAMP_K"] = str(kst)
        st = torch.zeros(8192 + 64, dtype=torch.int64, device=dev)
        _lib.lib().lfi_debug_set_stamps(st.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.inference(T, data, noise=noise)
        e1.record()
        torch.cuda.synchronize()
        _lib.lib().lfi_debug_set_stamps(None)
        n = min(frames, 128)
        t = st.cpu()[1024:1024 + 16 * n].view(n, 16)[:, :9].double()
        ph = t[:, 1:] - t[:, :-1]                      # (frames, 8) phases of one cell
        gap = torch.cat([t[1:, 0] - t[:-1, 8], torch.zeros(1, dtype=torch.float64)])   # from a cell's end to the next frame's cell entry
        lo, hi = min(8, n - 2), max(min(8, n - 2) + 1, n - 8)
        mid, gmid = ph[lo:hi], gap[lo:hi]
        cell = float(mid.sum(1).mean())
        frame_ticks = float((t[hi, 0] - t[lo, 0]) / (hi - lo))
        print("reverse chain, batch %d, workgroup (k = %d, tile 0), mean per generated frame over frames %d..%d (s_memtime ticks); whole call "
              "%.1f ms eager" % (B, kst, lo, hi, e0.elapsed_time(e1)))
        for nm, v in zip(NAMES, mid.mean(0).tolist()):
            print("   %-44s %8.0f ticks  %5.1f %% of the cell" % (nm, v, 100.0 * v / cell))
        print("   %-44s %8.0f ticks" % ("cell total", cell))
        print("   %-44s %8.0f ticks  (frame period %.0f ticks: the per-frame GEMMs, the window gather, the launch)"
              % ("from this cell's end to its next frame's entry", float(gmid.mean()), frame_ticks))


if __name__ == "__main__":
    main()
