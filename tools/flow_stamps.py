"""Phase timing of one forward flow cell (s_memtime stamps, diagnostics): python tools/flow_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from argparse import Namespace
    from lets_face_it_amd import _lib
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
    with torch.no_grad():
        m(batch)
        st = torch.zeros(16 * 16 + 64, dtype=torch.int64, device=dev)
        _lib.lib().lfi_debug_set_stamps(st.data_ptr())
        m(batch)
        torch.cuda.synchronize()
        _lib.lib().lfi_debug_set_stamps(None)
    enc = st.cpu()[128 + 0:128 + 8]  # overlaps flow rows >= 8: read the encoder stamps first (k = 0 / 7 rows below stay valid)
    print("encoder fwd (last modality run, step 5): MFMA %d, barrier %d, groups %s, barrier %d cycles" % (
        int(enc[1] - enc[0]), int(enc[2] - enc[1]), [int(enc[i + 1] - enc[i]) for i in range(2, 6)], int(enc[7] - enc[6])))
    s = st.cpu()[:256].view(16, 16)
    # the last diagonal to touch cell row k is the final launch: all rows hold the stamps of their last cell
    names = ["weights issued", "P0 start", "P0 done", "P1 done", "P2 done", "P3 done", "P4 done"]
    for k in (0, 7):
        d = [(int(s[k, i + 1]) - int(s[k, i])) * 10 for i in range(6)]  # 100 MHz ticks -> ns
        print("cell k=%d:" % k, ", ".join("%s %d ns" % (n, v) for n, v in zip(
            ["issue", "P0", "P1", "P2", "P3", "P4"], d)), " total %d ns" % ((int(s[k, 6]) - int(s[k, 0])) * 10))


if __name__ == "__main__":
    main()
