"""Phase timing of the persistent flow walks (s_memtime stamps of workgroup (Ks/2, tile 0)): python tools/pipe_stamps.py"""
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
    m.eval()
    g = torch.Generator().manual_seed(0)
    batch = {k: torch.randn(256, 80, d, generator=g).to(dev) for k, d in
             (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m(batch)[1].sum().backward()
    st = torch.zeros(8192 + 64, dtype=torch.int64, device=dev)
    _lib.lib().lfi_debug_set_stamps(st.data_ptr())
    m.zero_grad(set_to_none=True)
    m(batch)[1].sum().backward()
    torch.cuda.synchronize()
    _lib.lib().lfi_debug_set_stamps(None)
    s = st.cpu()
    N = 56
    # s_memtime ticks: calibrated against the walk's wall time below (the counter does not run at 100 MHz on this part)
    for d, names in ((0, ["gic issue + wait for step k-1", "P0 x load + actnorm", "P1 invconv", "P2 recurrent cell", "P3 LinearZeros",
                          "P4 coupling + tile stores", "drain + publish"]),
                     (1, ["stash issue + wait for step k+1 + dx loads", "Q0 coupling bwd", "Q1 cell bwd", "Q2 dz1 / dh_prev", "Q3 invconv/actnorm bwd",
                          "drain + publish", "z-tile waves' dh_prev"])):
        t = s[4096 + 2048 * d: 4096 + 2048 * d + 16 * N].view(N, 16)[:, :8].double()
        order = range(N) if d == 0 else range(N - 1, -1, -1)
        rows = [t[n] for n in order]
        ph = torch.stack([torch.cat([r[1:] - r[:-1], (rows[i + 1][0] - r[7]).reshape(1) if i + 1 < N else torch.zeros(1, dtype=torch.float64)])
                          for i, r in enumerate(rows)])
        mid = ph[8:48]
        tot = float(mid.sum(1).mean())
        print("%s walk, workgroup (k = Ks/2, tile 0), mean share of a steady-state timestep per phase (s_memtime ticks):"
              % ("backward" if d else "forward"))
        for nm, v in zip(names, mid.mean(0)[:7].tolist()):
            print("   %-44s %8.0f ticks  %5.1f %%" % (nm, v, 100.0 * v / tot))
        print("   %-44s %8.0f ticks  %5.1f %%" % ("loop back edge", float(mid.mean(0)[7]), 100.0 * float(mid.mean(0)[7]) / tot))
        print("   per timestep %.0f ticks; this workgroup's whole walk %.0f ticks = %d timesteps x %.0f"
              % (tot, float(rows[-1][7] - rows[0][0]), N, float(rows[-1][7] - rows[0][0]) / N))


if __name__ == "__main__":
    main()
