"""200-step loss-curve A/B of round 4's kernels against round 3's arithmetic: the v_mfma_f32_16x16x32_bf16 planes GEMMs, the
window encoders' forward recurrence on that shape and the direct plane epilogues sum the same products in another order than the
32 x 32 x 16 kernels they replace (LFI_PGEMM_16=0 LFI_PGEMM_16T=0 LFI_PGEMM_DIRECT=0 LFI_ENC_M16=0 bring those back bit for bit).

Both arms start from the same weights, see the same 8 synthetic batches of 256 x 80 frames in the same order, draw the same
dropout masks and take the same negative-example branches; default arithmetic otherwise (three products forward, two backward).
A third run - the new kernels, ANOTHER dropout seed - gives the scale of ordinary run-to-run variation.

    python tools/loss_curve_kernels_ab.py [--steps 200] > profiles/round4_loss_curve_kernels_ab.md        (GPU box, ~20 s)
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

OLD = {"LFI_PGEMM_16": "0", "LFI_PGEMM_16T": "0", "LFI_PGEMM_DIRECT": "0", "LFI_ENC_M16": "0"}


def run(hp, batches, steps, env, seed):
    from argparse import Namespace
    import copy
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    for k in OLD:
        os.environ.pop(k, None)
    os.environ.update(env)        # (the library reads these switches at every call)
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    m = LetsFaceItGlow(Namespace(**copy.deepcopy(hp)))
    dev = batches[0]["p1_face"].device
    m.to(dev).train()
    torch.manual_seed(seed)
    random.seed(99)
    eng = m.seq_glow._ensure_engine(dev)
    losses = []
    for i in range(steps):
        losses.append(m.fused_training_step(batches[i % len(batches)], float(hp["lr"])))
    out = torch.stack([l.reshape(()) for l in losses]).double().cpu()
    for k in OLD:
        os.environ.pop(k, None)
    return out, eng.params.detach().double().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    from bench import synthetic_batch
    from lets_face_it_amd.glow.utils import load_hparams_file
    hp = load_hparams_file(os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    hp["batch_size"] = 256
    hp["engine_precision"] = "bf16x3"
    dev = torch.device("cuda:0")
    batches = [synthetic_batch(256, 80, 50, 27, 500 + i, dev) for i in range(8)]
    a, pa = run(hp, batches, args.steps, {}, 1234)
    b, pb = run(hp, batches, args.steps, OLD, 1234)
    c, pc = run(hp, batches, args.steps, {}, 4321)
    a2, _ = run(hp, batches, args.steps, {}, 1234)
    print("# Loss curve, round 4's kernels (16 x 16 x 32 MFMA, k-tiles in pairs, direct plane epilogues) vs round 3's arithmetic\n")
    print("final_model.yaml at BASELINE's synthetic dims, batch 256, T = 80, lr %g, Adam + clip 20, %d steps over 8 fixed batches, "
          "default arithmetic (three products forward, two backward); per-step loss = the step's own training loss. `control` = the "
          "new kernels with another dropout seed. The new-kernel arm repeated: bit-identical = %s.\n"
          % (float(hp["lr"]), args.steps, bool(torch.equal(a, a2))))
    print("| steps | loss, round-4 kernels | loss, round-3 arithmetic | |A/B difference| | |control difference| (other dropout seed) |")
    print("|---|---|---|---|---|")
    for lo in range(0, args.steps, max(1, args.steps // 10)):
        hi = min(args.steps, lo + max(1, args.steps // 10))
        sl = slice(lo, hi)
        print("| %d-%d | %.4f | %.4f | mean %.2e, max %.2e | mean %.2e, max %.2e |"
              % (lo, hi - 1, float(a[sl].mean()), float(b[sl].mean()), float((a[sl] - b[sl]).abs().mean()),
                 float((a[sl] - b[sl]).abs().max()), float((a[sl] - c[sl]).abs().mean()), float((a[sl] - c[sl]).abs().max())))
    dn = float((pa - pb).norm() / pa.norm())
    dc = float((pa - pc).norm() / pa.norm())
    print("\nParameters after %d steps: relative L2 distance between the arms %.3e; to the other-seed control %.3e."
          % (args.steps, dn, dc))
    print("\nVerdict: the A/B difference is %s the run-to-run variation of the control."
          % ("within" if float((a - b).abs().mean()) <= float((a - c).abs().mean()) else "ABOVE"))


if __name__ == "__main__":
    main()
