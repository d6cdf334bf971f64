"""200-step loss-curve A/B: three bf16 products in every GEMM class (the shipped mode) against two products in the BACKWARD
classes (A operand rounded to bf16: the a_lo * b_hi product dropped), on fixed data.

Both arms start from the same weights, see the same 8 synthetic batches of 256 x 80 frames in the same order, draw the same
dropout masks (same seed, same call counter) and take the same negative-example branches (Python RNG reseeded). A third run
- three products, ANOTHER dropout seed - gives the scale of ordinary run-to-run variation to read the A/B difference against.

    python tools/loss_curve_ab.py [--steps 200] > profiles/round3_loss_curve_ab.md        (GPU box, ~15 s)
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

BWD = ("dpre", "cond_wgrad", "cond_dgrad", "flow_pgrads", "enc_dwih", "enc_dwhh", "enc_bptt")


def run(hp, batches, steps, skip, seed):
    from argparse import Namespace
    import copy
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    m = LetsFaceItGlow(Namespace(**copy.deepcopy(hp)))
    dev = batches[0]["p1_face"].device
    m.to(dev).train()
    torch.manual_seed(seed)          # the dropout-mask stream is keyed on torch's seed + the engine's call counter
    random.seed(99)                  # the negative-example branch
    eng = m.seq_glow._ensure_engine(dev)
    eng.backward_products = 3     # the arms differ by their explicit pass_skip entries only
    eng.pass_skip = dict(skip)
    losses = []
    for i in range(steps):
        losses.append(m.fused_training_step(batches[i % len(batches)], float(hp["lr"])))
    out = torch.stack([l.reshape(()) for l in losses]).double().cpu()
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
    b, pb = run(hp, batches, args.steps, {c: 1 for c in BWD}, 1234)
    c, pc = run(hp, batches, args.steps, {}, 4321)
    a2, _ = run(hp, batches, args.steps, {}, 1234)
    print("# Loss curve, 3 bf16 products everywhere vs 2 products in the backward GEMM classes\n")
    print("final_model.yaml at BASELINE's synthetic dims, batch 256, T = 80, lr %g, Adam + clip 20, %d steps over 8 fixed batches; "
          "per-step loss = the step's own training loss (negative-example steps included: loss x -0.1). "
          "`control` = three products with another dropout seed. The three-product arm repeated: bit-identical = %s.\n"
          % (float(hp["lr"]), args.steps, bool(torch.equal(a, a2))))
    print("| steps | loss 3 products | loss 2 products (bwd) | |A/B difference| | |control difference| (other dropout seed) |")
    print("|---|---|---|---|---|")
    for lo in range(0, args.steps, max(1, args.steps // 10)):
        hi = min(args.steps, lo + max(1, args.steps // 10))
        sl = slice(lo, hi)
        pos = a[sl] > -1e30
        print("| %d-%d | %.4f | %.4f | mean %.2e, max %.2e | mean %.2e, max %.2e |"
              % (lo, hi - 1, float(a[sl][pos].mean()), float(b[sl][pos].mean()), float((a[sl] - b[sl]).abs().mean()),
                 float((a[sl] - b[sl]).abs().max()), float((a[sl] - c[sl]).abs().mean()), float((a[sl] - c[sl]).abs().max())))
    dn = float((pa - pb).norm() / pa.norm())
    dc = float((pa - pc).norm() / pa.norm())
    print("\nParameters after %d steps: relative L2 distance 3-product vs 2-product arm %.3e; vs the other-seed control %.3e "
          "(movement from the initial weights is the same order as the control distance x 10 .. 100)." % (args.steps, dn, dc))
    print("\nVerdict: the A/B difference is %s the run-to-run variation of the control."
          % ("within" if float((a - b).abs().mean()) <= float((a - c).abs().mean()) else "ABOVE"))


if __name__ == "__main__":
    main()
