"""How many bf16 passes does each GEMM class of the training step need?

The engine's bf16x3 mode splits every fp32 GEMM operand into bf16 hi + lo and issues three products per k-step
(a_lo b_hi + a_hi b_lo + a_hi b_hi, the a_lo b_lo term of 2^-16 relative dropped). lfi_gemm_desc.precision bits 8 / 9 drop
the a_lo b_hi / a_hi b_lo product of ONE GEMM class at a time (GlowEngine.pass_skip), which is exactly what a 2-pass
(one operand rounded to bf16) or 1-pass (plain bf16 operands, fp32 accumulate) kernel would compute. For each class and
pass count this prints, at final_model.yaml widths (K = 16, H = 128, D = 512, 256/128/256-wide GRU windows, C = 50, S = 27):
  per-frame NLL max relative error vs the fp64 oracle (north_star gate: 1e-4),
  relative L2 error of the whole gradient and of the worst parameter tensor (tests/test_gpu_parity.py gate in bf16x3 mode: 2e-3).

    python tools/precision_sweep.py > profiles/precision_sweep.md      (GPU box; ~2 min)
    python tools/precision_sweep.py --batch 256 --seq-len 80 --bwd-only > profiles/precision_sweep_b256.md
        (the backward classes at the BENCHMARK's own size, 14 336 frames behind every weight row instead of
         96; the fp64 oracle's forward + backward at that size takes 1-2 minutes of host time, once)
Not covered (their bf16x3 products live inside the recurrence kernels, not in the GEMM library): the window encoders'
h W_hh^T products and the flow cells' recurrent / LinearZeros / invconv products.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

CLASSES = [("enc_xproj", "x W_ih^T of the window encoders (B T x 3 hid x in)", "fwd"),
           ("cond_fwd", "cond_transform forward, all 16 flow steps (F x 8192 x 890)", "fwd"),
           ("gic", "c W_ih[:, Ch:]^T + b_ih per flow step (F x 384 x 512, batch 16)", "fwd"),
           ("dpre", "d pre-activation of cond_transform (F x 512 x 384, batch 16)", "bwd"),
           ("cond_wgrad", "cond_transform weight gradient (8192 x 896 x F)", "bwd"),
           ("cond_dgrad", "gradient of the feature matrix (F x 640 x 8192)", "bwd"),
           ("flow_pgrads", "flow weight gradients: W_ih, W_hh, LinearZeros, invconv (long-K split-K products)", "bwd"),
           ("enc_dwih", "window encoders' dW_ih (3 hid x in x B T)", "bwd"),
           ("enc_dwhh", "window encoders' dW_hh (3 hid x hid x 23 F)", "bwd"),
           ("enc_bptt", "window encoders' BPTT recurrence d gates x W_hh (inside the fused kernel; only 'A rounded' exists, and only "
                        "together with enc_dwhh 'A rounded': the gradient stash between them is then bf16)", "bwd")]
MODES = [(0, "3 (a_lo b_hi + a_hi b_lo + a_hi b_hi)"), (1, "2, A rounded to bf16"), (2, "2, B rounded to bf16"), (3, "1 (plain bf16)")]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=36)
    ap.add_argument("--bwd-only", action="store_true", help="only the backward GEMM classes (they leave the NLL untouched)")
    args = ap.parse_args()
    from argparse import Namespace
    from helpers import max_rel
    from oracle import seqglow_oracle as oracle
    from test_gpu_parity import final_model_hparams, perturbed_model, to_dev
    dev = torch.device("cuda:0")
    hp = final_model_hparams(50, 27, K=16)
    m, sd = perturbed_model(hp, dev)
    m.precision = "bf16x3"
    m.train()
    B, T = args.batch, args.seq_len
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=17)
    N = T - 24
    g = torch.Generator().manual_seed(6)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    m.injected_masks = masks
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s"))) for k, v in sd.items()}
    _, oloss, onll = oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()}, {k: v.double() for k, v in masks.items()})
    oloss.sum().backward()
    ref = {k: v.grad for k, v in sdg.items() if v.grad is not None}
    total = float(torch.sqrt(sum((v ** 2).sum() for v in ref.values())))
    dbatch = to_dev(batch, dev)

    def run(skip):
        eng = m._ensure_engine(dev)
        eng.backward_products = 3     # only the class under test is reduced
        eng.pass_skip = dict(skip)
        m.zero_grad(set_to_none=True)
        _, loss, losses = m(dbatch)
        loss.sum().backward()
        eng.pass_skip = {}
        err = max_rel(torch.stack(losses), onll.detach(), floor=1.0)
        num, worst, raw = 0.0, ("", 0.0), 0.0
        for name, p in m.named_parameters():
            d = p.grad.double().cpu() - ref[name]
            num += float((d ** 2).sum())
            rel = float(d.norm() / max(float(ref[name].norm()), 1e-3 * total))
            raw = max(raw, float(d.norm() / max(float(ref[name].norm()), 1e-30)))
            if rel > worst[1]:
                worst = (name, rel)
        return err, (num ** 0.5) / total, (worst[0], worst[1], raw)

    print("# bf16 passes per GEMM class: error of one training forward + backward against the fp64 oracle\n")
    print("final_model.yaml widths (K = 16), C = 50 / S = 27, batch %d, T = %d (%d timesteps), dropout masks injected; engine mode "
          "bf16x3 with ONE class at a time reduced. Gates: per-frame NLL 1e-4 relative (north_star); gradient tensors 2e-3 "
          "relative L2 (tests/test_gpu_parity.py, bf16x3 mode). Generated by tools/precision_sweep.py on an MI355X.\n" % (B, T, N))
    print("| GEMM class | passes | per-frame NLL max rel err | whole-gradient rel L2 | worst tensor rel L2 (norm floored at 1e-3 of the whole gradient) | worst tensor | worst tensor rel L2, no floor |")
    print("|---|---|---|---|---|---|---|")
    err, gl2, worst = run({})
    print("| (all classes) | 3 | %.2e | %.2e | %.2e | %s | %.2e |" % (err, gl2, worst[1], worst[0], worst[2]))
    shipped = {"enc_dwhh": 3}   # the mildest reduction tried as a default (reverted: see the last column)
    err, gl2, worst = run(shipped)
    print("| tried as default, reverted: enc_dwhh with 1 product, everything else 3 | mixed | %.2e | %.2e | %.2e | %s | %.2e |"
          % (err, gl2, worst[1], worst[0], worst[2]))
    for cls, what, direction in CLASSES:
        if args.bwd_only and direction != "bwd":
            continue
        for bits, label in (MODES[1:2] if cls == "enc_bptt" else MODES[1:]):
            err, gl2, worst = run({cls: bits, "enc_dwhh": 1} if cls == "enc_bptt" else {cls: bits})
            print("| %s: %s | %s | %.2e | %.2e | %.2e | %s | %.2e |" % (cls, what, label, err, gl2, worst[1], worst[0], worst[2]))
    fwd = {c: 3 for c, _, d in CLASSES if d == "fwd"}
    allc = {c: 3 for c, _, _ in CLASSES}
    if args.bwd_only:
        bwd = [c for c, _, d in CLASSES if d == "bwd"]
        for bits, label in MODES[1:]:
            err, gl2, worst = run({c: bits for c in bwd})
            print("| EVERY backward class | %s | %.2e | %.2e | %.2e | %s | %.2e |" % (label, err, gl2, worst[1], worst[0], worst[2]))
        return
    for label, skip in (("every forward class", fwd), ("every class", allc)):
        err, gl2, worst = run(skip)
        print("| %s | 1 (plain bf16) | %.2e | %.2e | %.2e | %s | %.2e |" % (label, err, gl2, worst[1], worst[0], worst[2]))
    # the recurrences' own products cannot be switched per class; what a 2-product window-encoder recurrence (W_hh rounded to
    # bf16: its lo plane never loaded, h still hi + lo) would cost is measured by rounding the encoders' W_hh on the device
    # copy only (the lo plane is then zero and the three-product kernel computes exactly that), oracle on the original weights
    saved = {}
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("encoder.weight_hh_l0"):
                saved[name] = p.detach().clone()
                p.copy_(p.to(torch.bfloat16).to(torch.float32))
    err, gl2, worst = run({})
    print("| window encoders' h W_hh^T recurrence (forward AND backward kernels) | 2, W_hh rounded to bf16 | %.2e | %.2e | %.2e | %s | %.2e |"
          % (err, gl2, worst[1], worst[0], worst[2]))
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name in saved:
                p.copy_(saved[name])
    fwd2 = {c: 2 for c in fwd}
    all2 = {c: 2 for c in allc}
    for label, skip in (("every forward class", fwd2), ("every class", all2)):
        err, gl2, worst = run(skip)
        print("| %s | 2, B rounded to bf16 | %.2e | %.2e | %.2e | %s | %.2e |" % (label, err, gl2, worst[1], worst[0], worst[2]))


if __name__ == "__main__":
    main()
