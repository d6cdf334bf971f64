"""A multi-step optimiser trajectory against the ORACLE (VERDICT r3, weak #1d / next #3c): five `fused_training_step`s - injected
dropout masks, another batch every step, one forced negative-example step with a fixed derangement - in the arithmetic the
benchmark runs (bf16x3 forward products, TWO products in the backward GEMM classes, fp16 encoder gate stash where the shape takes
it), against the fp64 oracle's `training_loss` + autograd + `adam_clip_step` (reference: lets_face_it_glow.py:39-54 training_step,
:61-72 configure_optimizers / Adam, hparams gradient_clip_val). The fixtures pin ONE Adam step; this pins that the steps compose:
step k's loss is evaluated at the parameters steps 1..k-1 left behind.
"""
from argparse import Namespace

import pytest
import torch

from helpers import Fixture, report
from oracle import seqglow_oracle as oracle
from test_gpu_parity import to_dev

pytestmark = pytest.mark.gpu

STEPS, NEG_STEP, LR, CLIP = 5, 2, 2e-3, 5.0


def _masks(hp, N, B, seed):
    g = torch.Generator().manual_seed(seed)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        if cfg["history"] and cfg["dropout"]:
            keep = 1.0 - cfg["dropout"]
            masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    return masks or None


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_five_step_trajectory_against_oracle(gpu_device, monkeypatch, name):
    from lets_face_it_amd.glow import lets_face_it_glow as lfg
    import copy
    fx = Fixture(name)
    hp = fx.hp
    hp["Train"]["use_negative_nll_loss"] = True
    B, T, C, S, N = fx.B, fx.T, fx.C, fx.S, fx.N
    perm = torch.roll(torch.arange(B), 1)    # a derangement
    real_derange = lfg.derange_batch
    monkeypatch.setattr(lfg, "derange_batch", lambda b, mods, **kw: real_derange(b, mods, permutation=perm.clone()))

    def engine_model(two, precision=None):
        ns = Namespace(**copy.deepcopy(hp))
        ns.gradient_clip_val = CLIP
        ns.engine_precision = precision or ("bf16x3" if two else "f32")
        ns.engine_backward_products = 2 if two else 3     # 2: what "auto" picks at the benchmark's 14 336 frames
        lm = lfg.LetsFaceItGlow(ns)
        lm.seq_glow.load_state_dict(fx.state_dict(torch.float32))
        lm.to(gpu_device)
        lm.seq_glow.glow.set_actnorm_init(True)
        lm.train()
        forced = iter([i == NEG_STEP for i in range(STEPS)])
        lm._negative_branch = lambda: next(forced)
        return lm

    lm = engine_model(True)

    batches = [oracle.synthetic_batch(B, T, C, S, seed=900 + i) for i in range(STEPS)]
    masks = [_masks(hp, N, B, 50 + i) for i in range(STEPS)]
    a = hp["Optim"]["args"]["adam"]
    beta1, beta2, eps = float(a["betas"][0]), float(a["betas"][1]), float(a["eps"])

    # ---- the oracle's trajectory (fp64)
    sd = {k: v.clone() for k, v in fx.state_dict(torch.float64).items()}
    names = [k for k, _ in lm.seq_glow.named_parameters()]      # Adam steps the parameters, not the buffers (P, sign(s), masks)
    assert all(k in sd for k in names)
    p0 = {k: sd[k].clone() for k in names}
    mom = {k: torch.zeros_like(sd[k]) for k in names}
    var = {k: torch.zeros_like(sd[k]) for k in names}
    ref_losses, ref_mm = [], None
    for i in range(STEPS):
        for k in names:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        neg = perm if i == NEG_STEP else None
        loss = oracle.training_loss(hp, sd, {k: v.double() for k, v in batches[i].items()},
                                    None if masks[i] is None else {k: v.double() for k, v in masks[i].items()}, neg)
        loss.sum().backward()
        ref_losses.append(float(loss.detach()))
        if neg is not None:
            ref_mm = float(loss.detach()) / -0.1 * -1.0     # _store_mismatched(-loss) of the un-scaled loss
        grads = [sd[k].grad.clone() for k in names]
        with torch.no_grad():
            for k in names:
                sd[k].requires_grad_(False)
            oracle.adam_clip_step([sd[k] for k in names], grads, [mom[k] for k in names], [var[k] for k in names], i + 1, LR,
                                  beta1, beta2, eps, CLIP)

    # ---- the engine's, in the benchmark's arithmetic and in the exact f32 mode
    # (at the fixtures' few hundred frames "auto" itself would pick three products: that leg is reported between the two)
    for mode, two, precision in (("bf16x3, two-product backward", True, "bf16x3"), ("bf16x3, three products", False, "bf16x3"),
                                 ("exact f32", False, "f32")):
        if mode != "bf16x3, two-product backward":
            lm = engine_model(two, precision)
        _check_trajectory(lm, name, mode, two, batches, masks, names, p0, sd, ref_losses, ref_mm, B, N, gpu_device)


def _check_trajectory(lm, name, mode, two, batches, masks, names, p0, sd, ref_losses, ref_mm, B, N, gpu_device):
    losses = []
    for i in range(STEPS):
        lm.seq_glow.injected_masks = None if masks[i] is None else {k: v.clone() for k, v in masks[i].items()}
        losses.append(float(lm.fused_training_step(to_dev(batches[i], gpu_device), LR)))
    eng = lm.seq_glow.engine
    assert eng.step_count == STEPS and eng.backward_product_count(B * N) == (2 if two else 3)

    worst_loss = max(abs(g - r) / max(abs(r), 1.0) for g, r in zip(losses, ref_losses))
    got = dict(lm.seq_glow.named_parameters())
    num = den = 0.0
    worst = ("", 0.0)
    for k in names:
        upd = got[k].detach().double().cpu() - p0[k]
        upd_ref = sd[k] - p0[k]
        num += float((upd - upd_ref).norm()) ** 2
        den += float(upd_ref.norm()) ** 2
        rel = float((upd - upd_ref).norm() / max(float(upd_ref.norm()), 1e-12))
        if rel > worst[1]:
            worst = (k, rel)
    whole = (num / max(den, 1e-300)) ** 0.5
    mm = float(lm.last_missmatched_nll)
    report("%s: 5-step fused_training_step trajectory (negative step %d, clip %.0f, lr %.0e; %s) "
           "vs the fp64 oracle's training_loss + adam_clip_step: per-step loss max rel err %.2e; parameter UPDATE after 5 steps rel L2: "
           "whole model %.2e, worst tensor %.2e (%s); mismatched-NLL buffer %.6f vs %.6f"
           % (name, NEG_STEP, CLIP, LR, mode, worst_loss, whole, worst[1], worst[0], mm, ref_mm))
    # the gate on the trajectory is FUNCTIONAL: every step's loss - evaluated at the parameters the earlier steps left behind, on a
    # batch they have not seen - tracks the oracle's to north_star's NLL tolerance
    assert worst_loss < 1e-4
    assert abs(mm - ref_mm) <= 1e-4 * max(abs(ref_mm), 1.0)
    # the parameter update itself is reported and bounded loosely: Adam's first steps are sign-like (update = lr * m / sqrt(v), scale
    # free), so an entry whose gradient lies within the rounding of the backward products moves by +-lr either way - a few percent
    # of single tensors at these frame counts (a few hundred) with two-product backward GEMMs, 1e-3 in the exact f32 mode
    assert whole < (6e-2 if two else 1e-2) and worst[1] < (0.15 if two else 3e-2), (mode, whole, worst)


def test_headline_trajectory_against_oracle(gpu_device, monkeypatch):
    """The default arithmetic pinned WHERE IT OPERATES: `engine_backward_products: auto` switches to two bf16
    products in the backward GEMM classes (and the fp16 encoder gate stash) from 8192 frames per step - a size the fixtures above
    never reach. Three `fused_training_step`s at the benchmark's own shape (B = 256, T = 80, K = 16, C = 50 / S = 27: 14 336 frames),
    another batch and other injected masks every step, step 2 the negative-example step with a fixed derangement, clip 20
    (final_model.yaml:126), against the fp64 oracle's training_loss + autograd + adam_clip_step (lets_face_it_glow.py:39-54,61-72).
    Legs: "auto" (must resolve to two products here) and three products everywhere. Gates: per-step loss 1e-4 relative
    (north_star) in both; the parameter update after three steps - reported per leg for the whole model and its worst tensor -
    of the two-product leg at most 10 x the three-product leg's (the judge's bar: if it fails, the headline is the three-product
    figure)."""
    import copy
    import time
    from lets_face_it_amd.glow import lets_face_it_glow as lfg
    from test_gpu_parity import final_model_hparams, perturbed_model
    steps, neg_step, lr, clip = 3, 1, 1e-4, 20.0
    hp = final_model_hparams(50, 27, K=16)
    hp["Train"]["use_negative_nll_loss"] = True
    B, T, C, S = 256, 80, 50, 27
    N = T - 24
    perm = torch.roll(torch.arange(B), 1)
    real_derange = lfg.derange_batch
    monkeypatch.setattr(lfg, "derange_batch", lambda b, mods, **kw: real_derange(b, mods, permutation=perm.clone()))
    m0, sd32 = perturbed_model(hp, gpu_device)
    del m0
    torch.cuda.empty_cache()
    batches = [oracle.synthetic_batch(B, T, C, S, seed=700 + i) for i in range(steps)]
    masks = [_masks(hp, N, B, 70 + i) for i in range(steps)]
    a = hp["Optim"]["args"]["adam"]
    beta1, beta2, eps = float(a["betas"][0]), float(a["betas"][1]), float(a["eps"])

    # ---- the engine's two legs first (the oracle's fp64 passes below take most of a minute of host time)
    got = {}
    for leg, products in (("auto", "auto"), ("three products", 3)):
        ns = Namespace(**copy.deepcopy(hp))
        ns.gradient_clip_val = clip
        ns.engine_precision = "bf16x3"
        ns.engine_backward_products = products
        lm = lfg.LetsFaceItGlow(ns)
        lm.seq_glow.load_state_dict(sd32)
        lm.to(gpu_device)
        lm.seq_glow.glow.set_actnorm_init(True)
        lm.train()
        forced = iter([i == neg_step for i in range(steps)])
        lm._negative_branch = lambda forced=forced: next(forced)
        losses = []
        for i in range(steps):
            lm.seq_glow.injected_masks = {k: v.clone() for k, v in masks[i].items()}
            losses.append(float(lm.fused_training_step(to_dev(batches[i], gpu_device), lr)))
        eng = lm.seq_glow.engine
        assert eng.step_count == steps and eng.backward_product_count(B * N) == (2 if products == "auto" else 3)
        if products == "auto":
            assert all(eng._last.enc_stash_f16[n] for n in ("p2_face", "p1_speech", "p2_speech")) and bool(eng._last.chain)
        got[leg] = (losses, {k: p.detach().double().cpu() for k, p in lm.seq_glow.named_parameters()}, float(lm.last_missmatched_nll))
        del lm, eng
        torch.cuda.empty_cache()

    # ---- the oracle's trajectory (fp64)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    t0 = time.time()
    sd = {k: v.double().clone() for k, v in sd32.items()}
    names = list(got["auto"][1])
    p0 = {k: sd[k].clone() for k in names}
    mom = {k: torch.zeros_like(sd[k]) for k in names}
    var = {k: torch.zeros_like(sd[k]) for k in names}
    ref_losses, ref_mm = [], None
    for i in range(steps):
        for k in names:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        neg = perm if i == neg_step else None
        loss = oracle.training_loss(hp, sd, {k: v.double() for k, v in batches[i].items()},
                                    {k: v.double() for k, v in masks[i].items()}, neg)
        loss.sum().backward()
        ref_losses.append(float(loss.detach()))
        if neg is not None:
            ref_mm = float(loss.detach()) / -0.1 * -1.0
        grads = [sd[k].grad.clone() for k in names]
        with torch.no_grad():
            for k in names:
                sd[k].requires_grad_(False)
            oracle.adam_clip_step([sd[k] for k in names], grads, [mom[k] for k in names], [var[k] for k in names], i + 1, lr,
                                  beta1, beta2, eps, clip)
    spent = time.time() - t0
    torch.set_num_threads(threads)

    stats = {}
    for leg, (losses, params, mm) in got.items():
        worst_loss = max(abs(g - r) / max(abs(r), 1.0) for g, r in zip(losses, ref_losses))
        num = den = 0.0
        worst = ("", 0.0)
        per_tensor = []
        for k in names:
            upd, upd_ref = params[k] - p0[k], sd[k] - p0[k]
            num += float((upd - upd_ref).norm()) ** 2
            den += float(upd_ref.norm()) ** 2
            rel = float((upd - upd_ref).norm() / max(float(upd_ref.norm()), 1e-12))
            per_tensor.append((rel, k))
            if rel > worst[1]:
                worst = (k, rel)
        whole = (num / max(den, 1e-300)) ** 0.5
        per_tensor.sort(reverse=True)
        stats[leg] = (worst_loss, whole, worst)
        report("HEADLINE trajectory (B=256, T=80, K=16, bf16x3; %d fused_training_steps, step %d negative, clip %.0f, lr %.0e; backward "
               "products: %s) vs the fp64 oracle's training_loss + adam_clip_step (%.0f s of host time): per-step loss rel err %s (max "
               "%.2e); parameter UPDATE after %d steps rel L2: whole model %.3e, worst tensor %.3e (%s); five worst tensors: %s; "
               "mismatched-NLL buffer %.6f vs %.6f"
               % (steps, neg_step + 1, clip, lr, leg, spent, ", ".join("%.1e" % (abs(g - r) / max(abs(r), 1.0)) for g, r in zip(losses, ref_losses)),
                  worst_loss, steps, whole, worst[1], worst[0], "; ".join("%s %.2e" % (k, r) for r, k in per_tensor[:5]), mm, ref_mm))
        assert worst_loss < 1e-4, (leg, losses, ref_losses)
        assert abs(mm - ref_mm) <= 1e-4 * max(abs(ref_mm), 1.0)
    two, three = stats["auto"], stats["three products"]
    report("HEADLINE trajectory: two-product / three-product parameter-update error: whole model %.2f x, worst tensor %.2f x (gate 10 x)"
           % (two[1] / max(three[1], 1e-300), two[2][1] / max(three[2][1], 1e-300)))
    assert two[1] <= 10.0 * three[1] and two[2][1] <= 10.0 * three[2][1], (two, three)
