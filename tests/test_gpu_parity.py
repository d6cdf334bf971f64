"""GPU parity: the HIP path (through SeqGlow / LetsFaceItGlow and the C ABI) against the golden vectors captured
from the reference (fp64) and against the CPU oracle on fresh seeded inputs.

Tolerances (BASELINE.json north_star): per-frame NLL within 1e-4 relative; sampled frames within 1e-5 absolute at
fixed injected noise, both against the fp64 reference. Every fixture also carries the reference's OWN fp32 result on the
same noise (`infer/out_ref_fp32`): its distance from fp64 (8e-7 ... 2.4e-6 on these fixtures) is printed next to ours.
Measured errors go through helpers.report(): with LFI_PARITY_REPORT=<file> they are appended to that file
(profiles/parity_report_r02.txt is the final GPU run's).
"""
import os
from argparse import Namespace

import pytest
import torch

from helpers import FIXTURES, Fixture, max_rel, rel_err, report
from oracle import seqglow_oracle as oracle

pytestmark = pytest.mark.gpu

GPU_FIXTURES = FIXTURES   # every fixture, "dense" (LU_decomposed: false) included


def build(fx, device, train=False, precision="f32"):
    from lets_face_it_amd.glow.models import SeqGlow
    hp = Namespace(**fx.hp)
    m = SeqGlow(hp)
    m.precision = precision
    missing = m.load_state_dict(fx.state_dict(torch.float32), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.to(device)
    m.glow.set_actnorm_init(True)
    m.train(train)
    return m


def to_dev(d, device):
    return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in d.items()}


@pytest.fixture(scope="module", params=GPU_FIXTURES)
def fx(request):
    return Fixture(request.param)


def test_eval_forward_matches_reference(fx, gpu_device):
    m = build(fx, gpu_device)
    with torch.no_grad():
        z_seq, loss, losses = m(to_dev(fx.batch(), gpu_device))
    assert len(z_seq) == fx.N and len(losses) == fx.N and loss.shape == (1,)
    assert all(not t.is_cuda for t in losses)
    nll = torch.stack(losses)
    err = max_rel(nll, fx.get("eval/nll"), floor=1.0)
    report("%s: per-frame NLL max rel err vs fp64 reference %.3e (reference's own fp32: %.3e)" %
           (fx.name, err, max_rel(fx.get("eval/nll_ref_fp32"), fx.get("eval/nll"), floor=1.0)))
    assert err < 1e-4
    assert rel_err(torch.stack(z_seq), fx.get("eval/z")) < 1e-5
    assert rel_err(loss, fx.get("eval/loss")) < 1e-5


def test_train_forward_backward_matches_reference(fx, gpu_device):
    m = build(fx, gpu_device, train=True)
    m.injected_masks = fx.masks(torch.float32)
    z_seq, loss, losses = m(to_dev(fx.batch(), gpu_device))
    assert max_rel(torch.stack(losses), fx.get("train/nll"), floor=1.0) < 1e-4
    loss.sum().backward()
    grads = fx.group("grad/")
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        err = rel_err(p.grad, grads[name])
        rel = (p.grad.double().cpu() - grads[name]).norm() / max(float(grads[name].norm()), 1e-12)
        if float(rel) > worst[1]:
            worst = (name, float(rel))
        assert err < 1e-4 and float(rel) < 1e-3, (name, err, float(rel))
    report("%s: worst gradient relative L2 error %.3e (%s)" % (fx.name, worst[1], worst[0]))


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_fused_training_step_matches_reference_adam(fx, gpu_device, precision):
    """Native path: forward + backward + clip + Adam in the engine == clip_grad_norm_ + torch.optim.Adam on the reference, in the
    exact-product mode and in the default bf16x3 arithmetic (VERDICT r4 weak #1c; at the fixtures' few hundred frames "auto" keeps three
    products in the backward classes)."""
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    import copy
    hp = Namespace(**copy.deepcopy(fx.hp))
    hp.Train["use_negative_nll_loss"] = False
    hp.gradient_clip_val = float(fx.get("adam/clip"))
    hp.engine_precision = precision   # Adam's first step divides by |g| + 1e-8: the exact-product mode keeps the test about Adam
    lm = LetsFaceItGlow(hp)
    lm.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    lm.to(gpu_device)
    lm.seq_glow.glow.set_actnorm_init(True)
    lm.train()
    lm.seq_glow.injected_masks = fx.masks(torch.float32)
    loss = lm.fused_training_step(to_dev(fx.batch(), gpu_device), float(fx.get("adam/lr")))
    assert rel_err(loss, fx.get("train/loss")) < (1e-5 if precision == "f32" else 1e-4)
    gn = lm.seq_glow.engine.grad_norm()
    assert abs(gn - float(fx.get("adam/grad_norm"))) < (1e-4 if precision == "f32" else 1e-3) * float(fx.get("adam/grad_norm"))
    n_sat = n_sat_bad = 0
    for name, p in lm.seq_glow.named_parameters():
        ref = fx.get("adam/" + name)
        # first Adam step moves every weight by ~lr: compare the UPDATE, not the weight
        before = fx.get("sd/" + name).double()
        upd, upd_ref = p.detach().double().cpu() - before, ref - before
        lr = float(fx.get("adam/lr"))
        # step 1 of Adam is lr * g / (|g| + eps): +-lr wherever |g| >> eps = 1e-8. Entries whose (clipped) gradient is within
        # a decade of eps are not saturated and amplify the fp32 rounding of g by 1 / eps: they get a looser bound
        sat = upd_ref.abs() > 0.98 * lr
        err = (upd - upd_ref).abs()
        if precision == "f32":
            assert (err[sat].max() if sat.any() else 0.0) < 2e-2 * lr + 1e-7, name
            assert (err[~sat].max() if (~sat).any() else 0.0) < 0.25 * lr, name
        else:
            # three bf16 products put 2^-16 relative on the gradient: a saturated entry moves by +-lr whatever that does to |g| -
            # unless the rounding reaches the entry's own size (|g| of the order of 1e-5 of its tensor's scale), where the step is
            # sign-like; such entries are counted, not bounded one by one
            n_sat += int(sat.sum())
            n_sat_bad += int((err[sat] > 2e-2 * lr + 1e-7).sum())
            assert float(err.max()) <= 2.0 * lr * (1 + 1e-6), name
    if precision != "f32":
        report("%s: fused Adam step in bf16x3: %d of %d saturated entries off by more than 2 %% of lr" % (fx.name, n_sat_bad, n_sat))
        assert n_sat_bad <= 2e-3 * max(n_sat, 1) + 2


def test_negative_step_loss(fx, gpu_device):
    from lets_face_it_amd.glow.utils import derange_batch, get_mismatched_modalities
    m = build(fx, gpu_device)
    mods, _ = get_mismatched_modalities(Namespace(**fx.hp))
    mixed = derange_batch(to_dev(fx.batch(), gpu_device), mods, permutation=fx.get("neg/perm"))
    with torch.no_grad():
        _, loss, losses = m(mixed)
    assert max_rel(torch.stack(losses), fx.get("neg/nll"), floor=1.0) < 1e-4
    assert rel_err(loss * -0.1, fx.get("neg/loss")) < 1e-5


def test_inference_matches_reference(fx, gpu_device):
    m = build(fx, gpu_device)
    data = to_dev(fx.group("infer/data/"), gpu_device)
    noise = fx.get("infer/noise", torch.float32).to(gpu_device)
    out = m.inference(int(fx.get("infer/seq_len")), data, noise=noise)
    ref = fx.get("infer/out")
    assert tuple(out.shape) == tuple(ref.shape)
    err = (out.double().cpu() - ref).abs().max().item()
    own = (fx.get("infer/out_ref_fp32").double() - ref).abs().max().item()
    report("%s: sampled frames max abs err vs fp64 reference %.3e (reference's own fp32: %.3e; |x| <= %.1f)"
           % (fx.name, err, own, ref.abs().max()))
    assert err < 1e-5
    m.precision = "bf16x3"
    out3 = m.inference(int(fx.get("infer/seq_len")), data, noise=noise)
    err3 = (out3.double().cpu() - ref).abs().max().item()
    report("%s bf16x3: sampled frames max abs err vs fp64 reference %.3e" % (fx.name, err3))
    assert err3 < 1e-5


def test_invert_matches_reference(fx, gpu_device):
    m = build(fx, gpu_device)
    batch = to_dev(fx.batch(), gpu_device)
    with torch.no_grad():
        z_seq, loss, _ = m(batch)
    rec, bl = m.invert(z_seq, batch)
    x = batch["p1_face"][:, fx.start:].transpose(0, 1)
    assert rel_err(torch.stack(rec), x) < 1e-4
    # the reverse pass negates the log-det and keeps the prior term: fwd + bwd = -2 mean(log p(z)) / ln 2
    # (what MimicryLogger.test_invertability divides by the forward loss, mimicry_logger.py:241-251)
    logp = (-0.5 * (torch.stack(z_seq) ** 2 + oracle.LOG2PI)).sum(-1)
    assert rel_err(bl + loss, -2.0 * logp.mean().reshape(1) / oracle.LN2) < 1e-4
    if fx.has("invert/x"):
        assert rel_err(torch.stack(rec), fx.get("invert/x")) < 1e-4
        assert rel_err(bl, fx.get("invert/loss")) < 1e-4


@pytest.mark.parametrize("case", ["mid", "tiny_lstm", "final"])
def test_invert_single_launch_matches_cell_by_cell(gpu_device, monkeypatch, case):
    """SeqGlow.invert as ONE persistent launch (lfi_flow_seq_rev: workgroup (k, tile) walks the timesteps, tiles handed from step
    k + 1 to step k) runs the same reverse cells as N x Ks lfi_flow_step launches (LFI_INVERT_WALK=0): the reconstruction is
    bit-identical, the log-det agrees to the rounding of its sum over the flow steps (another order)."""
    if case == "final":
        hp = final_model_hparams(50, 27)
        mk = lambda: perturbed_model(hp, gpu_device)[0].eval()  # noqa: E731
        batch = to_dev(oracle.synthetic_batch(40, 24 + 9, 50, 27, seed=8), gpu_device)      # ragged last tile: 40 = 2 x 16 + 8
    else:
        fxm = Fixture(case)
        mk = lambda: build(fxm, gpu_device)  # noqa: E731
        batch = to_dev(fxm.batch(), gpu_device)
    outs = []
    for walk in ("1", "0"):
        monkeypatch.setenv("LFI_INVERT_WALK", walk)
        m = mk()
        with torch.no_grad():
            z_seq, loss, _ = m(batch)
        calls = []
        eng = m._ensure_engine(gpu_device)
        real = eng.L.lfi_flow_step
        eng.L.lfi_flow_step = lambda *a: (calls.append(1), real(*a))[1]
        try:
            rec, bl = m.invert(z_seq, batch)
        finally:
            eng.L.lfi_flow_step = real
        assert (len(calls) == 0) == (walk == "1"), (walk, len(calls))
        outs.append((torch.stack(rec), bl))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_err(outs[0][1], outs[1][1]) < 1e-6


def test_actnorm_data_dependent_init(fx, gpu_device):
    from lets_face_it_amd.glow.models import SeqGlow
    m = SeqGlow(Namespace(**fx.hp))
    m.load_state_dict(fx.state_dict(torch.float32))
    m.to(gpu_device)
    m.train()
    m.injected_masks = fx.masks(torch.float32, prefix="init/mask/")
    with torch.no_grad():
        _, _, losses = m(to_dev(fx.group("init/batch/"), gpu_device))
    assert m.glow.actnorm_inited()
    for k, layer in enumerate(m.glow.flow.layers):
        assert rel_err(layer.actnorm.bias, fx.get("init/glow.flow.layers.%d.actnorm.bias" % k)) < 1e-4
        assert rel_err(layer.actnorm.logs, fx.get("init/glow.flow.layers.%d.actnorm.logs" % k)) < 1e-4
    assert max_rel(torch.stack(losses), fx.get("init/nll"), floor=1.0) < 2e-4


def test_cpu_tensor_is_refused():
    fxl = Fixture("tiny")
    from lets_face_it_amd.glow.models import SeqGlow
    m = SeqGlow(Namespace(**fxl.hp))
    with pytest.raises(RuntimeError, match="GPU only"):
        m(fxl.batch(torch.float32))


# ------------------------------------------------------------------ bf16x3 mode stays inside the same tolerances
def test_bf16x3_mode_forward_backward(fx, gpu_device):
    m = build(fx, gpu_device, train=True, precision="bf16x3")
    m.injected_masks = fx.masks(torch.float32)
    z_seq, loss, losses = m(to_dev(fx.batch(), gpu_device))
    err = max_rel(torch.stack(losses), fx.get("train/nll"), floor=1.0)
    loss.sum().backward()
    grads = fx.group("grad/")
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        # relative to the tensor's own gradient norm, floored at 1e-3 of the whole gradient (a tensor that small does not
        # move under the global-norm clip + Adam, and its entries sit at the 2^-16 noise floor of the split products)
        diff = p.grad.double().cpu() - grads[name]
        rel = float(diff.norm() / max(float(grads[name].norm()), 1e-3 * total))
        if rel > 2e-3 and "cond_transform.0.weight" in name:
            # LeakyReLU is not smooth at 0: a pre-activation within the 2^-16 noise of the split products of 0 can land on the
            # other side of the kink than in the fp64 reference, which changes ONE (frame, unit) term of this unit's weight row
            # by a factor 100 (seen: one row at 1.7 %, every other row at 1e-5). Allow one such row per tensor.
            rows = diff.norm(dim=1)
            diff = diff.clone()
            diff[int(rows.argmax())] = 0
            rel = float(diff.norm() / max(float(grads[name].norm()), 1e-3 * total))
        if rel > worst[1]:
            worst = (name, rel)
    report("%s bf16x3: per-frame NLL max rel err %.3e, worst gradient rel L2 %.3e (%s)" % (fx.name, err, worst[1], worst[0]))
    assert err < 1e-4 and worst[1] < 2e-3


# ------------------------------------------------------------------ full-width model against the oracle
def final_model_hparams(C=50, S=27, K=16):
    hp = Fixture("mid").hp
    c = hp["Conditioning"]
    c["cond_dim"] = 512
    c["p1_face"]["dim"] = C
    c["p2_face"].update(dim=C, hidden_dim=256)
    c["p1_speech"]["hidden_dim"] = 128
    c["p2_speech"]["hidden_dim"] = 256
    hp["Data"]["speech_dim"] = S
    hp["Glow"].update(K=K, L=1, hidden_channels=128)
    hp["Train"]["seq_len"] = 80
    return hp


def perturbed_model(hp, device, seed=1234):
    from lets_face_it_amd.glow.models import SeqGlow
    import numpy as np
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = SeqGlow(Namespace(**hp))
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "final_linear" in name:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            elif "actnorm" in name:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(device)
    m.glow.set_actnorm_init(True)
    return m, sd


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("C,S", [(50, 27), (56, 30)])
def test_final_model_width_nll_against_oracle(gpu_device, C, S, precision):
    """final_model.yaml widths (K=16, H=128, D=512, GRU 256/128/256), T=80, reduced batch so the CPU oracle takes seconds."""
    hp = final_model_hparams(C, S)
    m, sd = perturbed_model(hp, gpu_device)
    m.precision = precision
    m.eval()
    batch = oracle.synthetic_batch(12, 80, C, S, seed=1234)
    with torch.no_grad():
        z_seq, loss, losses = m(to_dev(batch, gpu_device))
    z64, loss64, nll64 = oracle.seqglow_forward(hp, {k: v.double() for k, v in sd.items()},
                                                {k: v.double() for k, v in batch.items()})
    err = max_rel(torch.stack(losses), nll64, floor=1.0)
    report("final_model C=%d S=%d %s: per-frame NLL max rel err vs fp64 oracle %.3e" % (C, S, precision, err))
    assert err < 1e-4
    assert rel_err(torch.stack(z_seq), z64) < 1e-4


def test_final_model_width_gradients_against_oracle(gpu_device):
    hp = final_model_hparams(50, 27, K=4)
    m, sd = perturbed_model(hp, gpu_device)
    m.train()
    B, T = 6, 40
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=99)
    N = T - 24
    g = torch.Generator().manual_seed(5)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    m.injected_masks = masks
    _, loss, _ = m(to_dev(batch, gpu_device))
    loss.sum().backward()
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd.items()}
    oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()},
                           {k: v.double() for k, v in masks.items()})[1].sum().backward()
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        ref = sdg[name].grad
        rel = float((p.grad.double().cpu() - ref).norm() / max(float(ref.norm()), 1e-12))
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < 2e-3, (name, rel)
    report("final-width gradients (K=4, B=6, T=40): worst relative L2 error %.3e (%s)" % worst[::-1])


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_full_model_k16_gradients_against_oracle(gpu_device, precision):
    """The whole final_model.yaml network (K=16, H=128, D=512, 256/128/256-wide GRU windows, C=50/S=27) at a batch and
    length the fp64 oracle's autograd walks in seconds: per-frame NLL and EVERY parameter gradient, with dropout masks."""
    hp = final_model_hparams(50, 27, K=16)
    m, sd = perturbed_model(hp, gpu_device)
    m.precision = precision
    m.train()
    B, T = 5, 32
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=17)
    N = T - 24
    g = torch.Generator().manual_seed(6)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    m.injected_masks = masks
    _, loss, losses = m(to_dev(batch, gpu_device))
    loss.sum().backward()
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd.items()}
    _, oloss, onll = oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()},
                                            {k: v.double() for k, v in masks.items()})
    oloss.sum().backward()
    err = max_rel(torch.stack(losses), onll.detach(), floor=1.0)
    total = float(torch.sqrt(sum((v.grad ** 2).sum() for v in sdg.values() if v.grad is not None)))
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        ref = sdg[name].grad
        diff = p.grad.double().cpu() - ref
        rel = float(diff.norm() / max(float(ref.norm()), 1e-3 * total))
        if rel > 1e-3 and precision == "bf16x3" and "cond_transform.0.weight" in name:
            # LeakyReLU's kink (see test_bf16x3_mode_forward_backward): with only 40 frames behind each weight row, ONE
            # (frame, unit) pre-activation within the 2^-16 product noise of 0 changes that unit's row by 1e-2; allow one
            diff = diff.clone()
            diff[int(diff.norm(dim=1).argmax())] = 0
            rel = float(diff.norm() / max(float(ref.norm()), 1e-3 * total))
        if rel > worst[1]:
            worst = (name, rel)
    report("full model K=16 %s (B=5, T=32): per-frame NLL max rel err %.3e, worst gradient rel L2 %.3e (%s)"
           % (precision, err, worst[1], worst[0]))
    assert err < 1e-4 and worst[1] < 2e-3, worst


@pytest.mark.parametrize("B,T,K", [(64, 40, 4), (32, 33, 2)])
def test_planes_chain_matches_the_fp32_operand_path(gpu_device, monkeypatch, B, T, K):
    """LFI_PCHAIN=0 keeps round 2's data flow (c, dgi and d pre-activation exist as fp32 and every product but cond_transform
    forward splits its operands itself); the default chain has the producing kernels emit bf16 hi / lo operand planes and runs
    gic, dW_c, dpre, the cond_transform weight gradient and the feature gradient on lfi_gemm_planes. Same split, same three
    products per k-step: NLL and every gradient agree to rounding (the K splits of the long products differ)."""
    hp = final_model_hparams(50, 27, K=K)
    batch = to_dev(oracle.synthetic_batch(B, T, 50, 27, seed=B), gpu_device)
    g = torch.Generator().manual_seed(5)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(T - 24, B, cfg["history"], generator=g) < keep).float() / keep
    outs = []
    for chain in ("1", "0"):
        monkeypatch.setenv("LFI_PCHAIN", chain)
        m, _ = perturbed_model(hp, gpu_device)
        m.precision = "bf16x3"
        m.train()
        m.injected_masks = masks
        _, loss, losses = m(batch)
        loss.sum().backward()
        assert bool(m.engine._last.chain) == (chain == "1")
        outs.append((torch.stack(losses), {n: p.grad.clone() for n, p in m.named_parameters()}))
    err = max_rel(outs[0][0], outs[1][0], floor=1.0)
    worst = max(rel_err(outs[0][1][n], outs[1][1][n]) for n in outs[0][1])
    report("planes chain vs fp32-operand path (B=%d, T=%d, K=%d): per-frame NLL max rel diff %.2e, worst gradient diff %.2e" % (B, T, K, err, worst))
    assert err < 2e-6 and worst < 2e-5


def test_generic_and_register_resident_cells_agree(gpu_device, monkeypatch):
    """LFI_FLOW_GENERIC=1 keeps the streaming cell kernels (the path of H > 128 or C > 64): same NLL, same gradients."""
    fxm = Fixture("mid")
    outs = []
    for generic in ("0", "1"):
        monkeypatch.setenv("LFI_FLOW_GENERIC", generic)
        m = build(fxm, gpu_device, train=True)
        m.injected_masks = fxm.masks(torch.float32)
        z_seq, loss, losses = m(to_dev(fxm.batch(), gpu_device))
        loss.sum().backward()
        outs.append((torch.stack(losses), {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert max_rel(outs[0][0], outs[1][0], floor=1.0) < 2e-6
    for n in outs[0][1]:
        assert rel_err(outs[0][1][n], outs[1][1][n]) < 2e-5, n
    assert max_rel(outs[1][0], fxm.get("train/nll"), floor=1.0) < 1e-4


def test_wide_hidden_sizes_take_the_generic_paths(gpu_device):
    """hidden_channels 160 (> 128: streaming flow cells) and a 320-wide GRU window encoder (> 256: unfused recurrence), both
    in the reference's hparam search space (large_hparam_search.py:33,57): NLL and gradients against the fp64 oracle."""
    hp = final_model_hparams(50, 27, K=2)
    hp["Glow"]["hidden_channels"] = 160
    hp["Conditioning"]["p2_face"]["hidden_dim"] = 320
    hp["Conditioning"]["cond_dim"] = 64
    m, sd = perturbed_model(hp, gpu_device)
    m.precision = "f32"
    m.train()
    B, T = 5, 30
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=3)
    N = T - 24
    g = torch.Generator().manual_seed(5)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    m.injected_masks = masks
    _, loss, losses = m(to_dev(batch, gpu_device))
    loss.sum().backward()
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd.items()}
    _, oloss, onll = oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()},
                                            {k: v.double() for k, v in masks.items()})
    oloss.sum().backward()
    assert max_rel(torch.stack(losses), onll.detach(), floor=1.0) < 1e-4
    for name, p in m.named_parameters():
        ref = sdg[name].grad
        rel = float((p.grad.double().cpu() - ref).norm() / max(float(ref.norm()), 1e-12))
        assert rel < 2e-3, (name, rel)


def test_full_size_properties(gpu_device, monkeypatch):
    """BASELINE configs[1] size (B=256, T=80, K=16): size-independent properties instead of an oracle run."""
    hp = final_model_hparams(50, 27)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    batch = to_dev(oracle.synthetic_batch(256, 80, 50, 27, seed=1234), gpu_device)
    with torch.no_grad():
        z_seq, loss, losses = m(batch)
        nll = torch.stack(losses)
        assert torch.isfinite(nll).all() and nll.shape == (56, 256)
        # (1) the flow is a bijection: decode(encode(x)) == x, and the backward NLL mirrors the forward NLL
        rec, bl = m.invert(z_seq, batch)
        x = batch["p1_face"][:, 24:].transpose(0, 1)
        assert rel_err(torch.stack(rec), x) < 2e-4
        logp = (-0.5 * (torch.stack(z_seq) ** 2 + oracle.LOG2PI)).sum(-1)
        assert rel_err(bl + loss, -2.0 * logp.mean().reshape(1) / oracle.LN2) < 1e-4
        # (2) samples are independent: a permuted batch gives the permuted result
        perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).to(gpu_device)
        _, loss_p, losses_p = m({k: v[perm].contiguous() for k, v in batch.items()})
        assert torch.equal(torch.stack(losses_p), nll[:, perm.cpu()])
        # (3) a sub-batch reproduces its rows (no cross-sample coupling, tiles of 16 vs ragged 40)
        # (2 240 windows take the 32-window encoder kernels on the 32 x 32 x 16 MFMA, 14 336 the 64-window ones on 16 x 16 x 32: the same
        # products summed in another order, so this comparison is to fp32 rounding; cross-sample coupling would show as O(1))
        _, _, losses_s = m({k: v[:40].contiguous() for k, v in batch.items()})
        assert max_rel(torch.stack(losses_s), nll[:, :40], floor=1.0) < 1e-5
        # the same with the window encoders pinned to ONE kernel family at both sizes (the 32-window kernels): only then is every sum
        # taken in the same order, and the comparison is bitwise - the detector of cross-sample leakage
        monkeypatch.setenv("LFI_ENC_R64", "0")
        monkeypatch.setenv("LFI_ENC_M16", "0")
        _, _, losses_f = m(batch)
        _, _, losses_s = m({k: v[:40].contiguous() for k, v in batch.items()})
        assert torch.equal(torch.stack(losses_s), torch.stack(losses_f)[:, :40])
        monkeypatch.delenv("LFI_ENC_R64")
        monkeypatch.delenv("LFI_ENC_M16")


def test_config4_deep_flow_properties(gpu_device, monkeypatch):
    """BASELINE configs[4]: K=32 x L=3 (96 flow steps), seq_len 512, batch 128 - too large for the CPU oracle, so
    size-independent properties: finite NLL, decode(encode(x)) = x, forward/backward NLL identity, sub-batch invariance, and
    additivity of the gradient over a split of the batch (the loss is a batch mean: grad(B) = (grad(B1) + grad(B2)) / 2)."""
    hp = final_model_hparams(50, 27, K=32)
    hp["Glow"]["L"] = 3
    hp["Train"]["seq_len"] = 512
    m, _ = perturbed_model(hp, gpu_device)
    assert m.spec.Ks == 96
    m.eval()  # no dropout: the three passes below see the same function
    B, T = 128, 512
    batch = to_dev(oracle.synthetic_batch(B, T, 50, 27, seed=1234), gpu_device)
    with torch.no_grad():
        z_seq, loss, losses = m(batch)
        nll = torch.stack(losses)
        assert nll.shape == (T - 24, B) and torch.isfinite(nll).all()
        short = {k: v[:, :24 + 40].contiguous() for k, v in batch.items()}   # invert walks (n, k) cell by cell: keep it short
        z_short, loss_s, _ = m(short)
        # causal: a prefix reproduces its timesteps - to fp32 rounding: its 5 120 windows take the 32-window encoder kernels
        # (32 x 32 x 16 MFMA), the full sequence's 62 464 the 64-window ones (16 x 16 x 32: the same products, another summation order)
        assert rel_err(torch.stack(z_short), torch.stack(z_seq[:40])) < 1e-5
        # bitwise with the window encoders pinned to one kernel family (no leakage from later timesteps into earlier ones)
        monkeypatch.setenv("LFI_ENC_R64", "0")
        monkeypatch.setenv("LFI_ENC_M16", "0")
        z_full_p, _, _ = m(batch)
        z_short_p, _, _ = m(short)
        assert torch.equal(torch.stack(z_short_p), torch.stack(z_full_p[:40]))
        monkeypatch.delenv("LFI_ENC_R64")
        monkeypatch.delenv("LFI_ENC_M16")
        rec, bl = m.invert(z_short, short)
        # 96 chained fp32 inverses (W^-1 is the fp64 inverse cast to fp32, modules.py:175-177): 7e-4 measured; K=16: 2e-4
        assert rel_err(torch.stack(rec), short["p1_face"][:, 24:].transpose(0, 1)) < 3e-3
        logp = (-0.5 * (torch.stack(z_short) ** 2 + oracle.LOG2PI)).sum(-1)
        assert rel_err(bl + loss_s, -2.0 * logp.mean().reshape(1) / oracle.LN2) < 1e-4
        _, _, losses_s = m({k: v[:24].contiguous() for k, v in batch.items()})
        assert max_rel(torch.stack(losses_s), nll[:, :24], floor=1.0) < 1e-5     # (same kernels as the full batch here: 11 712 windows)

    def grads(b):
        m.zero_grad(set_to_none=True)
        _, l, _ = m(b)
        l.sum().backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    g_all = grads(batch)
    g_a = grads({k: v[:64].contiguous() for k, v in batch.items()})
    g_b = grads({k: v[64:].contiguous() for k, v in batch.items()})
    worst = 0.0
    for n, g in g_all.items():
        assert torch.isfinite(g).all(), n
        ref = 0.5 * (g_a[n] + g_b[n])
        worst = max(worst, float((g - ref).norm() / max(float(ref.norm()), 1e-20)))
    report("config[4] 96 flow steps x 488 timesteps x batch 128: gradient additivity over a batch split, worst rel L2 %.3e" % worst)
    assert worst < 1e-4


def test_config3_sampling_full_size(gpu_device, monkeypatch):
    """BASELINE configs[3]: autoregressive sampling, batch 1024, seq_len 300. Properties: finite; the hipGraph replay (second
    call of a shape) is bit-identical to the eager first call; a sub-batch reproduces its rows; and the teacher-forced
    forward pass maps the generated frames back to the injected prior noise (encode(decode(z)) = z)."""
    hp = final_model_hparams(50, 27)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    B, T = 1024, 300
    g = torch.Generator().manual_seed(11)
    data = {"p1_face": torch.zeros(B, T, 50)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, T, d, generator=g)
    data = to_dev(data, gpu_device)
    noise = (torch.randn(T - 24, B, 50, generator=g) * 0.8).to(gpu_device)
    out1 = m.inference(T, data, noise=noise)   # eager launches
    out2 = m.inference(T, data, noise=noise)   # captured + replayed
    out3 = m.inference(T, data, noise=noise)   # replayed
    assert out1.shape == (B, T - 24, 50) and torch.isfinite(out1).all()
    assert torch.equal(out1, out2) and torch.equal(out1, out3)
    # round 5: the static part of runs 2.. executes on a stream that owns half of every XCD BESIDE the chain (the default on this card);
    # on an ordinary second stream (LFI_SAMPLE_STATIC_CUS=0) or with another share of the chip the same kernels give the same bits
    eng = m._ensure_engine(gpu_device)
    if "LFI_SAMPLE_STATIC_CUS" not in os.environ:
        assert eng._sample_static_cus(T - 24) == 16 and len(eng._partial_streams) == 1
    for cus in ("0", "8"):
        monkeypatch.setenv("LFI_SAMPLE_STATIC_CUS", cus)
        assert torch.equal(m.inference(T, data, noise=noise), out1), "static part on %s CUs per XCD" % cus
    monkeypatch.undo()
    # a sub-batch reproduces its rows: to what 276 autoregressive frames make of an fp32 rounding difference in the window encoders
    # (the sampler encodes its windows in frame runs of 46 - 69 frames: 48 x 69 = 3 312 windows a run take the 32-window kernels on the 32 x 32 x 16
    # MFMA, the full batch's 70 656 the 64-window ones on 16 x 16 x 32 - the same products, another summation order)
    sub = m.inference(T, {k: v[:48].contiguous() for k, v in data.items()}, noise=noise[:, :48].contiguous())
    sub_err = float((sub - out1[:48]).abs().max() / out1[:48].abs().max().clamp(min=1.0))
    report("config[3] sub-batch of 48 against its rows of the batch of 1024: max abs difference %.3e of the largest value" % sub_err)
    assert sub_err < 1e-3
    # bitwise with the window encoders pinned to one kernel family at both batch sizes (the leakage detector)
    monkeypatch.setenv("LFI_ENC_R64", "0")
    monkeypatch.setenv("LFI_ENC_M16", "0")
    out_p = m.inference(T, data, noise=noise)
    sub_p = m.inference(T, {k: v[:48].contiguous() for k, v in data.items()}, noise=noise[:, :48].contiguous())
    assert torch.equal(sub_p, out_p[:48])
    monkeypatch.delenv("LFI_ENC_R64")
    monkeypatch.delenv("LFI_ENC_M16")
    full = dict(data)
    full["p1_face"] = torch.cat([data["p1_face"][:, :24], out1], dim=1).contiguous()
    with torch.no_grad():
        z_seq, _, _ = m(full)
    err = float((torch.stack(z_seq) - noise).abs().max())
    report("config[3] batch 1024 x 276 generated frames: |encode(decode(z)) - z| max %.3e" % err)
    assert err < 2e-3


@pytest.mark.parametrize("case", ["mid", "tiny_lstm", "final_ragged", "final_full"])
def test_pipeline_walk_matches_diagonal_walk(gpu_device, monkeypatch, case):
    """The persistent systolic walk (one launch, workgroup (k, tile) keeps flow step k's weights and walks the timesteps,
    tiles handed from step to step through sc1 stores + progress counters) does the same arithmetic in the same order as
    one launch per anti-diagonal (LFI_FLOW_PIPE=0): bit-identical NLL, z and gradients."""
    if case in ("mid", "tiny_lstm"):
        fxm = Fixture(case)
        mk = lambda: build(fxm, gpu_device, train=True)  # noqa: E731
        batch = to_dev(fxm.batch(), gpu_device)
        masks = fxm.masks(torch.float32)
    else:
        hp = final_model_hparams(50, 27)
        B, T = (40, 40) if case == "final_ragged" else (256, 80)
        mk = lambda: perturbed_model(hp, gpu_device)[0].train()  # noqa: E731  (engine default: bf16x3)
        batch = to_dev(oracle.synthetic_batch(B, T, 50, 27, seed=5), gpu_device)
        g = torch.Generator().manual_seed(5)
        masks = {}
        for name in ("p2_face", "p1_speech", "p2_speech"):
            cfg = hp["Conditioning"][name]
            keep = 1.0 - cfg["dropout"]
            masks[name] = (torch.rand(T - 24, B, cfg["history"], generator=g) < keep).float() / keep
    outs = []
    monkeypatch.setenv("LFI_PIPE_X3", "0")   # same arithmetic in both walks: the recurrent products on the exact f32 MFMA
    # The bf16x3 form of the walk's recurrent products exists for GRU cells with hidden_channels and C // 2 padded to multiples of
    # 32 only (lfi_flow.hip: H16 % 32 == 0 && Ch16 % 32 == 0 && !lstm): the fixture cases - "mid" H = 44, "tiny_lstm" an LSTM - have
    # no such kernel in ANY engine mode (VERDICT r3 weak #1c: their third leg printed 0.00e+00, vacuously; measured again in round
    # 4 with the leg moved to the bf16x3 engine mode: still bit-identical). They run the two bitwise legs; the bf16x3-recurrence leg
    # runs where the kernel exists, at final widths.
    fixture_case = case in ("mid", "tiny_lstm")
    for pipe in ("1", "0") + (() if fixture_case else ("x3",)):
        monkeypatch.setenv("LFI_FLOW_PIPE", "0" if pipe == "0" else "1")
        monkeypatch.setenv("LFI_PIPE_X3", "1" if pipe == "x3" else "0")
        m = mk()
        m.injected_masks = masks
        z_seq, loss, losses = m(batch)
        loss.sum().backward()
        outs.append((torch.stack(losses), torch.stack(z_seq), {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for n in outs[0][2]:
        if case == "tiny_lstm":  # the d c carry lives in a register there and its product is contracted into the consumer's fma
            assert rel_err(outs[0][2][n], outs[1][2][n]) < 1e-6, n
        elif n.endswith(("f.rnn.bias_ih", "f.rnn.bias_hh")):
            # the persistent walk sums dgi / dgh per workgroup while it walks (timesteps in walk order, then tiles); the
            # diagonal walk leaves them to a column-sum pass over the stash: same addends, another order
            assert rel_err(outs[0][2][n], outs[1][2][n]) < 2e-6, n
        else:
            assert torch.equal(outs[0][2][n], outs[1][2][n]), n
    # default of the bf16x3 engine mode: the recurrent products of the walk as three bf16 MFMAs (2^-16 relative per product)
    if fixture_case:
        return
    ref = outs[1]
    err = max_rel(outs[2][0], ref[0], floor=1.0)
    report("%s: persistent walk with bf16x3 recurrent products vs exact f32 cells: per-frame NLL max rel diff %.2e" % (case, err))
    assert err < 2e-5
    assert err > 0.0, "the bf16x3-recurrence leg ran the same arithmetic as its reference: the switch selected nothing"
    for n in outs[2][2]:
        assert rel_err(outs[2][2][n], ref[2][n]) < 2e-4, n


def test_abandoned_walk_is_loud(gpu_device, monkeypatch):
    """A persistent walk whose bounded spin times out sets the abort word; every workgroup then leaves its loop and the
    results are poisoned on the device (NaN NLL, NaN gradients) instead of passing for numbers. LFI_PIPE_FORCE_ABORT=1
    starts the walk with the word already set."""
    fxm = Fixture("mid")
    m = build(fxm, gpu_device, train=True)
    m.injected_masks = fxm.masks(torch.float32)
    batch = to_dev(fxm.batch(), gpu_device)
    monkeypatch.setenv("LFI_PIPE_FORCE_ABORT", "1")
    _, loss, losses = m(batch)
    assert torch.isnan(torch.stack(losses)).all() and torch.isnan(loss).all()
    m.engine.backward(1.0)
    torch.cuda.synchronize()
    for name in ("w_hh", "w_ih", "w_fl", "wct"):
        assert torch.isnan(m.engine.fview(name, m.engine.grads)).any(), name
    assert torch.isnan(m.engine.view("enc.p2_face.weight_hh", m.engine.grads)).any()
    monkeypatch.delenv("LFI_PIPE_FORCE_ABORT")
    _, loss, losses = m(batch)          # the next call is clean again: the state block is re-zeroed by every walk
    assert max_rel(torch.stack(losses), fxm.get("train/nll"), floor=1.0) < 1e-4


def test_abandoned_walk_is_loud_with_bf16_gradient_rows(gpu_device, monkeypatch):
    """with the backward walk's dgi | dgh rows stored as bf16 (two-product mode on the planes chain, the benchmark's
    arithmetic) the poison kernel's fp32 NaN stores landed on other rows or outside the bf16 region, so w_hh of even flow steps and
    half of the w_ih targets stayed finite after an abandoned walk. Every flow step's w_ih (both column blocks), w_hh, w_fl and the
    LU parameters must come out NaN - through the one-pass thin products and through the four split-K products alike."""
    hp = final_model_hparams(50, 27, K=3)
    B, T = 32, 24 + 8
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=5)
    for fused in ("1", "0"):
        monkeypatch.setenv("LFI_FLOW_WGRAD_FUSED", fused)
        m, _ = perturbed_model(hp, gpu_device)
        m.precision = "bf16x3"
        m.eval()
        eng = m._ensure_engine(gpu_device)
        eng.backward_products = 2
        _, loss, _ = m(to_dev(batch, gpu_device))      # a clean forward: only the backward walk is abandoned
        monkeypatch.setenv("LFI_PIPE_FORCE_ABORT", "1")
        eng.backward(1.0)
        torch.cuda.synchronize()
        monkeypatch.delenv("LFI_PIPE_FORCE_ABORT")
        assert bool(eng._last.chain)
        Ks, s = eng.spec.Ks, eng.spec
        for name in ("w_hh", "w_fl", "inv_l"):
            g = eng.fview(name, eng.grads).reshape(Ks, -1)
            assert torch.isnan(g).any(dim=1).all(), (fused, name)
        w_ih = eng.fview("w_ih", eng.grads).reshape(Ks, s.G, s.I)
        assert torch.isnan(w_ih[:, :, :s.Ch]).reshape(Ks, -1).any(dim=1).all(), fused
        assert torch.isnan(w_ih[:, :, s.Ch:]).reshape(Ks, -1).any(dim=1).all(), fused
        del m, eng


def test_two_bucket_allreduce_protocol(gpu_device):
    """fused_training_step under data parallelism hands the gradient to the all-reduce callable in two buckets (flow block
    asynchronously before the encoder BPTT, encoder block after): every float exactly once, and with a stand-in that behaves
    like two identical ranks (sum = 2 g, then the optimiser's 1/world) the step is bit-identical to the single-rank step."""
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fxm = Fixture("mid")

    import random

    def make():
        torch.manual_seed(0)
        # the negative-example branch is on in this fixture and is drawn from PYTHON's generator (lets_face_it_glow.py:40, 10 % of the
        # steps): unseeded, one of the two models took it in 18 % of the runs and the bitwise comparison below failed (round 5:
        # tools/step_determinism_probe.py showed every gradient x -0.1 in those runs - the branch, not the engine)
        random.seed(0)
        m = LetsFaceItGlow(Namespace(**fxm.hp))
        m.seq_glow.load_state_dict(fxm.state_dict(torch.float32))
        m.to(gpu_device).train()
        m.seq_glow.glow.set_actnorm_init(True)
        m.seq_glow.injected_masks = fxm.masks(torch.float32)
        return m

    batch = to_dev(fxm.batch(), gpu_device)
    m1 = make()
    m1.fused_training_step(batch, 1e-3, 1, None)
    calls = []

    class Work:
        waited = False

        def wait(self):
            Work.waited = True

    def fake_allreduce(t, async_op=False):
        calls.append((t.data_ptr(), t.numel(), async_op))
        t.mul_(2.0)          # two identical ranks
        return Work() if async_op else None

    m2 = make()
    m2.fused_training_step(batch, 1e-3, 2, fake_allreduce)
    eng = m2.seq_glow.engine
    base = eng.grads.data_ptr()
    assert len(calls) == 2 and calls[0][2] is True and calls[1][2] is False and Work.waited
    assert calls[0][0] == base + 4 * eng.flow_offset and calls[0][1] == eng.n_params - eng.flow_offset
    assert calls[1][0] == base and calls[1][1] == eng.flow_offset and eng.flow_offset > 0
    assert torch.equal(m1.seq_glow.engine.params, eng.params)


@pytest.mark.parametrize("fname", ["final_model.yaml", "final_model_synthetic.yaml", "no_speech.yaml", "no_nll_trick.yaml"])
def test_every_shipped_hparams_file_trains(gpu_device, fname):
    """The runnable hparams files of the reference (hparams/*.yaml: the final model, no_speech, no_nll_trick; its no_face.yaml sets
    p1_face.dim = 0, a 0-channel flow that the reference cannot construct either) plus BASELINE's synthetic
    dims, through LetsFaceItGlow.fused_training_step on random data: ActNorm data-dependent init on the first step, finite
    loss, parameters move, and the loss of a fixed batch goes down over a few steps."""
    import os
    import random
    import numpy as np
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hp = load_hparams_file(os.path.join(root, "lets_face_it_amd", "hparams", fname))
    hp["batch_size"] = 32
    hp["gradient_clip_val"] = 20
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    m = LetsFaceItGlow(Namespace(**hp)).to(gpu_device).train()
    spec = m.seq_glow.spec
    T = spec.start + 12
    g = torch.Generator().manual_seed(1)
    batch = {"p1_face": torch.randn(32, T, spec.C, generator=g)}
    for e in spec.encoders:
        if e.name != "p1_face":
            batch[e.name] = torch.randn(32, T, e.in_dim, generator=g)
    batch = to_dev(batch, gpu_device)
    before = m.seq_glow._ensure_engine(gpu_device).params.clone()
    losses = [float(m.fused_training_step(batch, 1e-3)) for _ in range(6)]
    assert all(np.isfinite(losses)), losses
    assert m.seq_glow.glow.actnorm_inited()
    assert not torch.equal(before, m.seq_glow.engine.params)
    assert losses[-1] < losses[1], losses     # (step 0 includes the ActNorm init; the negative-example branch needs mm_nll > 0)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_final_width_sampling_against_oracle(gpu_device, precision):
    """SeqGlow.inference at final_model widths (H = 128, D = 512, 256-wide GRU windows; K = 4 so that the CPU oracle takes
    seconds) with injected prior noise against the fp64 oracle: in bf16x3 mode this runs the per-frame reverse chain with
    the recurrent products as three bf16 MFMAs."""
    hp = final_model_hparams(50, 27, K=4)
    m, sd = perturbed_model(hp, gpu_device)
    m.precision = precision
    m.eval()
    B, seq_len = 6, 24 + 14
    g = torch.Generator().manual_seed(3)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(seq_len - 24, B, 50, generator=g) * 0.8
    out = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))
    out2 = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))   # graph replay
    ref = oracle.seqglow_inference(hp, {k: v.double() for k, v in sd.items()}, seq_len, {k: v.double() for k, v in data.items()},
                                   noise.double())
    err = float((out.cpu().double() - ref).abs().max())
    ref32 = oracle.seqglow_inference(hp, sd, seq_len, data, noise)    # the same op sequence in plain fp32 (CPU)
    own = float((ref32.double() - ref).abs().max())
    report("final-width sampling (%s): max abs err vs fp64 oracle %.2e (plain fp32 torch on the CPU: %.2e)" % (precision, err, own))
    assert err < 1e-5 and torch.equal(out, out2)


def test_real_allocation_failure_is_worded_for_the_optuna_harness(gpu_device):
    """SURVEY.md par. 8(f) row 4: the reference's Optuna harness halves the batch when a trial dies with a RuntimeError whose
    text starts with "CUDA out of memory" (hparams_tuning.py:162-168,200-205). A GENUINE allocator failure inside
    GlowEngine.forward — the caching allocator capped at 8 GiB above what is live, a batch whose workspaces need ~40 GB — must
    surface in that wording (PyTorch-ROCm says "HIP out of memory"), and the engine must keep working afterwards."""
    hp = final_model_hparams(50, 27)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    small = to_dev(oracle.synthetic_batch(8, 40, 50, 27, seed=1), gpu_device)
    with torch.no_grad():
        before = torch.stack(m(small)[2])
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    big = {k: torch.zeros(1024, 80, v.shape[2], device=gpu_device) for k, v in small.items()}
    total = torch.cuda.get_device_properties(gpu_device).total_memory
    cap = torch.cuda.memory_allocated(gpu_device) + 8 * 2 ** 30     # what is live now + 8 GiB; the forward needs ~40 GiB
    torch.cuda.set_per_process_memory_fraction(min(1.0, cap / total), gpu_device)
    try:
        with pytest.raises(RuntimeError) as ei:
            with torch.no_grad():
                m(big)
        assert isinstance(ei.value, torch.OutOfMemoryError)
        assert str(ei.value).startswith("CUDA out of memory"), str(ei.value)[:200]
        assert "HIP out of memory" in str(ei.value)      # the allocator's own message is kept inside
    finally:
        torch.cuda.set_per_process_memory_fraction(1.0, gpu_device)
        del big
        torch.cuda.empty_cache()
    with torch.no_grad():
        after = torch.stack(m(small)[2])
    assert torch.equal(before, after)


def test_checkpoint_resume_keeps_the_optimiser_state(gpu_device, tmp_path):
    """a checkpoint holds Adam's moments and step count next to the weights, and a resumed run continues
    bit-identically to an uninterrupted one (with betas[1] = 0.9999 restarted moments would change every later step). Also:
    a no-op `.to(device)` — what a second Trainer.fit() does — must not drop the engine or its optimiser state."""
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.trainer import Trainer
    fxm = Fixture("mid")

    def make():
        hp = Namespace(**Fixture("mid").hp)
        hp.Train["use_negative_nll_loss"] = False
        hp.gradient_clip_val = 20
        hp.checkpoint_dir = str(tmp_path)
        m = LetsFaceItGlow(hp)
        m.seq_glow.load_state_dict(fxm.state_dict(torch.float32))
        m.to(gpu_device).train()
        m.seq_glow.glow.set_actnorm_init(True)
        m.seq_glow.injected_masks = fxm.masks(torch.float32)
        return hp, m

    batch = to_dev(fxm.batch(), gpu_device)
    hp, a = make()
    for _ in range(4):
        a.fused_training_step(batch, 1e-3)
    hp, b = make()
    tr = Trainer(hp, device=gpu_device)
    for _ in range(2):
        b.fused_training_step(batch, 1e-3)
    eng = b.seq_glow.engine
    b.to(gpu_device)                                    # no-op move: same engine, same Adam state
    assert b.seq_glow.engine is eng and eng.step_count == 2
    tr.global_step, tr.epoch = 2, 1
    path = str(tmp_path / "last.ckpt")
    tr.save_checkpoint(b, path)
    hp2, c = make()
    tr2 = Trainer(hp2, device=gpu_device)
    tr2.resume(c, path)
    assert tr2.global_step == 2 and tr2.epoch == 1 and c.seq_glow.engine.step_count == 2
    c.seq_glow.injected_masks = fxm.masks(torch.float32)
    c.train()
    for _ in range(2):
        c.fused_training_step(batch, 1e-3)
    assert torch.equal(a.seq_glow.engine.params, c.seq_glow.engine.params)
    assert torch.equal(a.seq_glow.engine.adam_v, c.seq_glow.engine.adam_v)
    # a dtype round trip re-creates parameter storage: the engine is rebuilt, the optimiser state travels with it
    c.double().float()
    c.seq_glow._ensure_engine(gpu_device)
    assert c.seq_glow.engine.step_count == 4 and torch.equal(a.seq_glow.engine.adam_m, c.seq_glow.engine.adam_m)


@pytest.mark.parametrize("B,T,K", [(64, 24 + 8, 3), (32, 24 + 1, 2), (32, 24 + 40, 2)])
def test_one_pass_thin_weight_gradients_match_the_four_products(gpu_device, monkeypatch, B, T, K):
    """Round 6: the flow's thin weight-gradient products (w_hh, w_ih[:, :Ch], w_fl, b_fl and the invconv's dW: autograd of
    glow/models.py:204-214, glow/modules.py:93-95,147-177) in ONE pass over the backward stash (lfi_wgrad.hip) against the four
    batched split-K products + column-sum pass they replace (LFI_FLOW_WGRAD_FUSED=0), final widths, two-product backward (the
    mode the one-pass form exists for: dgi | dgh arrive as bf16 rows). Same operands, same roundings, another summation order:
    every flow gradient agrees to fp32 rounding; every other gradient bit for bit (they do not pass through the changed code).
    Cases (the planes chain needs F % 32 == 0): several batch tiles and timestep ranges (the w_hh role: 3 flow steps x 4 tiles x 2 ranges), a single timestep (w_hh = 0), a longer walk."""
    hp = final_model_hparams(50, 27, K=K)
    N = T - 24
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=77)
    got = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("LFI_FLOW_WGRAD_FUSED", fused)
        m, _ = perturbed_model(hp, gpu_device)
        m.precision = "bf16x3"
        m.eval()
        eng = m._ensure_engine(gpu_device)
        eng.backward_products = 2
        _, loss, _ = m(to_dev(batch, gpu_device))
        loss.sum().backward()
        torch.cuda.synchronize()
        assert bool(eng._last.chain) and eng.backward_product_count(B * N) == 2
        got[fused] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        del m, eng
    touched = ("f.rnn.weight_hh", "f.rnn.weight_ih", "f.final_linear.weight", "f.final_linear.bias", "invconv.")
    worst = ("", 0.0)
    for name, ref in got["0"].items():
        g = got["1"][name]
        if any(t in name for t in touched):
            e = rel_err(g, ref)
            if e > worst[1]:
                worst = (name, e)
            assert e < 2e-5, (name, e)
        else:
            assert torch.equal(g, ref), name
    report("one-pass thin weight gradients vs the four split-K products (B=%d, N=%d, K=%d): worst tensor rel L2 %.2e (%s)"
           % (B, N, K, worst[1], worst[0]))


def test_first_forward_of_a_new_engine_waits_for_its_buffers_births(gpu_device, monkeypatch):
    """Round 6: an engine's workspaces are born zeroed on first use, on the stream that is current then; work forked onto the second
    stream BEFORE that (the smallest window encoder, the prev_p1_face gather) was not ordered behind the fill, and when the fill ran late
    it wiped the feature columns they had written - in the FIRST forward pass of a new engine only, which is the pass ActNorm's
    data-dependent init runs in (seen as run-to-run different parameters on two ranks sharing a card; tools/dp_gloo_check.py,
    profiles/round6_first_forward_race.md). Two checks. Structural: every buffer born while the second stream is forked makes that
    stream wait for the creating stream (no timing involved). Behavioural: with the main stream kept busy for milliseconds right after
    every fork - the fill is then late whenever the two streams run concurrently - the first forward pass of a fresh engine gives the
    NLL of an undisturbed one bit for bit. Whether the old behaviour (LFI_NO_BIRTH_ORDER=1) reproduces the wipe under that load depends on
    the two streams sitting on different hardware queues (alone in a process it does; late in a long test process it did not): reported,
    not asserted."""
    fxm = Fixture("mid")
    batch = to_dev(fxm.batch(), gpu_device)
    big = torch.randn(6144, 6144, device=gpu_device)

    def first_forward(delay, ordered, log=None):
        if ordered:
            monkeypatch.delenv("LFI_NO_BIRTH_ORDER", raising=False)
        else:
            monkeypatch.setenv("LFI_NO_BIRTH_ORDER", "1")
        m = build(fxm, gpu_device, train=False)
        eng = m._ensure_engine(gpu_device)
        if delay:
            fork = eng._fork

            def slow_fork():
                side = fork()
                for _ in range(4):
                    torch.mm(big, big)      # the main stream falls behind the second one
                return side
            eng._fork = slow_fork
        if log is not None:
            buf, birth = eng._buf, eng._order_birth

            def logged_buf(name, floats, zero=False):
                new = name not in eng._ws
                t = buf(name, floats, zero)
                if new:
                    log.append(("born", name, eng._side_stream is not None))
                return t

            def logged_birth():
                log.append(("ordered",))
                return birth()
            eng._buf, eng._order_birth = logged_buf, logged_birth
        with torch.no_grad():
            _, _, losses = m(batch)
        torch.cuda.synchronize()
        return torch.stack(losses)

    log = []
    ref = first_forward(False, True, log)
    assert max_rel(ref, fxm.get("eval/nll"), floor=1.0) < 1e-4
    births = [i for i, e in enumerate(log) if e[0] == "born"]
    assert any(log[i][1] == "cond" and log[i][2] for i in births), "the feature matrix was not born after the fork: the scenario is gone"
    # every birth is preceded (inside _buf) by its ordering call
    assert all(i > 0 and log[i - 1] == ("ordered",) for i in births), log[:12]
    assert torch.equal(first_forward(True, True), ref), "the first forward pass of a new engine depends on stream timing"
    old = first_forward(True, False)
    report("first forward pass of a new engine with the buffers' zero fill forced late: ordered births bit-identical to an undisturbed pass; "
           "the old behaviour (no ordering) under the same load: %s" % ("reproduces the wipe (NLL differs)" if not torch.equal(old, ref)
                                                                          else "did not lose the race in this process"))
