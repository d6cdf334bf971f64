"""Oracle parity at the DEPTH of BASELINE configs[4] (missing #3 / weak #1b): K = 32 x L = 3 = 96 flow steps at
final widths - the depth at which the persistent walk runs 96 x tiles workgroups as successive groups - for the per-frame NLL,
every parameter gradient, autoregressive sampling and `invert`, against the fp64 CPU oracle (reference loops:
glow/models.py:413-434 FlowNet, :534-561 forward, :567-596 inference, :617-645 invert), in both GEMM modes.

The oracle walks 96 flow steps x 8 timesteps in seconds at these batch sizes; configs[4]'s own size (T = 512, B = 128) stays
with the size-independent properties of tests/test_gpu_parity.py::test_config4_deep_flow_properties.
"""
import pytest
import torch

from helpers import max_rel, rel_err, report
from oracle import seqglow_oracle as oracle
from test_gpu_parity import final_model_hparams, perturbed_model, to_dev

pytestmark = pytest.mark.gpu


def deep_hparams():
    hp = final_model_hparams(50, 27, K=32)
    hp["Glow"]["L"] = 3
    return hp


def _masks(hp, N, B, seed):
    g = torch.Generator().manual_seed(seed)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    return masks


@pytest.mark.parametrize("B", [6, 64])
def test_deep_flow_against_oracle(gpu_device, B):
    """96 flow steps, T = 24 + 4, injected dropout masks, both engine modes: per-frame NLL (1e-4 relative) and EVERY parameter gradient (2e-3
    relative L2) against the fp64 oracle. B = 6 is one ragged 16-sample tile per flow step (96 workgroups); B = 64 is four
    tiles per step = 384 (k, tile) workgroups, more than the chip has CUs: the walk then runs them as successive groups, and it
    is the ORACLE that pins that path here, not the walk compared with itself."""
    hp = deep_hparams()
    T = 24 + 4
    N = T - 24
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=40 + B)
    masks = _masks(hp, N, B, 8)
    got = {}
    for precision in ("f32", "bf16x3"):
        m, sd = perturbed_model(hp, gpu_device)
        assert m.spec.Ks == 96
        m.precision = precision
        m.train()
        m.injected_masks = masks
        z_seq, loss, losses = m(to_dev(batch, gpu_device))
        loss.sum().backward()
        got[precision] = (torch.stack(losses).cpu(), torch.stack(z_seq).detach().cpu(),
                          {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()})
        del m
    # (ONE oracle pass for both engine modes: 96 flow steps x 4 timesteps in fp64 autograd is the slow part of this test; its ops are
    # small, a modest thread team is faster than all of the box's cores)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd.items()}
    z64, oloss, onll = oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()},
                                              {k: v.double() for k, v in masks.items()})
    oloss.sum().backward()
    torch.set_num_threads(threads)
    total = float(torch.sqrt(sum((v.grad ** 2).sum() for v in sdg.values() if v.grad is not None)))
    for precision, (nll, z, grads) in got.items():
        err = max_rel(nll, onll.detach(), floor=1.0)
        zerr = rel_err(z, z64.detach())
        worst = ("", 0.0)
        for name, g in grads.items():
            ref = sdg[name].grad
            diff = g - ref
            rel = float(diff.norm() / max(float(ref.norm()), 1e-3 * total))
            if rel > 1e-3 and precision == "bf16x3" and "cond_transform.0.weight" in name:
                # LeakyReLU's kink (test_full_model_k16_gradients_against_oracle): with a few hundred frames behind each weight row,
                # ONE (frame, unit) pre-activation within the 2^-16 product noise of 0 changes that unit's row by 1e-2; allow one
                diff = diff.clone()
                diff[int(diff.norm(dim=1).argmax())] = 0
                rel = float(diff.norm() / max(float(ref.norm()), 1e-3 * total))
            if rel > worst[1]:
                worst = (name, rel)
        report("DEEP flow K=32 x L=3 (96 steps) %s, B=%d, T=%d vs fp64 oracle: per-frame NLL max rel err %.3e, z rel err %.3e, worst "
               "gradient rel L2 %.3e (%s)" % (precision, B, T, err, zerr, worst[1], worst[0]))
        assert err < 1e-4 and zerr < 1e-4, precision
        assert worst[1] < 2e-3, (precision, worst)


def test_deep_flow_sampling_and_invert_against_oracle(gpu_device):
    """96 reverse flow steps per generated frame: `inference` over 4 frames with injected prior noise and `invert` of the
    teacher-forced latents, against the fp64 oracle. Sampling gate as at K = 16 (test_k16_sampling_against_oracle): north_star's
    1e-5 absolute or 1.5 x what plain fp32 torch does on the same inputs, whichever is larger - both are reported."""
    hp = deep_hparams()
    m, sd = perturbed_model(hp, gpu_device)
    m.eval()
    B, seq_len = 4, 24 + 4
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))      # (the oracle's ops are small)
    g = torch.Generator().manual_seed(13)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(seq_len - 24, B, 50, generator=g) * 0.8
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = oracle.seqglow_inference(hp, sd64, seq_len, {k: v.double() for k, v in data.items()}, noise.double())
    ref32 = oracle.seqglow_inference(hp, sd, seq_len, data, noise)
    # 96 inverse couplings (z2 / scale - shift with scale = sigmoid(.) < 1) of a random-init flow blow the frames up frame over frame
    # (|x| reaches 1e9 and more by the eighth): the error is measured per generated frame relative to that frame's largest magnitude
    # (never below 1: for |x| <= 1 it is north_star's absolute 1e-5)
    # and that growth is chaotic: by the eighth frame plain fp32 torch has lost every digit against fp64 (relative error ~1). Each
    # frame is therefore gated against what fp32 itself keeps of it: max(1e-5, 3 x the fp32 error of THAT frame); the first frames,
    # where fp32 still means something, are the test
    scale = ref.abs().amax(dim=(0, 2)).clamp(min=1.0)                               # (frames,)
    own = (ref32.double() - ref).abs().amax(dim=(0, 2)) / scale
    seen = {}
    for precision in ("f32", "bf16x3"):
        m.precision = precision
        out = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))
        out2 = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))   # hipGraph replay
        per_frame = (out.cpu().double() - ref).abs().amax(dim=(0, 2)) / scale
        report("DEEP flow (96 steps) sampling %s, batch 4 x %d generated frames (|x| up to %.1e): per-frame max error relative to the "
               "frame's largest value vs fp64 oracle %s; plain fp32 torch on the CPU %s"
               % (precision, seq_len - 24, float(ref.abs().max()), " ".join("%.1e" % v for v in per_frame.tolist()),
                  " ".join("%.1e" % v for v in own.tolist())))
        assert torch.equal(out, out2)
        assert torch.isfinite(out).all()
        # exact-f32 engine mode: the same arithmetic class as plain fp32 torch, every frame within 3 x of what that keeps. bf16x3
        # mode: its static part (window encoders, the non-autoregressive cond_transform columns) carries 2^-16, which these
        # dynamics amplify by ~50 x per frame like any other perturbation - bounded by 10 x the f32 mode's own error per frame
        seen[precision] = per_frame
        for fi in range(per_frame.numel()):
            bound = max(1e-5, 3.0 * float(own[fi]))
            if precision == "bf16x3":
                bound = max(bound, 10.0 * float(seen["f32"][fi]))
            assert float(per_frame[fi]) <= bound, (precision, fi, float(per_frame[fi]), float(own[fi]))

    # invert: teacher-forced reverse pass of the oracle's own latents on a fresh batch
    batch = oracle.synthetic_batch(B, seq_len, 50, 27, seed=71)
    b64 = {k: v.double() for k, v in batch.items()}
    z64, floss, _ = oracle.seqglow_forward(hp, sd64, b64)
    rec64, bl64 = oracle.seqglow_invert(hp, sd64, z64, b64)
    rec32, _ = oracle.seqglow_invert(hp, sd, z64.float(), batch)
    own_inv = float((rec32.double() - rec64).abs().max())
    torch.set_num_threads(threads)
    for precision in ("f32", "bf16x3"):
        m.precision = precision
        rec, bl = m.invert(list(z64.float().to(gpu_device).unbind(0)), to_dev(batch, gpu_device))
        ierr = float((torch.stack(rec).cpu().double() - rec64).abs().max())
        lerr = rel_err(bl, bl64)
        report("DEEP flow (96 steps) invert %s, batch 4 x %d frames: reconstruction max abs err vs fp64 oracle %.2e (plain fp32 torch: "
               "%.2e), backward loss rel err %.2e" % (precision, seq_len - 24, ierr, own_inv, lerr))
        assert ierr <= max(1e-4, 3.0 * own_inv), (precision, ierr, own_inv)
        assert lerr < 1e-4, precision
