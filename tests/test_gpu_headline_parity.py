"""Parity AT THE BENCHMARKED CONFIGURATION: the HIP path against the fp64 CPU oracle where bench.py
actually runs — B = 256, T = 80, K = 16, C = 50 / S = 27, bf16x3 GEMM mode — for the per-frame NLL and EVERY parameter
gradient, and K = 16 autoregressive sampling over 56 generated frames.

Only at F = 14 336 frames do the 85-way split-K of the 768 x 256 x 330 k dW_hh product, the 3 584-workgroup planes GEMM, the
per-tile column-sum epilogue at 112 row tiles and the XCD tile order exist at all; the small-batch oracle tests never see them.
The oracle's fp64 forward + autograd backward at this size is ~1-2 minutes of host time (reference loop:
glow/models.py:534-561; sampling loop :567-596).
"""
import time
from argparse import Namespace

import pytest
import torch

from helpers import max_rel, report
from oracle import seqglow_oracle as oracle
from test_gpu_parity import final_model_hparams, perturbed_model, to_dev

pytestmark = pytest.mark.gpu


def _masks(hp, N, B, seed):
    g = torch.Generator().manual_seed(seed)
    masks = {}
    for name in ("p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][name]
        keep = 1.0 - cfg["dropout"]
        masks[name] = (torch.rand(N, B, cfg["history"], generator=g) < keep).float() / keep
    return masks


def test_headline_config_gradients_against_oracle(gpu_device, monkeypatch):
    """BASELINE configs[1] itself: B=256, T=80, K=16, injected dropout masks, bf16x3. Per-frame NLL (gate 1e-4 relative,
    north_star) and every parameter gradient (relative L2 against the fp64 oracle, gate 2e-3 as everywhere else; no
    LeakyReLU-kink allowance is needed with 14 336 frames behind each weight row) - in the engine's default arithmetic at this
    size (engine_backward_products "auto": two bf16 products in the backward GEMM classes from 8192 frames up) and with three
    products everywhere. The default arithmetic also keeps the window encoders' gate stash as fp16 (round 4): the same leg with
    that stash in fp32 (LFI_ENC_STASH_F16=0) is reported beside it, so the stash's share of the error is visible."""
    hp = final_model_hparams(50, 27, K=16)
    B, T = 256, 80
    N = T - 24
    batch = oracle.synthetic_batch(B, T, 50, 27, seed=1234)
    masks = _masks(hp, N, B, 6)
    got = {}
    for products in ("auto", "auto, fp32 gate stash", 3):
        monkeypatch.delenv("LFI_ENC_STASH_F16", raising=False)
        if products == "auto, fp32 gate stash":
            monkeypatch.setenv("LFI_ENC_STASH_F16", "0")
        m, sd = perturbed_model(hp, gpu_device)
        m.precision = "bf16x3"
        m.train()
        m.injected_masks = masks
        eng = m._ensure_engine(gpu_device)
        eng.backward_products = 3 if products == 3 else "auto"
        _, loss, losses = m(to_dev(batch, gpu_device))
        loss.sum().backward()
        torch.cuda.synchronize()
        assert eng.backward_product_count(B * N) == (3 if products == 3 else 2) and bool(eng._last.chain)
        assert all(eng._last.enc_stash_f16[n] == (products == "auto") for n in ("p2_face", "p1_speech", "p2_speech"))
        got[products] = (torch.stack(losses).double().cpu(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()})
        del m, eng
        torch.cuda.empty_cache()

    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    t0 = time.time()
    sdg = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd.items()}
    _, oloss, onll = oracle.seqglow_forward(hp, sdg, {k: v.double() for k, v in batch.items()},
                                            {k: v.double() for k, v in masks.items()})
    oloss.sum().backward()
    spent = time.time() - t0
    torch.set_num_threads(threads)
    total = float(torch.sqrt(sum((v.grad ** 2).sum() for v in sdg.values() if v.grad is not None)))
    monkeypatch.delenv("LFI_ENC_STASH_F16", raising=False)
    for products, (nll, grads) in got.items():
        err = max_rel(nll, onll.detach(), floor=1.0)
        worst, raw, num = ("", 0.0), ("", 0.0), 0.0
        for name, g in grads.items():
            ref = sdg[name].grad
            diff = float((g - ref).norm())
            num += diff ** 2
            rel = diff / max(float(ref.norm()), 1e-3 * total)
            if rel > worst[1]:
                worst = (name, rel)
            if diff / max(float(ref.norm()), 1e-30) > raw[1]:
                raw = (name, diff / max(float(ref.norm()), 1e-30))
        report("HEADLINE config (B=256, T=80, K=16, bf16x3, planes chain, backward products %s) vs fp64 oracle (%.0f s of host "
               "time): per-frame NLL max rel err %.3e; gradient rel L2: whole %.3e, worst tensor %.3e (%s), worst tensor without "
               "the norm floor %.3e (%s); %d tensors" % (products, spent, err, num ** 0.5 / total, worst[1], worst[0], raw[1],
                                                         raw[0], len(grads)))
        assert err < 1e-4
        assert raw[1] < 2e-3, (products, raw)


def test_k16_sampling_against_oracle(gpu_device):
    """SeqGlow.inference at the full depth (K = 16, final widths), batch 8, 56 generated frames, injected prior noise, against
    the fp64 oracle. north_star's bar is 1e-5 absolute; SURVEY.md par. 7 puts the reference's OWN fp32-vs-fp64 spread at this
    depth x length at ~3e-5, so the gate is max(1e-5, 1.5 x the plain-fp32-torch error on the same inputs); both are reported."""
    hp = final_model_hparams(50, 27, K=16)
    m, sd = perturbed_model(hp, gpu_device)
    m.eval()
    B, seq_len = 8, 24 + 56
    g = torch.Generator().manual_seed(3)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(seq_len - 24, B, 50, generator=g) * 0.8
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))   # (batch-8 ops: a modest thread team is several times faster than all of the box's cores)
    ref = oracle.seqglow_inference(hp, {k: v.double() for k, v in sd.items()}, seq_len,
                                   {k: v.double() for k, v in data.items()}, noise.double())
    ref32 = oracle.seqglow_inference(hp, sd, seq_len, data, noise)    # the same op sequence in plain fp32 (CPU)
    torch.set_num_threads(threads)
    own = float((ref32.double() - ref).abs().max())
    own_rms = float((ref32.double() - ref).pow(2).mean().sqrt())
    gate = max(1e-5, 1.5 * own)
    for precision in ("f32", "bf16x3"):
        m.precision = precision
        out = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))
        out2 = m.inference(seq_len, to_dev(data, gpu_device), noise=noise.to(gpu_device))   # hipGraph replay
        err = float((out.cpu().double() - ref).abs().max())
        per_frame = (out.cpu().double() - ref).abs().amax(dim=(0, 2))
        # (the maximum over 22 400 values of an error that compounds through 56 autoregressive frames is a noisy statistic: any two
        # fp32 roundings of the same computation differ by +-15 % in it; the RMS over all values is the stable one - round 6)
        rms = float((out.cpu().double() - ref).pow(2).mean().sqrt())
        report("K=16 sampling, batch 8 x 56 generated frames (%s): max abs err vs fp64 oracle %.2e (first frame %.2e, last frame "
               "%.2e); plain fp32 torch on the CPU: %.2e; gate %.2e; rms err %.3e, plain fp32 torch rms %.3e"
               % (precision, err, float(per_frame[0]), float(per_frame[-1]), own, gate, rms, own_rms))
        assert torch.equal(out, out2)
        assert err <= gate, (precision, err, gate)


def test_sampler_fused_conditioning_matches_two_gemms(gpu_device, monkeypatch):
    """The sampler's per-frame conditioning as one launch (lfi_sample.hip: c = LeakyReLU(pre_static + window Wct^T) kept in LDS as
    fp16 pieces and multiplied by W_ih[k] at once; glow/models.py:598-615 + the coupling cell's input projection) against the two
    GEMMs it replaces (LFI_SAMPLE_FUSED=0), both in three fp16 products: a ragged batch (64 + 6 rows: two row tiles, one partial),
    6 generated frames at final widths - equal to fp32 rounding carried through the frames, and not bit-equal (it did run)."""
    hp = final_model_hparams(50, 27, K=16)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    m.precision = "bf16x3"
    B, seq_len = 70, 24 + 6
    g = torch.Generator().manual_seed(21)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = (torch.randn(seq_len - 24, B, 50, generator=g) * 0.8).to(gpu_device)
    dd = to_dev(data, gpu_device)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("LFI_SAMPLE_FUSED", fused)
        monkeypatch.setenv("LFI_NO_GRAPH", "1")
        outs[fused] = m.inference(seq_len, dd, noise=noise).clone()
    a, b = outs["1"], outs["0"]
    assert torch.isfinite(a).all()
    err = float((a - b).abs().max() / b.abs().max().clamp(min=1.0))
    report("sampler: fused per-frame conditioning vs its two GEMMs, batch 70 x 6 frames at K = 16: max abs difference %.2e of the "
           "largest value" % err)
    assert not torch.equal(a, b), "the fused conditioning did not run"
    assert err < 2e-5
    # round 5: the conditioning kernel clears the reverse chain's state words itself (no memset node per frame), and can split the
    # window's fp16 pieces itself too (LFI_SAMPLE_XFRAG=0; off by default: measured slower): the same pieces, the same products
    monkeypatch.setenv("LFI_SAMPLE_FUSED", "1")
    monkeypatch.setenv("LFI_SAMPLE_XFRAG", "0")
    c = m.inference(seq_len, dd, noise=noise).clone()
    per_frame = (a - c).abs().amax(dim=(0, 2))
    assert torch.equal(a, c), ("in-kernel window split differs from the fragment kernel, per frame:", per_frame.tolist())
    # ... and by default the reverse chain's step-0 workgroups leave the next frame's fragments (no fragment launch from a run's second
    # frame on); LFI_SAMPLE_XF_CHAIN=0 keeps the fragment kernel in front of every frame: the same fragments
    monkeypatch.delenv("LFI_SAMPLE_XFRAG")
    monkeypatch.setenv("LFI_SAMPLE_XF_CHAIN", "0")
    e = m.inference(seq_len, dd, noise=noise).clone()
    assert torch.equal(a, e), ("fragments left by the chain differ from the fragment kernel's, per frame:", (a - e).abs().amax(dim=(0, 2)).tolist())
    # ... and the reverse cells take their weights as the fp16 hi / lo fragment images lfi_flow_prep leaves (flow_prep_x3h_kernel)
    # instead of splitting the f32 images in registers in every workgroup of every frame (LFI_SAMPLE_WFRAG16=0): the same pieces
    monkeypatch.delenv("LFI_SAMPLE_XF_CHAIN")
    monkeypatch.setenv("LFI_SAMPLE_WFRAG16", "0")
    w = m.inference(seq_len, dd, noise=noise).clone()
    assert torch.equal(a, w), ("pre-split weight fragments differ from the in-register split, per frame:", (a - w).abs().amax(dim=(0, 2)).tolist())
    # round 6: LFI_SAMPLE_COND_ROWS=32 gives the conditioning kernel's workgroups 32 samples instead of 64 (two workgroups per CU;
    # measured slower, kept as a switch): the same products per output element in the same order
    monkeypatch.delenv("LFI_SAMPLE_WFRAG16")
    monkeypatch.setenv("LFI_SAMPLE_COND_ROWS", "32")
    r32 = m.inference(seq_len, dd, noise=noise).clone()
    assert torch.equal(a, r32), ("32-sample conditioning tiles differ from 64-sample ones, per frame:", (a - r32).abs().amax(dim=(0, 2)).tolist())


def test_sampler_static_part_beside_the_chain(gpu_device, monkeypatch):
    """The sampler cuts a call into runs of frames and computes run i + 1's static part (window encoders, static cond_transform
    columns) while run i's chain executes - on a stream that owns some CUs of every XCD (lfi_stream_create_partial), the call itself
    on a private non-blocking stream between two joins with the caller's. Three short runs forced on a ragged batch at final
    widths: the same bits as one run in line, as an ordinary second stream, and as another share of the chip; eager and replayed;
    called from the default stream and from a side stream."""
    hp = final_model_hparams(50, 27, K=16)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    m.precision = "bf16x3"
    B, seq_len = 70, 24 + 11
    g = torch.Generator().manual_seed(23)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = (torch.randn(seq_len - 24, B, 50, generator=g) * 0.8).to(gpu_device)
    dd = to_dev(data, gpu_device)
    monkeypatch.setenv("LFI_SAMPLE_RUNS", "1")
    ref = m.inference(seq_len, dd, noise=noise).clone()
    assert torch.isfinite(ref).all()
    eng = m._ensure_engine(gpu_device)
    for runs, cus in (("3", None), ("3", "0"), ("4,3", "8"), ("3", "16")):
        monkeypatch.setenv("LFI_SAMPLE_RUNS", runs)
        if cus is None:
            monkeypatch.delenv("LFI_SAMPLE_STATIC_CUS", raising=False)
        else:
            monkeypatch.setenv("LFI_SAMPLE_STATIC_CUS", cus)
        assert eng._sample_static_cus(seq_len - 24) == (16 if cus is None else int(cus))
        for _ in range(3):   # eager, captured, replayed
            assert torch.equal(m.inference(seq_len, dd, noise=noise), ref), (runs, cus)
    side = torch.cuda.Stream(device=gpu_device)
    side.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(side):
        out = m.inference(seq_len, dd, noise=noise)
    torch.cuda.current_stream(gpu_device).wait_stream(side)
    assert torch.equal(out, ref)
    monkeypatch.setenv("LFI_NO_OVERLAP", "1")
    assert eng._sample_static_cus(seq_len - 24) == 0
    assert torch.equal(m.inference(seq_len, dd, noise=noise), ref)
    assert set(eng._partial_streams) == {8, 16}
    eng.close()
    assert not eng._partial_streams
    # a driver that refuses the CU mask: a warning, the ordinary second stream from then on, the same frames
    monkeypatch.delenv("LFI_NO_OVERLAP")
    monkeypatch.setenv("LFI_SAMPLE_STATIC_CUS", "16")
    monkeypatch.setattr(eng.L, "lfi_stream_create_partial", lambda *a: -1)
    with pytest.warns(RuntimeWarning, match="partial-chip stream"):
        out = m.inference(seq_len, dd, noise=noise)
    assert torch.equal(out, ref) and eng._partial_refused and eng._sample_static_cus(seq_len - 24) == 0 and not eng._partial_streams
    assert torch.equal(m.inference(seq_len, dd, noise=noise), ref)
    eng._partial_refused = False


def test_sampler_leaves_fp16_pieces_for_out_of_range_inputs(gpu_device):
    """The sampler's default per-frame arithmetic splits operands into fp16 pieces (fp32-grade inside fp16's range). Inputs
    beyond 1e3 must send it to the six-product bf16 form instead (no range caveat): same frames as the all-f32-MFMA mode to
    fp32 accuracy, finite, and the switch is visible in the flow dims the engine hands to the library."""
    hp = final_model_hparams(50, 27, K=4)
    m, _ = perturbed_model(hp, gpu_device)
    m.eval()
    m.precision = "bf16x3"
    B, seq_len = 8, 24 + 6
    g = torch.Generator().manual_seed(11)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    data["p2_speech"][0, 3, 5] = 4.0e3          # one unstandardised value: beyond the guard's 1e3
    noise = (torch.randn(seq_len - 24, B, 50, generator=g) * 0.8).to(gpu_device)
    dd = to_dev(data, gpu_device)
    eng = m._ensure_engine(gpu_device)
    seen = []
    real = eng.L.lfi_flow_sample_seq_from

    def spy(dims, *a):
        seen.append(int(dims._obj.gemm_precision))
        return real(dims, *a)

    eng.L.lfi_flow_sample_seq_from = spy
    try:
        out = m.inference(seq_len, dd, noise=noise)
    finally:
        eng.L.lfi_flow_sample_seq_from = real
    assert seen and seen[0] == 5, seen            # six bf16 products, not fp16 pieces
    eng.sample_frame_precision = 0
    ref = m.inference(seq_len, dd, noise=noise)
    eng.sample_frame_precision = None
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 2e-5 * max(scale, 1.0)
    data["p2_speech"][0, 3, 5] = 4.0              # inside the range: fp16 pieces
    seen.clear()
    eng.L.lfi_flow_sample_seq_from = spy
    try:
        m.inference(seq_len, to_dev(data, gpu_device), noise=noise)
    finally:
        eng.L.lfi_flow_sample_seq_from = real
    assert seen and seen[0] == 9, seen


def test_strong_scaling_anchor_batch_matches_its_sub_batches(gpu_device):
    """bench.py's strong_scaling_anchor steps ONE GPU at the global batch of BASELINE configs[2] (2048 = 8 x 256). That batch
    must be the same function as its eight per-GPU shards: per-frame NLL bit-identical to the shards' (samples are independent),
    and the gradient of the global-batch mean loss equal to the mean of the shards' gradients - which is exactly what the
    data-parallel all-reduce + 1/world computes on 8 GPUs."""
    hp = final_model_hparams(50, 27, K=16)
    m, _ = perturbed_model(hp, gpu_device)
    m.precision = "bf16x3"
    m.eval()   # no dropout: every pass sees the same function
    B, T, W = 2048, 80, 8
    batch = to_dev(oracle.synthetic_batch(B, T, 50, 27, seed=77), gpu_device)

    def grads(b):
        m.zero_grad(set_to_none=True)
        _, loss, losses = m(b)
        loss.sum().backward()
        return torch.stack(losses), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double()

    nll_all, g_all = grads(batch)
    assert torch.isfinite(nll_all).all() and nll_all.shape == (T - 24, B)
    g_sum = torch.zeros_like(g_all)
    for r in range(W):
        shard = {k: v[r * (B // W):(r + 1) * (B // W)].contiguous() for k, v in batch.items()}
        nll_r, g_r = grads(shard)
        assert torch.equal(nll_r, nll_all[:, r * (B // W):(r + 1) * (B // W)]), r
        g_sum += g_r
    err = float((g_all - g_sum / W).norm() / g_all.norm())
    report("global batch 2048 on one GPU vs the mean of its 8 shards of 256: per-frame NLL bit-identical, gradient rel L2 diff %.2e" % err)
    assert err < 2e-5


@pytest.mark.parametrize("products", ["auto", 2])
def test_graphed_training_step_is_bit_identical_to_eager(gpu_device, monkeypatch, products):
    """(products = 2, round 6: the two-product backward at this small size - bf16 gradient rows from the walk, hence the one-pass thin
    weight gradients of csrc/lfi_wgrad.hip on both streams - inside the captured step.)
    VERDICT r2 item 6: with step_graph on (LFI_STEP_GRAPH=1), from its third call a single-GPU fused_training_step is ONE replayed hipGraph (dropout masks,
    forward, backward, clip, Adam; the dropout key and Adam's step size read from a device block set before every replay; the
    negative-example step a second graph). Twelve steps at the benchmark's shape with the negative branch forced on steps 4 and
    9: parameters, Adam moments, the loss of every step and the mismatched-NLL buffer are bit-identical to eager launches."""
    import random
    import numpy as np
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hp = load_hparams_file(os.path.join(root, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    hp["batch_size"] = 64
    hp["gradient_clip_val"] = 20
    hp["engine_backward_products"] = products
    batches = [to_dev(oracle.synthetic_batch(64, 48, 50, 27, seed=100 + i), gpu_device) for i in range(3)]
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LFI_STEP_GRAPH", mode)   # "1": opt in
        random.seed(7)
        np.random.seed(7)
        torch.manual_seed(7)
        m = LetsFaceItGlow(Namespace(**__import__("copy").deepcopy(hp))).to(gpu_device).train()
        forced = iter([False, False, False, False, True, False, False, False, False, True, False, False])
        m._negative_branch = lambda: next(forced)
        losses = [m.fused_training_step(batches[i % 3], 1e-4 * (1 + i % 2)).clone() for i in range(12)]
        eng = m.seq_glow.engine
        assert eng.backward_product_count(64 * 24) == (2 if products == 2 else 3)
        graphs = [v for v in getattr(m, "_step_graphs", {}).values() if isinstance(v, dict)]
        assert len(graphs) == (2 if mode == "1" else 0), "expected the ordinary and the negative-example graph"
        runs[mode] = (torch.stack(losses).cpu(), eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.step_count,
                      eng._mask_calls, m.last_missmatched_nll.clone())
    a, b = runs["0"], runs["1"]
    assert a[4] == b[4] == 12 and a[5] == b[5]
    assert torch.equal(a[0], b[0]), (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[6], b[6])
    report("hipGraph-replayed training step vs eager launches, 12 steps (2 negative): losses, parameters, Adam moments bit-identical")
