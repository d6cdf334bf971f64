"""GPU: the module-level API — the only tests the reference holds for this path are its print-only round trips in
code/glow_pytorch/test_modules.py (test_actnorm, test_conv1x1, test_flow_step, test_flow_net). The same calls here, with
assertions, through the HIP cells (lfi_flow_step); then the reference's per-timestep loop body (SeqGlow.create_conditioning
+ Glow.forward, models.py:546-559) against the golden vectors and against the fused SeqGlow.forward."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import Fixture, max_rel, rel_err
from oracle import seqglow_oracle as oracle

pytestmark = pytest.mark.gpu


def test_actnorm(gpu_device):  # test_modules.py:9-18
    from lets_face_it_amd.glow import modules
    actnorm = modules.ActNorm2d(54).to(gpu_device)
    x = torch.tensor(np.random.RandomState(0).rand(6, 54), dtype=torch.float32, device=gpu_device)
    actnorm.initialize_parameters(x)
    assert actnorm.inited
    y, det = actnorm(x, 0)
    x_, det2 = actnorm(y, det, True)
    assert float((x_ - x).abs().max()) < 1e-5 and abs(float(det2)) < 1e-4
    assert float(y.mean(0).abs().max()) < 1e-5                      # data-dependent init: zero mean, unit scale
    assert float((y.std(0, unbiased=False) - 1.0).abs().max()) < 1e-3
    # against the formulas of modules.py:45-66 in torch
    want = (x + actnorm.bias) * torch.exp(actnorm.logs)
    assert rel_err(y, want) < 1e-6 and abs(float(det) - float(actnorm.logs.sum()) * 54) < 1e-3


@pytest.mark.parametrize("lu", [False, True])
def test_conv1x1(gpu_device, lu):  # test_modules.py:21-29
    from lets_face_it_amd.glow import modules
    np.random.seed(0)
    conv = modules.InvertibleConv1x1(96, LU_decomposed=lu).to(gpu_device)
    with torch.no_grad():   # leave the orthogonal initialisation so that the log-det is not trivially 0
        (conv.log_s if lu else conv.weight).add_(0.05 * torch.randn_like(conv.log_s if lu else conv.weight))
    x = torch.tensor(np.random.rand(5, 96), dtype=torch.float32, device=gpu_device)
    y, det = conv(x, 0)
    x_, det2 = conv(y, det, True)
    assert float((x_ - x).abs().max()) < 1e-4 and abs(float(det2)) < 1e-3
    # against the fp64 oracle's weight (P L U resp. the dense matrix; log-det x C, modules.py:171)
    sd = {"w." + k: v.detach().cpu().double() for k, v in conv.state_dict().items()}
    W, ld = oracle.invconv_weight(sd, "w.")
    assert rel_err(y, x.cpu().double() @ W) < 1e-5
    assert abs(float(det) - float(ld) * 96) < 1e-3 * max(1.0, abs(float(ld) * 96))


def _perturb(mod, seed=3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if "final_linear" in name:  # LinearZeros is zero at init: the coupling net would be dead (SURVEY.md finding 5)
                p.add_(torch.randn(p.shape, generator=g) * 0.05)


@pytest.mark.parametrize("rnn,coupling,hidden", [("gru", "affine", 256), ("lstm", "affine", 64), ("gru", "additive", 96)])
def test_flow_step(gpu_device, rnn, coupling, hidden):  # test_modules.py:32-49
    from lets_face_it_amd.glow import models
    torch.manual_seed(0)
    np.random.seed(0)
    step = models.FlowStep(54, hidden, flow_permutation="invconv", flow_coupling=coupling, cond_dim=32,
                           feature_encoder_dim=64, glow_rnn_type=rnn)
    _perturb(step)
    sd = {k: v.detach().clone().double() for k, v in step.state_dict().items()}
    step.to(gpu_device).eval()
    x = torch.tensor(np.random.rand(2, 54), dtype=torch.float32)
    cond = torch.tensor(np.random.rand(2, 64), dtype=torch.float32)
    y, det = step(x.to(gpu_device), cond.to(gpu_device), 0, False)
    assert step.f.hidden is not None and step.f.hidden.shape == (2, hidden)
    # against the CPU oracle (fp64) on the same weights
    hp = {"Glow": {"K": 1, "L": 1, "flow_coupling": coupling, "rnn_type": rnn, "scale_eps": step.scale_eps,
                   "LU_decomposed": False}}
    sd_o = {"glow.flow.layers.0." + k: v for k, v in sd.items()}
    state = oracle._new_state(hp)
    y_o, det_o, _ = oracle.flow_forward(hp, sd_o, x.double(), cond.double(), state)
    assert rel_err(y, y_o) < 1e-5 and rel_err(det, det_o) < 1e-5
    # the reference's round trip re-uses the state the forward call left behind, so it is not an inverse; from the
    # same (zero) state it is
    step.init_rnn_hidden()
    x_, det0 = step(y, cond.to(gpu_device), det, True)
    assert float((x_.cpu() - x).abs().max()) < 1e-4
    assert float(det0.abs().max()) < 1e-4


def test_flow_net(gpu_device):  # test_modules.py:52-67
    from lets_face_it_amd.glow import models
    torch.manual_seed(0)
    np.random.seed(0)
    net = models.FlowNet(C=54, hidden_channels=256, cond_dim=64, K=3, L=1, feature_encoder_dim=32, glow_rnn_type="gru")
    _perturb(net)
    net.to(gpu_device).eval()
    x = torch.tensor(np.random.rand(4, 54), dtype=torch.float32, device=gpu_device)
    cond = torch.tensor(np.random.rand(4, 32), dtype=torch.float32, device=gpu_device)
    y, det = net(x, cond)
    assert y.shape == (4, 54) and det.shape == (4,)
    net.init_rnn_hidden()
    x_, det0 = net(y, cond, reverse=True)
    assert float((x_ - x).abs().max()) < 1e-4
    assert float((det + det0).abs().max()) < 1e-4
    # two timesteps: the recurrent state carries over, like f_seq.hidden in the reference (models.py:204-214)
    net.init_rnn_hidden()
    y1, _ = net(x, cond)
    y2, _ = net(x, cond)
    assert float((y1 - y).abs().max()) == 0.0 and float((y2 - y1).abs().max()) > 1e-6


def test_actnorm_init_through_module_call(gpu_device):
    """First training-mode call initialises every ActNorm from its own input (modules.py:32-43,69-70): afterwards the
    per-layer outputs of the actnorm are standardised, i.e. bias = -mean and logs = log(scale / (std + 1e-6))."""
    from lets_face_it_amd.glow import models
    torch.manual_seed(0)
    np.random.seed(0)
    net = models.FlowNet(C=20, hidden_channels=32, cond_dim=16, K=2, L=1, feature_encoder_dim=8, glow_rnn_type="gru",
                         flow_coupling="affine")
    net.to(gpu_device).train()
    x = torch.randn(64, 20, device=gpu_device) * 3.0 + 1.5
    cond = torch.randn(64, 8, device=gpu_device)
    assert not net.layers[0].actnorm.inited
    net(x, cond)
    assert all(l.actnorm.inited for l in net.layers)
    a0 = net.layers[0].actnorm
    assert rel_err(a0.bias.view(-1), -x.mean(0)) < 1e-5
    var = ((x - x.mean(0)) ** 2).mean(0)
    assert rel_err(a0.logs.view(-1), torch.log(1.0 / (var.sqrt() + 1e-6))) < 1e-5


@pytest.mark.parametrize("name", ["tiny", "tiny_lstm", "odd", "p1enc", "mlp", "framenb", "lstmenc"])
def test_per_timestep_loop_matches_reference(gpu_device, name):
    """The reference's own loop body (SeqGlow.forward, models.py:546-559; invert, :630-641), one timestep per call."""
    from lets_face_it_amd.glow.models import SeqGlow
    fx = Fixture(name)
    m = SeqGlow(Namespace(**fx.hp))
    m.precision = "f32"
    m.load_state_dict(fx.state_dict(torch.float32), strict=True)
    m.to(gpu_device).eval()
    m.glow.set_actnorm_init(True)
    batch = {k: v.to(device=gpu_device, dtype=torch.float32).contiguous() for k, v in fx.batch().items()}
    start = fx.start
    use_nb = fx.hp["Conditioning"]["use_frame_nb"]

    m.glow.init_rnn_hidden()
    frame_nb = batch["frame_nb"].clone() + start * 2 if use_nb else None
    z_seq, losses = [], []
    for t in range(start, fx.T):
        condition = m.create_conditioning(batch, t, frame_nb, batch["p1_face"])
        assert condition.shape == (fx.B, m.feature_encoder.dim)
        z_enc, objective = m.glow(x=batch["p1_face"][:, t, :], condition=condition)
        losses.append(m.loss(objective, z_enc).cpu())
        z_seq.append(z_enc)
        if use_nb:
            frame_nb += 2
    assert max_rel(torch.stack(losses), fx.get("eval/nll"), floor=1.0) < 1e-4
    assert rel_err(torch.stack(z_seq), fx.get("eval/z")) < 1e-5
    with torch.no_grad():
        zf, _, lf = m(batch)  # the fused path on the same weights
    assert max_rel(torch.stack(losses), torch.stack(lf), floor=1.0) < 1e-5

    if fx.has("invert/x"):
        m.glow.init_rnn_hidden()
        frame_nb = batch["frame_nb"].clone() + start * 2 if use_nb else None
        rec = []
        for n, z_enc in enumerate(z_seq):
            condition = m.create_conditioning(batch, start + n, frame_nb, batch["p1_face"])
            x_rec, _ = m.glow(z=z_enc, condition=condition, eps_std=1, reverse=True)
            rec.append(x_rec)
            if use_nb:
                frame_nb += 2
        assert rel_err(torch.stack(rec), fx.get("invert/x")) < 1e-4
