"""The CPU oracle against the golden vectors captured from the reference (tests/golden/make_golden.py).

CPU only. This is what pins ``oracle/seqglow_oracle.py`` on machines where /root/reference does not exist.
"""
import pytest
import torch

from oracle import seqglow_oracle as oracle
from helpers import FIXTURES, Fixture, rel_err


@pytest.fixture(scope="module", params=FIXTURES)
def fx(request):
    return Fixture(request.param)


def test_eval_forward(fx):
    z, loss, nll = oracle.seqglow_forward(fx.hp, fx.state_dict(), fx.batch())
    assert z.shape == (fx.N, fx.B, fx.C) and nll.shape == (fx.N, fx.B)
    assert rel_err(z, fx.get("eval/z")) < 1e-10
    assert rel_err(nll, fx.get("eval/nll")) < 1e-10
    assert rel_err(loss, fx.get("eval/loss")) < 1e-10


def test_fp32_oracle_within_reference_noise(fx):
    """fp32 oracle vs fp64 reference: same spread as the reference's own fp32 path (per-frame NLL, rel)."""
    sd, batch = fx.state_dict(torch.float32), fx.batch(torch.float32)
    nll = oracle.seqglow_forward(fx.hp, sd, batch)[2]
    ref64 = fx.get("eval/nll")
    own = ((nll.double() - ref64).abs() / ref64.abs().clamp(min=1.0)).max().item()
    ref = ((fx.get("eval/nll_ref_fp32").double() - ref64).abs() / ref64.abs().clamp(min=1.0)).max().item()
    assert own < 1e-4 and own < 20 * max(ref, 1e-7)


def test_train_forward_and_grads(fx):
    sd = fx.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")):
            v.requires_grad_(True)
    z, loss, nll = oracle.seqglow_forward(fx.hp, sd, fx.batch(), fx.masks())
    assert rel_err(nll.detach(), fx.get("train/nll")) < 1e-10
    loss.sum().backward()
    grads = fx.group("grad/")
    assert grads
    for name, g in grads.items():
        assert rel_err(sd[name].grad, g) < 1e-9, name


def test_adam_clip_step(fx):
    sd = fx.state_dict()
    grads = fx.group("grad/")
    names = list(grads)
    ps = [sd[n].clone() for n in names]
    gs = [grads[n].clone() for n in names]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    betas = fx.hp["Optim"]["args"]["adam"]["betas"]
    gn = oracle.adam_clip_step(ps, gs, m, v, 1, float(fx.get("adam/lr")), betas[0], betas[1],
                               fx.hp["Optim"]["args"]["adam"]["eps"], float(fx.get("adam/clip")))
    assert abs(gn - float(fx.get("adam/grad_norm"))) < 1e-9 * gn
    for n, p in zip(names, ps):
        assert rel_err(p, fx.get("adam/" + n)) < 1e-10, n


def test_negative_step(fx):
    loss = oracle.training_loss(fx.hp, fx.state_dict(), fx.batch(), None, fx.get("neg/perm"))
    assert rel_err(loss, fx.get("neg/loss")) < 1e-10


def test_inference(fx):
    data = fx.group("infer/data/", torch.float64)
    out = oracle.seqglow_inference(fx.hp, fx.state_dict(), int(fx.get("infer/seq_len")), data,
                                   fx.get("infer/noise"))
    assert out.shape == fx.get("infer/out").shape
    assert rel_err(out, fx.get("infer/out")) < 1e-10


def test_invert_roundtrip(fx):
    sd, batch = fx.state_dict(), fx.batch()
    z, loss, _ = oracle.seqglow_forward(fx.hp, sd, batch)
    rec, bl = oracle.seqglow_invert(fx.hp, sd, z, batch)
    # decode(encode(x)) == x and the backward NLL mirrors the forward one
    assert rel_err(rec, batch["p1_face"][:, fx.start:].transpose(0, 1)) < 1e-9
    if fx.has("invert/x"):
        assert rel_err(rec, fx.get("invert/x")) < 1e-10
        assert rel_err(bl, fx.get("invert/loss")) < 1e-10


def test_actnorm_init(fx):
    batch = fx.group("init/batch/", torch.float64)
    masks = fx.masks(prefix="init/mask/")
    sd = oracle.actnorm_init(fx.hp, fx.state_dict(), batch, masks)
    for k in range(oracle.n_flow_steps(fx.hp)):
        for leaf in ("bias", "logs"):
            key = "glow.flow.layers.%d.actnorm.%s" % (k, leaf)
            assert rel_err(sd[key], fx.get("init/" + key)) < 1e-10, key
    nll = oracle.seqglow_forward(fx.hp, sd, batch, masks)[2]
    assert rel_err(nll, fx.get("init/nll")) < 1e-10


def test_reference_op_set_matches_the_spelled_out_cells(fx):
    """`oracle.reference_op_set()` (what bench.py's torch_gpu_baseline times: torch._VF.gru / gru_cell / lstm / lstm_cell, the ops
    nn.GRU / nn.GRUCell / nn.LSTM / nn.LSTMCell dispatch to, glow/models.py:21-27,60-64,176-185,206-213) is the same function as
    the spelled-out cell equations the fixtures pin: forward against the reference's golden NLL, gradients against its golden
    gradients, in fp64."""
    sd = fx.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")):
            v.requires_grad_(True)
    with oracle.reference_op_set():
        z, loss, nll = oracle.seqglow_forward(fx.hp, sd, fx.batch(), fx.masks())
        loss.sum().backward()
    assert rel_err(nll.detach(), fx.get("train/nll")) < 1e-10
    for name, g in fx.group("grad/").items():
        assert rel_err(sd[name].grad, g) < 1e-9, name
    with oracle.reference_op_set():
        data = fx.group("infer/data/", torch.float64)
        out = oracle.seqglow_inference(fx.hp, fx.state_dict(), int(fx.get("infer/seq_len")), data, fx.get("infer/noise"))
    assert rel_err(out, fx.get("infer/out")) < 1e-10
