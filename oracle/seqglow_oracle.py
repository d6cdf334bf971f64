"""CPU oracle for the conditional-Glow hot path of jonepatr/lets_face_it.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lets_face_it_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and there only as the checker / the timed CPU port.

It is a functional restatement (plain torch CPU ops on a ``state_dict``; any
float dtype, fp64 for goldens) of the reference algorithm, one timestep and one
flow step at a time exactly as the reference walks them.  Citations are
``path:line`` relative to ``/root/reference/code/glow_pytorch/``.  Every tensor it
creates lives on its inputs' device, so the same op sequence also runs through
stock PyTorch-ROCm on a GPU: that is ``bench.py``'s ``torch_gpu_baseline`` leg
(the "reference single-GPU PyTorch" figure of BASELINE.md par. 3), a timed
baseline like ``cpu_baseline``, never the product.

Parity pin: the reference holds no golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, imported in
the build container by ``tests/golden/make_golden.py`` (which asserts
oracle == reference to 1e-10 in fp64 before it writes the fixtures that
``tests/test_oracle_golden.py`` re-checks everywhere).
"""
import contextlib
import math

import torch

LOG2PI = float(math.log(2.0 * math.pi))  # glow/modules.py:198
LN2 = float(math.log(2.0))  # glow/models.py:565
MODALITIES = ("p2_face", "p1_speech", "p2_speech")  # concat order, glow/models.py:127-143


def longest_history(cond):
    """glow/utils.py:44-50."""
    return max(cond[m]["history"] for m in ("p1_face", "p1_speech", "p2_speech", "p2_face"))


# --------------------------------------------------------------------------- cells
# The cells below spell nn.GRUCell / nn.LSTMCell / nn.GRU / nn.LSTM out as their equations (~10 small ATen ops per cell). The
# reference itself calls the modules (glow/models.py:21-27,60-64,176-185,206-213), which dispatch to ONE fused op per cell
# (torch._VF.gru_cell / lstm_cell: a fused pointwise kernel behind two addmm on a GPU) and ONE per window (torch._VF.gru / lstm:
# MIOpen's RNN on a ROCm GPU). `with reference_op_set():` switches this module to exactly those calls - same arithmetic
# (tests/test_oracle_golden.py pins the two forms to each other), the reference's own kernel count: it is what bench.py's
# `torch_gpu_baseline` times, so that the ">= 10x the reference single-GPU PyTorch" ratio is taken against the reference's
# op set, not against a slower spelling of it.
_REFERENCE_OPS = False


@contextlib.contextmanager
def reference_op_set(on=True):
    global _REFERENCE_OPS
    old, _REFERENCE_OPS = _REFERENCE_OPS, bool(on)
    try:
        yield
    finally:
        _REFERENCE_OPS = old


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell equations (gate order r, z, n); used by glow/models.py:176-179,206-208."""
    if _REFERENCE_OPS:
        return torch._VF.gru_cell(x, h, w_ih, w_hh, b_ih, b_hh)
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    u = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - u) * n + u * h


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell equations (gate order i, f, g, o); glow/models.py:181-185,209-213.

    The reference branch passes (None, None) on the first step and crashes
    (SURVEY.md finding 2); the semantics pinned here are "zero initial (h, c)".
    """
    if _REFERENCE_OPS:
        return torch._VF.lstm_cell(x, (h, c), w_ih, w_hh, b_ih, b_hh)
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[1]
    i = torch.sigmoid(g[:, :H])
    f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def encode_window(x, enc, sd, prefix, mask=None):
    """ModalityEncoder.forward, glow/models.py:55-80.

    x: (B, hist, in).  mask: optional (B, hist) multiplier = Dropout(p)(ones) (:56-58).
    """
    if mask is not None:
        x = x * mask.unsqueeze(-1)
    B = x.shape[0]
    if enc == "rnn":  # nn.GRU from h0 = 0, output cat(seq[:, -1], h_n[0]) (:60-64)
        w_ih, w_hh = sd[prefix + "encoder.weight_ih_l0"], sd[prefix + "encoder.weight_hh_l0"]
        b_ih, b_hh = sd[prefix + "encoder.bias_ih_l0"], sd[prefix + "encoder.bias_hh_l0"]
        h = x.new_zeros(B, w_hh.shape[1])
        if _REFERENCE_OPS:   # nn.GRU(batch_first=True): one fused call per window batch (:60-64); `train` as module.training:
            # dropout is 0 either way, but MIOpen's RNN backward insists on a training-mode forward
            seq, hn = torch._VF.gru(x.contiguous(), h.unsqueeze(0), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, torch.is_grad_enabled(), False, True)
            return torch.cat([seq[:, -1], hn[0]], dim=1)
        for s in range(x.shape[1]):
            h = gru_cell(x[:, s], h, w_ih, w_hh, b_ih, b_hh)
        return torch.cat([h, h], dim=1)
    if enc == "lstm":  # (:65-69)
        w_ih, w_hh = sd[prefix + "encoder.weight_ih_l0"], sd[prefix + "encoder.weight_hh_l0"]
        b_ih, b_hh = sd[prefix + "encoder.bias_ih_l0"], sd[prefix + "encoder.bias_hh_l0"]
        h = x.new_zeros(B, w_hh.shape[1])
        c = x.new_zeros(B, w_hh.shape[1])
        if _REFERENCE_OPS:   # nn.LSTM(batch_first=True) (:65-69)
            seq, hn, _ = torch._VF.lstm(x.contiguous(), (h.unsqueeze(0), c.unsqueeze(0)), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0,
                                        torch.is_grad_enabled(), False, True)
            return torch.cat([seq[:, -1], hn[0]], dim=1)
        for s in range(x.shape[1]):
            h, c = lstm_cell(x[:, s], h, c, w_ih, w_hh, b_ih, b_hh)
        return torch.cat([h, h], dim=1)
    if enc == "mlp":  # Linear + LeakyReLU(0.01) on the flattened window (:70-71)
        y = x.reshape(B, -1) @ sd[prefix + "encoder.0.weight"].t() + sd[prefix + "encoder.0.bias"]
        return torch.nn.functional.leaky_relu(y, 0.01)
    if enc == "none":  # (:76-77)
        return x.reshape(B, -1)
    raise NotImplementedError(enc)


def conditioning(hp, sd, data, t, prev_p1_faces, masks=None, n=None, frame_nb=None):
    """SeqGlow.create_conditioning + FeatureEncoder.forward, glow/models.py:598-615,127-145."""
    cond = hp["Conditioning"]
    h1 = cond["p1_face"]["history"]
    parts = [encode_window(prev_p1_faces[:, t - h1:t], cond["p1_face"]["enc"], sd,
                           "feature_encoder.p1_face_encoder.",
                           None if masks is None or "p1_face" not in masks else masks["p1_face"][n])]
    for m in MODALITIES:
        hist = cond[m]["history"]
        if hist:
            win = data[m][:, t - hist + 1:t + 1]  # includes t (:607-610)
            mk = None if masks is None or m not in masks else masks[m][n]
            parts.append(encode_window(win, cond[m]["enc"], sd, "feature_encoder.%s_encoder." % m, mk))
    if cond["use_frame_nb"]:
        parts.append(frame_nb)
    return torch.cat(parts, dim=1)


# --------------------------------------------------------------------------- flow primitives
def invconv_weight(sd, prefix, reverse=False):
    """InvertibleConv1x1.get_weight, glow/modules.py:147-178. Returns (W, sum(log|s|))."""
    if prefix + "weight" in sd:  # non-LU mode (:151-161)
        w = sd[prefix + "weight"]
        ld = torch.slogdet(w)[1]
        if reverse:
            w = torch.inverse(w.double()).to(w.dtype)
        return w, ld
    l, u, log_s = sd[prefix + "l"], sd[prefix + "u"], sd[prefix + "log_s"]
    p, sign_s = sd[prefix + "p"].to(l.dtype), sd[prefix + "sign_s"].to(l.dtype)
    C = l.shape[0]
    l_mask = torch.tril(torch.ones(C, C, dtype=l.dtype, device=l.device), -1)
    lm = l * l_mask + torch.eye(C, dtype=l.dtype, device=l.device)
    um = u * l_mask.t() + torch.diag(sign_s * torch.exp(log_s))
    if not reverse:
        return p @ (lm @ um), log_s.sum()
    li = torch.inverse(lm.double()).to(l.dtype)  # fp64 inverse then cast (:175-176)
    ui = torch.inverse(um.double()).to(l.dtype)
    return ui @ (li @ torch.inverse(p)), log_s.sum()


def coupling_net(z1, cond_vec, hidden, cell, sd, prefix, rnn_type):
    """f_seq.forward, glow/models.py:204-214 (+ LinearZeros, glow/modules.py:93-95)."""
    c = torch.nn.functional.leaky_relu(
        cond_vec @ sd[prefix + "cond_transform.0.weight"].t() + sd[prefix + "cond_transform.0.bias"], 0.01)
    x = torch.cat([z1, c], dim=1)
    w_hh = sd[prefix + "rnn.weight_hh"]
    if hidden is None:
        hidden = z1.new_zeros(z1.shape[0], w_hh.shape[1])
    if rnn_type == "gru":
        hidden = gru_cell(x, hidden, sd[prefix + "rnn.weight_ih"], w_hh,
                          sd[prefix + "rnn.bias_ih"], sd[prefix + "rnn.bias_hh"])
    else:
        if cell is None:
            cell = torch.zeros_like(hidden)
        hidden, cell = lstm_cell(x, hidden, cell, sd[prefix + "rnn.weight_ih"], w_hh,
                                 sd[prefix + "rnn.bias_ih"], sd[prefix + "rnn.bias_hh"])
    out = (hidden @ sd[prefix + "final_linear.weight"].t() + sd[prefix + "final_linear.bias"]) \
        * torch.exp(sd[prefix + "final_linear.logs"] * 3.0)
    return out, hidden, cell


def n_flow_steps(hp):
    return hp["Glow"]["K"] * hp["Glow"]["L"]  # no squeeze/split between levels, glow/models.py:413-434


def flow_forward(hp, sd, x, cond_vec, state):
    """Glow.normal_flow -> FlowNet.encode -> FlowStep.normal_flow, glow/models.py:504-506,444-451,311-342."""
    C = x.shape[1]
    logdet = torch.zeros_like(x[:, 0])
    eps = hp["Glow"]["scale_eps"]
    scales = []
    for k in range(n_flow_steps(hp)):
        pre = "glow.flow.layers.%d." % k
        # actnorm (glow/modules.py:45-66): log-det scaled by x.size(1) == C (the quirk)
        logs = sd[pre + "actnorm.logs"]
        z = (x + sd[pre + "actnorm.bias"]) * torch.exp(logs)
        logdet = logdet + logs.sum() * C
        # invertible 1x1 (glow/modules.py:180-189): row-vector convention z @ W
        W, ld = invconv_weight(sd, pre + "invconv.")
        z = z @ W
        logdet = logdet + ld * C
        z1, z2 = z[:, :C // 2], z[:, C // 2:]
        h, state[k][0], state[k][1] = coupling_net(z1, cond_vec, state[k][0], state[k][1], sd, pre + "f.",
                                                    hp["Glow"]["rnn_type"])
        if hp["Glow"]["flow_coupling"] == "additive":  # (:330-331)
            z2 = z2 + h
        else:  # (:332-341)
            shift, scale = h[:, 0::2], h[:, 1::2]
            scale = torch.sigmoid(scale + 2.0).clamp(min=eps)
            scales.append(scale)
            z2 = (z2 + shift) * scale
            logdet = torch.log(scale).sum(dim=1) + logdet
        x = torch.cat([z1, z2], dim=1)
    return x, logdet, scales


def flow_reverse(hp, sd, z, cond_vec, state):
    """FlowNet.decode -> FlowStep.reverse_flow, glow/models.py:453-462,345-373."""
    C = z.shape[1]
    logdet = torch.zeros_like(z[:, 0])
    eps = hp["Glow"]["scale_eps"]
    for k in reversed(range(n_flow_steps(hp))):
        pre = "glow.flow.layers.%d." % k
        z1, z2 = z[:, :C // 2], z[:, C // 2:]
        h, state[k][0], state[k][1] = coupling_net(z1, cond_vec, state[k][0], state[k][1], sd, pre + "f.",
                                                    hp["Glow"]["rnn_type"])
        if hp["Glow"]["flow_coupling"] == "additive":
            z2 = z2 - h
        else:
            shift, scale = h[:, 0::2], h[:, 1::2]
            scale = torch.sigmoid(scale + 2.0).clamp(min=eps)
            z2 = z2 / scale
            z2 = z2 - shift
            logdet = -torch.log(scale).sum(dim=1) + logdet
        z = torch.cat([z1, z2], dim=1)
        W, ld = invconv_weight(sd, pre + "invconv.", reverse=True)
        z = z @ W
        logdet = logdet - ld * C
        logs = sd[pre + "actnorm.logs"]
        z = z * torch.exp(-logs)  # scale then center (glow/modules.py:76-79)
        logdet = logdet - logs.sum() * C
        z = z - sd[pre + "actnorm.bias"]
    return z, logdet


def nll_bits(logdet, z):
    """SeqGlow.loss + GaussianDiag.logp_simplified, glow/models.py:563-565, glow/modules.py:201-212."""
    obj = logdet + (-0.5 * (z ** 2 + LOG2PI)).sum(dim=1)
    return -obj / LN2


def _new_state(hp):
    return [[None, None] for _ in range(n_flow_steps(hp))]


# --------------------------------------------------------------------------- sequence level
def seqglow_forward(hp, sd, batch, masks=None, collect_scales=False):
    """SeqGlow.forward, glow/models.py:534-561.

    masks: optional {modality: (N, B, hist)} dropout multipliers (train mode, injected).
    Returns z_seq (N, B, C), loss (1,), losses (N, B) [, scales].
    """
    state = _new_state(hp)  # init_rnn_hidden (:535)
    start = longest_history(hp["Conditioning"])
    T = batch["p1_face"].shape[1]
    frame_nb = None
    if hp["Conditioning"]["use_frame_nb"]:
        frame_nb = batch["frame_nb"].clone() + start * 2
    zs, losses, all_scales = [], [], []
    loss = 0.0
    for n, t in enumerate(range(start, T)):
        x = batch["p1_face"][:, t, :]
        cv = conditioning(hp, sd, batch, t, batch["p1_face"], masks, n, frame_nb)
        z, logdet, scales = flow_forward(hp, sd, x, cv, state)
        nll = nll_bits(logdet, z)
        losses.append(nll)
        loss = loss + nll.mean()
        zs.append(z)
        all_scales.append(scales)
        if frame_nb is not None:
            frame_nb = frame_nb + 2
    out = (torch.stack(zs), (loss / len(zs)).reshape(1), torch.stack(losses))
    return out + (all_scales,) if collect_scales else out


def seqglow_inference(hp, sd, seq_len, data, noise):
    """SeqGlow.inference, glow/models.py:567-596, with the prior noise injected.

    noise: (seq_len - start, B, C) = eps_std * N(0, 1) draws (GaussianDiag.sample, glow/modules.py:231-235).
    Returns (B, seq_len - start, C).
    """
    state = _new_state(hp)
    start = longest_history(hp["Conditioning"])
    faces = data["p1_face"]
    frame_nb = None
    if hp["Conditioning"]["use_frame_nb"]:
        frame_nb = torch.ones(faces.shape[0], 1, dtype=faces.dtype, device=faces.device)
    for n, t in enumerate(range(start, seq_len)):
        cv = conditioning(hp, sd, data, t, faces, None, None, frame_nb)
        x, _ = flow_reverse(hp, sd, noise[n], cv, state)
        faces = torch.cat([faces, x.unsqueeze(1)], dim=1)
        if frame_nb is not None:
            frame_nb = frame_nb + 2
    return faces[:, start:]


def seqglow_invert(hp, sd, z_seq, data):
    """SeqGlow.invert, glow/models.py:617-645. Returns reconstr (N, B, C), backward_loss (1,)."""
    state = _new_state(hp)
    start = longest_history(hp["Conditioning"])
    frame_nb = None
    if hp["Conditioning"]["use_frame_nb"]:
        frame_nb = data["frame_nb"].clone() + start * 2
    rec, loss = [], 0.0
    for n in range(z_seq.shape[0]):
        cv = conditioning(hp, sd, data, start + n, data["p1_face"], None, None, frame_nb)
        x, logdet = flow_reverse(hp, sd, z_seq[n], cv, state)
        loss = loss + nll_bits(logdet, z_seq[n]).mean()
        rec.append(x)
        if frame_nb is not None:
            frame_nb = frame_nb + 2
    return torch.stack(rec), (loss / z_seq.shape[0]).reshape(1)


def actnorm_init(hp, sd, batch, masks=None):
    """Data-dependent ActNorm init as the first training-mode forward performs it.

    glow/modules.py:32-43,69-70: layer k is initialised from the first timestep's batch as
    transformed by the already-initialised layers < k. Returns a new state_dict.
    """
    sd = dict(sd)
    start = longest_history(hp["Conditioning"])
    x = batch["p1_face"][:, start, :]
    frame_nb = None
    if hp["Conditioning"]["use_frame_nb"]:
        frame_nb = batch["frame_nb"].clone() + start * 2
    cv = conditioning(hp, sd, batch, start, batch["p1_face"], masks, 0, frame_nb)
    state = _new_state(hp)
    C = x.shape[1]
    eps = hp["Glow"]["scale_eps"]
    scale0 = float(hp["Glow"]["actnorm_scale"])
    for k in range(n_flow_steps(hp)):
        pre = "glow.flow.layers.%d." % k
        bias = -x.mean(dim=0, keepdim=True)
        var = ((x + bias) ** 2).mean(dim=0, keepdim=True)
        logs = torch.log(scale0 / (torch.sqrt(var) + 1e-6))
        sd[pre + "actnorm.bias"], sd[pre + "actnorm.logs"] = bias, logs
        z = (x + bias) * torch.exp(logs)
        W, _ = invconv_weight(sd, pre + "invconv.")
        z = z @ W
        z1, z2 = z[:, :C // 2], z[:, C // 2:]
        h, state[k][0], state[k][1] = coupling_net(z1, cv, None, None, sd, pre + "f.", hp["Glow"]["rnn_type"])
        if hp["Glow"]["flow_coupling"] == "additive":
            z2 = z2 + h
        else:
            z2 = (z2 + h[:, 0::2]) * torch.sigmoid(h[:, 1::2] + 2.0).clamp(min=eps)
        x = torch.cat([z1, z2], dim=1)
    return sd


# --------------------------------------------------------------------------- training-step level
def training_loss(hp, sd, batch, masks=None, negative_perm=None):
    """LetsFaceItGlow.training_step, glow/lets_face_it_glow.py:39-54, with the branch decided by the caller.

    negative_perm: None for the ordinary step; a batch permutation to run the
    mismatched step (derange_batch on the p2 modalities, glow/utils.py:85-100; loss * -0.1).
    """
    if negative_perm is None:
        return seqglow_forward(hp, sd, batch, masks)[1]
    mixed = dict(batch)
    for m in ("p2_face", "p2_speech"):
        if hp["Conditioning"][m]["history"] > 0:
            mixed[m] = batch[m][negative_perm]
    return seqglow_forward(hp, sd, mixed, masks)[1] * -0.1


def adam_clip_step(params, grads, m, v, step, lr, beta1, beta2, eps, clip):
    """clip_grad_norm_(clip) then torch.optim.Adam (no weight decay, no amsgrad).

    glow/lets_face_it_glow.py:61-72; hparams/final_model.yaml:126,130. Lists of tensors, updated in place.
    Returns the total grad norm before clipping.
    """
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    coef = min(1.0, clip / (float(total) + 1e-6)) if clip and clip > 0 else 1.0
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    for p, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        m_.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v_.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m_, denom, value=-lr / bc1)
    return float(total)


# --------------------------------------------------------------------------- synthetic model / data
def synthetic_batch(B, T, C, S, seed=1234, dtype=torch.float32):
    """g = Generator(1234); p1_face, p2_face ~ N(0,1) (B,T,C); p1_speech, p2_speech ~ N(0,1) (B,T,S)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, d in (("p1_face", C), ("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        out[name] = torch.randn(B, T, d, generator=g, dtype=torch.float32).to(dtype)
    return out
