"""SeqGlow and its sub-modules with the constructor signatures, attribute names, hparams handling, return tuples
and state-dict keys of glow_pytorch/glow/models.py — computed by the fused HIP engine.

Reference -> here
  ModalityEncoder (models.py:12-80), FeatureEncoder (:83-145), f_seq (:148-214), FlowStep (:217-376),
  FlowNet (:379-467), Glow (:470-521): parameter containers built in the reference's construction order (so the same
  torch / numpy seeds give the same initial weights) whose tensors are views of the engine's flat buffer.
  SeqGlow.forward (:534-561), .inference (:567-596), .invert (:617-645): one call each into
  lets_face_it_amd.engine.GlowEngine instead of the Python loops over timesteps and flow steps.

Differences that are deliberate:
  * GPU only. A CPU tensor raises; there is no eager fallback.
  * Dropout masks of the window encoders come from one torch.bernoulli per modality per call (same distribution,
    different random stream than the per-timestep nn.Dropout calls); `injected_masks` overrides them for parity tests.
  * The recurrent state of the coupling nets lives in the engine for the duration of one call, so an instance is
    re-entrant across calls (the reference keeps it on f_seq.hidden, models.py:193-194).
"""
import torch
import torch.nn as nn

from .. import engine as _engine
from . import modules
from .utils import get_longest_history

_LN2 = 0.6931471805599453


class ModalityEncoder(nn.Module):
    def __init__(self, input_size, params):
        super().__init__()
        self.dropout = nn.Dropout(params["dropout"]) if params["dropout"] > 0 else None
        kind = params["enc"]
        if kind == "rnn":
            self.encoder = nn.GRU(input_size=input_size, hidden_size=params["hidden_dim"], batch_first=True)
            self.dim = params["hidden_dim"] * 2
        elif kind == "lstm":
            self.encoder = nn.LSTM(input_size=input_size, hidden_size=params["hidden_dim"], batch_first=True)
            self.dim = params["hidden_dim"] * 2
        elif kind == "mlp":
            self.encoder = nn.Sequential(nn.Linear(input_size * params["history"], params["hidden_dim"]), nn.LeakyReLU())
            self.dim = params["hidden_dim"]
        elif kind == "none":
            self.encoder = None
            self.dim = input_size * params["history"]
        else:
            raise NotImplementedError(kind)
        self.kind = kind


class FeatureEncoder(nn.Module):
    def __init__(self, conditioning_hparams, data_hparams):
        super().__init__()
        c = conditioning_hparams
        self.use_frame_nb = c["use_frame_nb"]
        self.p1_speech_history = c["p1_speech"]["history"]
        self.p2_speech_history = c["p2_speech"]["history"]
        self.p2_face_history = c["p2_face"]["history"]
        speech_dim = data_hparams["speech_dim"]
        self.p1_face_encoder = ModalityEncoder(c["p1_face"]["dim"], c["p1_face"])
        self.dim = self.p1_face_encoder.dim
        if self.p2_face_history:
            self.p2_face_encoder = ModalityEncoder(c["p2_face"]["dim"], c["p2_face"])
            self.dim += self.p2_face_encoder.dim
        if self.p1_speech_history:
            self.p1_speech_encoder = ModalityEncoder(speech_dim, c["p1_speech"])
            self.dim += self.p1_speech_encoder.dim
        if self.p2_speech_history:
            self.p2_speech_encoder = ModalityEncoder(speech_dim, c["p2_speech"])
            self.dim += self.p2_speech_encoder.dim
        if self.use_frame_nb:
            self.dim += 1
        self._encode = None  # set by the owning SeqGlow: the engine call behind forward()

    def forward(self, condition):
        """One timestep's conditioning dict -> (B, dim) feature vector (models.py:127-145), through the same HIP window
        encoders as the fused path (dropout: fresh masks in training mode, as ModalityEncoder.forward draws them)."""
        if self._encode is None:
            raise RuntimeError("FeatureEncoder.forward needs its SeqGlow (the window encoders run on that model's engine)")
        return self._encode(condition)


class f_seq(nn.Module):  # noqa: N801 - reference class name
    """Coupling net: cond_transform (Linear + LeakyReLU) -> GRUCell/LSTMCell -> LinearZeros."""

    def __init__(self, input_size, output_size, hidden_size, cond_dim, feature_encoder_dim, rnn_type):
        super().__init__()
        self.hidden_size, self.input_size, self.output_size, self.rnn_type = hidden_size, input_size, output_size, rnn_type
        if rnn_type == "gru":
            self.rnn = nn.GRUCell(input_size=input_size + cond_dim, hidden_size=hidden_size)
        elif rnn_type == "lstm":
            self.rnn = nn.LSTMCell(input_size=input_size + cond_dim, hidden_size=hidden_size)
        self.cond_transform = nn.Sequential(nn.Linear(feature_encoder_dim, cond_dim), nn.LeakyReLU())
        self.final_linear = modules.LinearZeros(hidden_size, output_size)
        self.hidden = None
        self.cell = None

    def init_rnn_hidden(self):
        self.hidden = None
        self.cell = None


def _flow_named_flat(layers):
    """(engine layout name, flow-step index, tensor) for the parameters of a list of FlowStep modules."""
    out = []
    for k, layer in enumerate(layers):
        out.append(("flow.an_bias", k, layer.actnorm.bias))
        out.append(("flow.an_logs", k, layer.actnorm.logs))
        if layer.invconv.LU:
            out.append(("flow.inv_l", k, layer.invconv.l))
            out.append(("flow.inv_u", k, layer.invconv.u))
            out.append(("flow.inv_logs", k, layer.invconv.log_s))
        else:
            out.append(("flow.inv_w", k, layer.invconv.weight))
        f = layer.f
        out.append(("flow.w_ih", k, f.rnn.weight_ih))
        out.append(("flow.w_hh", k, f.rnn.weight_hh))
        out.append(("flow.b_ih", k, f.rnn.bias_ih))
        out.append(("flow.b_hh", k, f.rnn.bias_hh))
        out.append(("flow.wct", k, f.cond_transform[0].weight))
        out.append(("flow.bct", k, f.cond_transform[0].bias))
        out.append(("flow.w_fl", k, f.final_linear.weight))
        out.append(("flow.b_fl", k, f.final_linear.bias))
        out.append(("flow.l_fl", k, f.final_linear.logs))
    return out


def _bind_flat(eng, named, layers, device):
    """Move the listed tensors into the engine's flat parameter buffer (values kept) and alias the module tensors to it."""
    for name, k, t in named:
        v = eng.view(name)
        dst = v if k is None else v[k]
        dst.view(-1).copy_(t.detach().reshape(-1).to(device=device, dtype=torch.float32))
        t.data = dst.view(t.shape)
    for k, layer in enumerate(layers):
        if layer.invconv.LU:
            eng.inv_p[k].copy_(layer.invconv.p.to(device))
            eng.inv_sign[k].copy_(layer.invconv.sign_s.to(device))
            layer.invconv.p.data = eng.inv_p[k]
            layer.invconv.sign_s.data = eng.inv_sign[k]


def _is_bound(eng, named):
    lo, hi = eng.params.data_ptr(), eng.params.data_ptr() + 4 * eng.n_params
    return all(lo <= t.data_ptr() < hi for _, _, t in named)


class _FlowRuntime:
    """Module-level calls of FlowStep / FlowNet (what the reference's test_modules.py exercises, models.py:304-308,
    436-462): the same HIP cells as SeqGlow's fused walk, one (B, C) batch and one timestep per call through
    lfi_flow_step, with the recurrent state kept on the modules (f_seq.hidden / .cell) as the reference keeps it.
    Inference-only: gradients flow through SeqGlow.forward (the training path), not through these calls."""

    _engine = None
    allreduce_hook = None

    def _flow_layers(self):
        raise NotImplementedError

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._engine = None  # .to() / .cuda() re-create parameter storage: re-bind lazily
        return out

    def _runtime(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("%s (lets_face_it_amd) runs on the GPU only; got a %s tensor (there is no CPU fallback by "
                               "design)" % (type(self).__name__, device.type))
        layers = self._flow_layers()
        named = _flow_named_flat(layers)
        eng = self._engine
        if eng is None or eng.device != device or eng.spec.Ks != len(layers) or not _is_bound(eng, named):
            l0 = layers[0]
            lin = l0.f.cond_transform[0]
            spec = _engine.ModelSpec.flow_only(
                C=l0.actnorm.num_features, H=l0.f.hidden_size, D=lin.out_features, Ks=len(layers), E=lin.in_features,
                affine=l0.flow_coupling == "affine", rnn_type=l0.f.rnn_type, lu=l0.invconv.LU, scale_eps=l0.scale_eps,
                actnorm_scale=l0.actnorm.scale)
            eng = _engine.GlowEngine(spec, device)
            _bind_flat(eng, named, layers, device)
            self._engine = eng
        return eng

    def _run_flow(self, input_, condition, logdet, reverse):
        eng = self._runtime(input_.device)
        layers = self._flow_layers()
        lstm = eng.spec.rnn_type == "lstm"
        h_prev = [l.f.hidden for l in layers]
        c_prev = [l.f.cell for l in layers] if lstm else None
        if all(h is None for h in h_prev):
            h_prev = c_prev = None
        elif any(h is None for h in h_prev):
            raise RuntimeError("some flow steps carry a recurrent state and others do not: call init_rnn_hidden() first")
        init = None
        if self.training and not reverse and not all(l.actnorm.inited for l in layers):
            hook = self.allreduce_hook
            init = hook if hook is not None else (lambda sums: 1)
        with torch.no_grad():
            out, ld, h_new, c_new = eng.flow_net(input_.float(), condition.float(), h_prev, c_prev, reverse, init)
        for k, l in enumerate(layers):
            l.f.hidden = h_new[k]
            l.f.cell = c_new[k] if lstm else None
            if init is not None:
                l.actnorm.inited = True
        if logdet is None:
            return out, None
        return out, ld + logdet


class FlowStep(_FlowRuntime, nn.Module):
    FlowCoupling = ["additive", "affine"]
    FlowPermutation = ["reverse", "shuffle", "invconv"]

    def __init__(self, in_channels, hidden_channels, cond_dim, actnorm_scale=1.0, flow_permutation="shuffle",
                 flow_coupling="additive", LU_decomposed=False, L=1, K=1, scale_eps=1e-6, scale_logging=False,
                 feature_encoder_dim=0, glow_rnn_type=None):
        assert flow_coupling in FlowStep.FlowCoupling, "flow_coupling should be in `{}`".format(FlowStep.FlowCoupling)
        assert flow_permutation in FlowStep.FlowPermutation, \
            "float_permutation should be in `{}`".format(FlowStep.FlowPermutation)
        if flow_permutation != "invconv":
            raise NotImplementedError("only flow_permutation='invconv' is usable (Permute2d is broken in the reference)")
        super().__init__()
        self.flow_permutation, self.flow_coupling = flow_permutation, flow_coupling
        self.scale = None  # last coupling scales, kept when scale_logging (read by MimicryLogger.log_scales)
        self.scale_logging, self.scale_eps, self.L, self.K = scale_logging, scale_eps, L, K
        self.actnorm = modules.ActNorm2d(in_channels, actnorm_scale)
        self.invconv = modules.InvertibleConv1x1(in_channels, LU_decomposed=LU_decomposed)
        if flow_coupling == "additive":
            out = in_channels - in_channels // 2
        else:
            out = in_channels if in_channels % 2 == 0 else in_channels + 1
        self.f = f_seq(in_channels // 2, out, hidden_channels, cond_dim, feature_encoder_dim, glow_rnn_type)

    def init_rnn_hidden(self):
        self.f.init_rnn_hidden()

    def _flow_layers(self):
        return [self]

    def forward(self, input_, audio_features, logdet=None, reverse=False):
        """One flow step on a (B, C) batch (models.py:304-376): actnorm -> invconv -> coupling, or its inverse."""
        assert audio_features is not None  # "Require some conditioning" (models.py:326)
        return self._run_flow(input_, audio_features, logdet, reverse)

    def normal_flow(self, input_, condition, logdet):
        return self.forward(input_, condition, logdet, reverse=False)

    def reverse_flow(self, input_, condition, logdet):
        return self.forward(input_, condition, logdet, reverse=True)


class FlowNet(_FlowRuntime, nn.Module):
    def __init__(self, C, hidden_channels, cond_dim, K, L, actnorm_scale=1.0, flow_permutation="invconv",
                 flow_coupling="additive", LU_decomposed=False, scale_eps=1e-6, scale_logging=False,
                 feature_encoder_dim=0, glow_rnn_type=None):
        super().__init__()
        self.layers = nn.ModuleList()
        self.output_shapes = []
        self.K, self.L = K, L
        for level in range(L):  # no squeeze / split between levels (models.py:413-434)
            for k in range(K):
                self.layers.append(FlowStep(
                    in_channels=C, hidden_channels=hidden_channels, cond_dim=cond_dim, actnorm_scale=actnorm_scale,
                    flow_permutation=flow_permutation, flow_coupling=flow_coupling, LU_decomposed=LU_decomposed,
                    L=level, K=k, scale_eps=scale_eps, scale_logging=scale_logging,
                    feature_encoder_dim=feature_encoder_dim, glow_rnn_type=glow_rnn_type))
                self.output_shapes.append([-1, C])

    def init_rnn_hidden(self):
        for layer in self.layers:
            layer.init_rnn_hidden()

    def _flow_layers(self):
        return list(self.layers)

    def forward(self, input_, condition, logdet=0.0, reverse=False, eps_std=None):
        if not reverse:
            return self.encode(input_, condition, logdet)
        return self.decode(input_, condition, eps_std)

    def encode(self, z, condition, logdet=0.0):
        """All K*L flow steps forward on one (B, C) batch (models.py:444-451)."""
        return self._run_flow(z, condition, logdet, reverse=False)

    def decode(self, z, condition, eps_std=None):
        """All flow steps in reverse order (models.py:453-462); the returned log-det starts from 0."""
        return self._run_flow(z, condition, 0.0, reverse=True)


class Glow(nn.Module):
    def __init__(self, hparams, feature_encoder_dim=0):
        super().__init__()
        g = hparams.Glow
        self.flow = FlowNet(
            C=hparams.Conditioning["p1_face"]["dim"], hidden_channels=g["hidden_channels"],
            cond_dim=hparams.Conditioning["cond_dim"], K=g["K"], L=g["L"], actnorm_scale=g["actnorm_scale"],
            flow_permutation=g["flow_permutation"], flow_coupling=g["flow_coupling"], LU_decomposed=g["LU_decomposed"],
            scale_eps=g["scale_eps"], scale_logging=hparams.Validation["scale_logging"],
            feature_encoder_dim=feature_encoder_dim, glow_rnn_type=g["rnn_type"])

    def forward(self, x=None, condition=None, z=None, eps_std=None, reverse=False, output_shape=None):
        """One timestep of the flow (models.py:489-513): (z, log-det) forward; (x, log-det) from z / a prior draw in reverse."""
        if not reverse:
            return self.normal_flow(x, condition)
        return self.reverse_flow(z, condition, eps_std, output_shape)

    def normal_flow(self, x, condition):
        return self.flow(x, condition, logdet=torch.zeros_like(x[:, 0]), reverse=False)

    def reverse_flow(self, z, condition, eps_std, output_shape):
        with torch.no_grad():
            if z is None:
                z = modules.GaussianDiag.sample(output_shape, eps_std)
            return self.flow(z, condition, eps_std=eps_std, reverse=True)

    def set_actnorm_init(self, inited=True):
        for m in self.modules():
            if isinstance(m, modules.ActNorm2d):
                m.inited = inited

    def actnorm_inited(self):
        return all(m.inited for m in self.modules() if isinstance(m, modules.ActNorm2d))

    def init_rnn_hidden(self):
        self.flow.init_rnn_hidden()


class _SeqGlowNLL(torch.autograd.Function):
    """Autograd bridge: loss.backward() on the returned loss runs the hand-written backward kernels."""

    @staticmethod
    def forward(ctx, model, batch, masks, init, *params):
        eng = model.engine
        z, nll = eng.forward(batch, masks, with_stash=True, init_actnorm=init)
        ctx.model = model
        ctx.count = nll.numel()
        ctx.fwd_id = model._fwd_counter
        ctx.mark_non_differentiable(z, nll)
        return nll.mean().reshape(1), z, nll

    @staticmethod
    def backward(ctx, gloss, gz, gnll):
        model = ctx.model
        if ctx.fwd_id != model._fwd_counter:
            raise RuntimeError("SeqGlow: backward() after another forward() of the same module; the activation stash "
                               "belongs to the most recent forward (retain only one graph per module)")
        eng = model.engine
        eng.backward(float(gloss.item()) / ctx.count)
        flat = eng.grads.clone()  # own storage: autograd may keep these views as .grad across later backward calls
        grads = []
        for name, p in model._param_map:
            g = eng.view(str(name), flat)
            grads.append((g if name.k is None else g[name.k]).view(p.shape))
        return (None, None, None, None) + tuple(grads)


class SeqGlow(nn.Module):
    def __init__(self, hparams) -> None:
        super().__init__()
        self.hparams = hparams
        if not hparams.Glow.get("rnn_type"):
            hparams.Glow["rnn_type"] = "gru"  # get_hparams default (utils.py:32-33)
        self.feature_encoder = FeatureEncoder(hparams.Conditioning, hparams.Data)
        self.glow = Glow(hparams, self.feature_encoder.dim)
        object.__setattr__(self.feature_encoder, "_encode", self._encode_condition)  # not a submodule / parameter
        self.spec = _engine.ModelSpec(hparams)
        assert self.spec.E == self.feature_encoder.dim
        self.engine = None
        # GEMM arithmetic (include/lfi.h lfi_gemm_desc.precision). "bf16x3" (default): every fp32 operand is split into
        # bf16 hi + lo on the fly and each product is three bf16 MFMAs into fp32 accumulators (~2^-16 relative per product:
        # per-frame NLL within 1e-6 of the fp64 reference, two orders inside the 1e-4 gate, tests/test_gpu_parity.py);
        # "f32": bit-exact fp32 FMA chains on the f32-input MFMA (1/16 of the bf16 rate). Everything outside the GEMMs is
        # fp32 in both modes. hparams key `engine_precision` or the LFI_PRECISION environment variable select it.
        import os
        self.precision = str(getattr(hparams, "engine_precision", None) or os.environ.get("LFI_PRECISION") or "bf16x3")
        if self.precision not in ("f32", "bf16x3"):
            raise ValueError("engine_precision must be 'f32' or 'bf16x3', got %r" % self.precision)
        self.injected_masks = None  # {modality: (N, B, hist)} overrides the random dropout masks (tests)
        self.mask_seed_offset = 0   # data parallelism: the rank, so that every rank draws other masks
        self.allreduce_hook = None  # set by the data-parallel trainer: sums ActNorm init statistics over ranks
        self._param_map = []
        self._fwd_counter = 0

    # ------------------------------------------------------------------ engine binding
    def _named_flat(self):
        """(engine layout name, index, tensor) for every parameter / buffer that lives in the flat buffers."""
        out = []
        fe = self.feature_encoder
        for e in self.spec.encoders:
            if e.enc in ("rnn", "lstm"):
                gru = getattr(fe, e.name + "_encoder").encoder   # nn.GRU / nn.LSTM: same leaf names
                for leaf in _engine.ENC_LEAVES:
                    out.append(("enc.%s.%s" % (e.name, leaf), None, getattr(gru, leaf + "_l0")))
            elif e.enc == "mlp":
                lin = getattr(fe, e.name + "_encoder").encoder[0]
                out.append(("enc.%s.mlp_weight" % e.name, None, lin.weight))
                out.append(("enc.%s.mlp_bias" % e.name, None, lin.bias))
        out.extend(_flow_named_flat(self.glow.flow.layers))
        return out

    def _bind(self, device):
        """Move every parameter into the engine's flat buffer (values preserved) and alias the module tensors to it."""
        if self.spec.rnn_type not in ("gru", "lstm"):
            raise NotImplementedError("Glow.rnn_type=%r: the coupling cell is 'gru' or 'lstm' (models.py:176-185)"
                                      % self.spec.rnn_type)
        prev = getattr(self, "engine", None)
        if prev is not None:
            prev.close()       # (a re-bind leaked the previous engine's CU-masked streams)
        eng = _engine.GlowEngine(self.spec, device)

        _bind_flat(eng, self._named_flat(), self.glow.flow.layers, device)
        old, self._stale_opt = getattr(self, "_stale_opt", None), None
        if old is not None and old[0] == eng.n_params:
            eng.load_optimizer_state(old[1])   # .to()/.float() mid-training keeps Adam's moments, step count and mask counter
        self.engine = eng
        self.glow.flow._engine = eng  # module-level Glow / FlowNet calls (one timestep) run on the same buffers
        # autograd sees each Parameter; map them to gradient views of the flat gradient buffer
        self._param_map = []
        for name, k, t in self._named_flat():
            self._param_map.append((_SlotName(name, k), t))
        return eng

    def _ensure_engine(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("SeqGlow (lets_face_it_amd) runs on the GPU only; got a %s tensor. Move the batch and the "
                               "module to cuda (there is no CPU fallback by design)" % device.type)
        eng = self.engine
        if eng is None or eng.device != device or not self._still_bound():
            eng = self._bind(device)
        eng.precision = 1 if self.precision == "bf16x3" else 0
        bp = getattr(self.hparams, "engine_backward_products", None)
        if bp is not None:
            eng.backward_products = eng.check_backward_products(bp)
        return eng

    def _still_bound(self):
        return _is_bound(self.engine, self._named_flat())

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.float() may re-create parameter storage: then the binding is dropped and rebuilt lazily, with the
        # optimiser state carried over. A no-op move (same device, same dtype: Trainer.fit's model.to(device) on a second
        # fit()) leaves the parameters aliased to the flat buffer and the engine — Adam moments, step count, captured
        # sampling graphs — stays as it is.
        out = super()._apply(fn, *args, **kwargs)
        if self.engine is not None and not self._still_bound():
            # only Adam's moments + counters survive (on the host): the dropped engine's workspaces and stashes - tens of GB
            # at BASELINE sizes - are released now, not when the next engine is already allocated beside them
            st = self.engine.optimizer_state()
            self._stale_opt = (self.engine.n_params, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()})
            self.engine = None
            self.glow.flow._engine = None
        return out

    # ------------------------------------------------------------------ helpers
    def _draw_masks(self, B, N, device):
        if self.injected_masks is not None:
            return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in self.injected_masks.items()}
        if not self.training:
            return None
        # nn.Dropout on ones(B, hist) per timestep (models.py:56-58): all modalities in one HIP launch, Philox keyed on torch's
        # seed (torch.manual_seed / seed_everything) + this module's rank offset (the data-parallel trainer sets it) + a call counter
        eng = self._ensure_engine(device)
        return eng.draw_masks(B, N, torch.initial_seed() + self.mask_seed_offset)

    def _allreduce(self):
        hook = self.allreduce_hook
        return hook if hook is not None else (lambda sums: 1)

    # ------------------------------------------------------------------ reference surface
    def forward(self, batch):
        """-> (z_seq: list[N] of (B, C), loss: (1,), losses: list[N] of (B,) CPU tensors)   (models.py:534-561)"""
        x = batch["p1_face"]
        eng = self._ensure_engine(x.device)
        B, T = x.shape[0], x.shape[1]
        N = T - get_longest_history(self.hparams.Conditioning)
        masks = self._draw_masks(B, N, x.device)
        init = None
        if self.training and not self.glow.actnorm_inited():
            init = self._allreduce()
        self._fwd_counter += 1
        params = [t for _, t in self._param_map]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            loss, z, nll = _SeqGlowNLL.apply(self, batch, masks, init, *params)
        else:
            z, nll = eng.forward(batch, masks, with_stash=False, init_actnorm=init)
            loss = nll.mean().reshape(1)
        if init is not None:
            self.glow.set_actnorm_init(True)
        if self.hparams.Validation["scale_logging"]:
            self._publish_scales(eng, B, N)
        losses = list(nll.detach().cpu().unbind(0))  # one device->host copy instead of one per timestep (:554)
        return list(z.unbind(0)), loss, losses

    def _publish_scales(self, eng, B, N):
        """FlowStep.scale of the last timestep, as the reference leaves it after forward (models.py:336-337)."""
        if not self.spec.affine:
            return
        import ctypes as C
        dims = eng._flow_dims(B, N)
        stash = eng._ws.get("flow_stash")
        if stash is None:
            return
        off = (eng.L.lfi_flow_stash_ptr(C.byref(dims), stash.data_ptr(), 5) - stash.data_ptr()) // 4
        s = self.spec
        ldo = (s.Cout + 3) // 4 * 4  # stash rows are padded to 4 floats (lfi.h, lfi_flow_stash_floats)
        o = stash[off:off + s.Ks * N * B * ldo].view(s.Ks, N, B, ldo)[..., :s.Cout]
        for k, layer in enumerate(self.glow.flow.layers):
            layer.scale = torch.sigmoid(o[k, N - 1, :, 1::2] + 2.0).clamp(min=s.scale_eps)

    def loss(self, objective, z):
        return (-(objective + modules.GaussianDiag.logp_simplified(z))) / _LN2

    def inference(self, seq_len, data=None, noise=None):
        """Autoregressive sampling (models.py:567-596). noise: optional (seq_len - start, B, C) prior draws * eps."""
        seed = data["p1_face"]
        eng = self._ensure_engine(seed.device)
        start = get_longest_history(self.hparams.Conditioning)
        B = seed.shape[0]
        if noise is None:
            shape = torch.zeros(seq_len - start, B, self.spec.C, device=seed.device)
            noise = modules.GaussianDiag.sample(shape, self.hparams.Infer["eps"])
        masks = self._draw_masks(B, seq_len - start, seed.device)
        with torch.no_grad():
            return eng.sample(seq_len, data, noise.contiguous().float(), masks)

    def invert(self, z_seq, data):
        """-> (reconstr_seq: list[N] of (B, C), backward_loss (1,))   (models.py:617-645)"""
        x = data["p1_face"]
        eng = self._ensure_engine(x.device)
        z = torch.stack(list(z_seq)).contiguous().float()
        masks = self._draw_masks(x.shape[0], z.shape[0], x.device)
        with torch.no_grad():
            rec, logdet = eng.invert(z, data, masks)
            logp = modules.GaussianDiag.logp_simplified(z.view(-1, z.shape[-1])).view_as(logdet)
            backward_loss = (-(logdet + logp) / _LN2).mean().reshape(1)
        return list(rec.unbind(0)), backward_loss

    def create_conditioning(self, data, time_st, frame_nb, prev_p1_faces):
        """The conditioning windows of ONE timestep, encoded (models.py:598-615). SeqGlow.forward / inference do this for
        all timesteps at once; this is the reference's per-timestep call for callers that drive Glow.forward themselves."""
        c = self.hparams.Conditioning
        h1 = c["p1_face"]["history"]
        output = {"prev_p1_face": prev_p1_faces[:, time_st - h1:time_st]}
        for modality in ("p1_speech", "p2_speech", "p2_face"):
            history = c[modality]["history"]
            if history:
                output[modality] = data[modality][:, (time_st - history) + 1:time_st + 1]
        if c["use_frame_nb"]:
            output["frame_nb"] = frame_nb
        return self.feature_encoder(output)

    def _encode_condition(self, condition):
        dev = condition["prev_p1_face"].device
        eng = self._ensure_engine(dev)
        cond = {k: v.to(dtype=torch.float32).contiguous() for k, v in condition.items()}
        B = cond["prev_p1_face"].shape[0]
        masks = self._draw_masks(B, 1, dev)
        with torch.no_grad():
            return eng.encode_condition(cond, masks)


class _SlotName(str):
    """Engine layout name plus step index; GlowEngine.view() resolves it."""

    def __new__(cls, name, k):
        obj = super().__new__(cls, name)
        obj.k = k
        return obj
