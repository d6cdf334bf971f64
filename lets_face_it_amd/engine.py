"""Host-side orchestration of the HIP kernels for one SeqGlow instance.

PyTorch is plumbing here: device memory (torch.empty), the current HIP stream and
torch.distributed. All arithmetic of the hot path happens in liblfi_hip.so; this
module only sequences the C-ABI calls of include/lfi.h:

  forward   encoders (input-projection GEMM + per-step GEMM/gate kernels) -> feature matrix `cond`
            -> c = LeakyReLU(cond Wct^T + b) for all flow steps in ONE GEMM -> gic = c W_ih[:, Ch:]^T + b_ih
            (batched GEMM) -> anti-diagonal walk of the (timestep, flow step) grid -> per-frame NLL
  backward  reverse walk -> deferred weight-gradient GEMMs -> cond_transform / encoder BPTT -> flat gradient buffer
  step      global-norm clip + Adam on the flat parameter buffer (one kernel each)

What the reference does instead: SeqGlow.forward's Python loop over timesteps and flow steps
(/root/reference/code/glow_pytorch/glow/models.py:534-561), ~5.4k ATen calls per timestep.
"""
import atexit
import ctypes as C
import math
import contextlib
import os
import warnings
import weakref

import torch

from . import _lib
from ._lib import EncDesc, FlowDims, FlowGrads, FlowParams, GemmDesc, P1Enc, PGemmDesc, check, ptr, translate_oom

ENC_ORDER = ("p1_face", "p2_face", "p1_speech", "p2_speech")  # FeatureEncoder concat order (models.py:127-143)
FLOW_FIELDS = ("an_bias", "an_logs", "inv_l", "inv_u", "inv_logs", "inv_w", "w_ih", "w_hh", "b_ih", "b_hh",
               "wct", "bct", "w_fl", "b_fl", "l_fl")
ENC_LEAVES = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
# GEMM classes of the backward pass (tools/precision_sweep.py): they leave the NLL untouched
BWD_CLASSES = ("dpre", "cond_wgrad", "cond_dgrad", "flow_pgrads", "enc_dwih", "enc_dwhh", "enc_bptt")   # enc_bptt: the d gates x W_hh
# recurrence of the window encoders' BPTT (inside lfi_encode_windows_bwd)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class EncoderSpec:
    """One modality of the FeatureEncoder (models.py:83-125)."""

    def __init__(self, name, enc, in_dim, hist, hid, dropout):
        self.name, self.enc, self.in_dim, self.hist, self.hid, self.dropout = name, enc, in_dim, hist, hid, dropout
        self.ng = 4 if enc == "lstm" else 3   # gate blocks of the recurrent window encoders (nn.LSTM: i, f, g, o)
        if enc in ("rnn", "lstm"):       # nn.GRU / nn.LSTM: cat(seq[:, -1], h_n[0]), the same vector twice (models.py:60-69)
            self.dim = 2 * hid
        elif enc == "mlp":
            self.dim = hid
        elif enc == "none":
            self.dim = in_dim * hist
        elif enc == "frame_nb":  # FeatureEncoder's frame-counter column (models.py:116-117,143-144), not a ModalityEncoder
            self.dim = 1
        else:
            raise NotImplementedError(
                "encoder type %r for %s: 'rnn' (GRU), 'lstm', 'mlp' and 'none' have HIP kernels in this build ('cnn' has a "
                "broken output size in the reference, models.py:48)" % (enc, name))
        self.col = 0
        self.win = (in_dim * hist + 3) // 4 * 4  # row stride of the flattened-window matrix of an mlp encoder


class ModelSpec:
    """Static shape description derived from the hparams namespace."""

    def __init__(self, hparams):
        cond, glow, data = hparams.Conditioning, hparams.Glow, hparams.Data
        self.C = int(cond["p1_face"]["dim"])
        self.S = int(data["speech_dim"])
        self.D = int(cond["cond_dim"])
        self.H = int(glow["hidden_channels"])
        self.Ks = int(glow["K"]) * int(glow["L"])
        self.affine = 1 if glow["flow_coupling"] == "affine" else 0
        self.rnn_type = glow.get("rnn_type") or "gru"
        self.lu = bool(glow["LU_decomposed"])
        self.scale_eps = float(glow["scale_eps"])
        self.actnorm_scale = float(glow["actnorm_scale"])
        if glow["flow_permutation"] != "invconv":
            raise NotImplementedError("flow_permutation %r: only 'invconv' works in the reference too "
                                      "(Permute2d is broken, modules.py:98-118)" % glow["flow_permutation"])
        self.use_frame_nb = bool(cond.get("use_frame_nb"))
        self.start = max(cond[m]["history"] for m in ENC_ORDER)
        self.encoders = []
        col = 0
        for m in ENC_ORDER:
            cfg = cond[m]
            hist = int(cfg["history"])
            if m != "p1_face" and not hist:
                continue
            in_dim = int(cfg["dim"]) if m.endswith("face") else self.S
            e = EncoderSpec(m, cfg["enc"], in_dim, hist, int(cfg["hidden_dim"] or 0), float(cfg["dropout"] or 0))
            e.col = col
            col += e.dim
            self.encoders.append(e)
        if self.use_frame_nb:  # last column of the feature vector (models.py:143-144)
            e = EncoderSpec("frame_nb", "frame_nb", 1, 1, 0, 0.0)
            e.col = col
            col += 1
            self.encoders.append(e)
        self.E = col
        self.Ch = self.C // 2
        self.C2 = self.C - self.Ch
        self.Cout = 2 * self.C2 if self.affine else self.C2
        self.G = (4 if self.rnn_type == "lstm" else 3) * self.H
        self.I = self.Ch + self.D
        p1 = self.encoders[0]
        self.p1_cols = p1.dim  # width of the autoregressive (prev_p1_face) block, first in the feature vector
        # folded feature layout: a GRU encoder's output appears twice in the reference's vector (cat(seq[:, -1], h_n[0]),
        # models.py:63-64); the engine stores it once and adds the two weight column blocks instead (lfi_cols_fold)
        # every block starts on a 4-float boundary so that column blocks of the folded matrices stay 16-byte aligned;
        # the padding columns are zero in both the features and the folded weights
        fcol, self.fold_a, self.fold_b, self.unfold = 0, [], [], []
        for e in self.encoders:
            while fcol % 4:
                self.fold_a.append(-1)
                self.fold_b.append(-1)
                fcol += 1
            e.fcol = fcol
            e.fdim = e.hid if e.enc in ("rnn", "lstm") else e.dim
            for j in range(e.fdim):
                self.fold_a.append(e.col + j)
                self.fold_b.append(e.col + e.hid + j if e.enc in ("rnn", "lstm") else -1)
            self.unfold += [fcol + (j % e.fdim) for j in range(e.dim)]
            fcol += e.fdim
        self.Ef = fcol
        # row stride of the folded feature matrix / weights: whole PAIRS of 16-column k-tiles (the planes products on
        # v_mfma_f32_16x16x32_bf16 take k-tiles two at a time; the padding columns are zeros)
        self.ldf = (fcol + 31) // 32 * 32

    @classmethod
    def flow_only(cls, C, H, D, Ks, E, affine, rnn_type, lu, scale_eps, actnorm_scale):
        """Shapes of a stand-alone FlowNet / FlowStep (the module-level API the reference's test_modules.py exercises):
        no window encoders, the feature vector is whatever the caller passes as `condition` (B x E)."""
        s = object.__new__(cls)
        s.C, s.S, s.D, s.H, s.Ks, s.E = int(C), 0, int(D), int(H), int(Ks), int(E)
        s.affine, s.rnn_type, s.lu = int(bool(affine)), rnn_type or "gru", bool(lu)
        s.scale_eps, s.actnorm_scale = float(scale_eps), float(actnorm_scale)
        s.use_frame_nb, s.start, s.encoders, s.p1_cols = False, 0, [], 0
        s.Ch = s.C // 2
        s.C2 = s.C - s.Ch
        s.Cout = 2 * s.C2 if s.affine else s.C2
        s.G = (4 if s.rnn_type == "lstm" else 3) * s.H
        s.I = s.Ch + s.D
        s.fold_a, s.fold_b, s.unfold = list(range(s.E)), [-1] * s.E, list(range(s.E))
        s.Ef = s.E
        s.ldf = (s.E + 31) // 32 * 32
        return s

    def flow_shapes(self):
        Ks, C, H, D, E, G, I, Cout = self.Ks, self.C, self.H, self.D, self.E, self.G, self.I, self.Cout
        shapes = {
            "an_bias": (Ks, C), "an_logs": (Ks, C),
            "w_ih": (Ks, G, I), "w_hh": (Ks, G, H), "b_ih": (Ks, G), "b_hh": (Ks, G),
            "wct": (Ks, D, E), "bct": (Ks, D),
            "w_fl": (Ks, Cout, H), "b_fl": (Ks, Cout), "l_fl": (Ks, Cout),
        }
        if self.lu:
            shapes.update({"inv_l": (Ks, C, C), "inv_u": (Ks, C, C), "inv_logs": (Ks, C)})
        else:
            shapes["inv_w"] = (Ks, C, C)
        return shapes


class _Ctx:
    pass


class GlowEngine:
    """Owns the flat parameter / gradient / optimiser buffers and the workspaces of one model on one GPU."""

    @translate_oom
    def __init__(self, spec, device):
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LfiError("GlowEngine needs a GPU device (got %s); there is no CPU path" % device)
        self.L = _lib.lib()
        self.layout = {}  # name -> (offset, shape)
        off = 0
        for e in spec.encoders:
            if e.enc in ("rnn", "lstm"):
                g = e.ng * e.hid
                for leaf, shape in zip(ENC_LEAVES, ((g, e.in_dim), (g, e.hid), (g,), (g,))):
                    self.layout["enc.%s.%s" % (e.name, leaf)] = (off, shape)
                    off += math.prod(shape)
            elif e.enc == "mlp":
                for leaf, shape in (("mlp_weight", (e.hid, e.in_dim * e.hist)), ("mlp_bias", (e.hid,))):
                    off = (off + 3) // 4 * 4  # 16-byte aligned: GEMM operand
                    self.layout["enc.%s.%s" % (e.name, leaf)] = (off, shape)
                    off += math.prod(shape)
        shapes = spec.flow_shapes()
        for name in FLOW_FIELDS:
            if name in shapes:
                self.layout["flow." + name] = (off, shapes[name])
                off += math.prod(shapes[name])
        self.n_params = off
        # first float of the flow block of the flat buffers (the encoder leaves come first): the data-parallel trainer
        # all-reduces [flow_offset:) while the encoder BPTT still runs
        self.flow_offset = min(o for n, (o, _) in self.layout.items() if n.startswith("flow."))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(off, **f32)
        self.grads = torch.zeros(off, **f32)
        self.adam_m = None
        self.adam_v = None
        self.step_count = 0
        self.inv_p = torch.zeros(spec.Ks, spec.C, spec.C, **f32)
        self.inv_sign = torch.zeros(spec.Ks, spec.C, **f32)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.sumsq_work = torch.zeros(1024, dtype=torch.float64, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.fold_a = torch.tensor(spec.fold_a, **i32)
        self.fold_b = torch.tensor(spec.fold_b, **i32)
        self.unfold = torch.tensor(spec.unfold, **i32)
        self.wct_f = torch.zeros(spec.Ks * spec.D, spec.ldf, **f32)  # folded cond_transform weights, rebuilt by run_prep
        self._side_stream = None
        self._partial_streams = {}
        self._sample_stream = None
        self._partial_refused = False
        self._ws = {}
        self._sample_graphs = {}
        self._sample_seen = {}
        self.prep = None
        self._last = None
        self.timers = None  # {tag: [(start_event, end_event), ...]} when kernel timing is switched on (bench.py)
        self._timer_tags = None
        # GEMM arithmetic: 0 = exact fp32 on the f32-input MFMA; 1 = bf16x3 (fp32 operands split into bf16 hi + lo on the
        # fly, three bf16 MFMAs per step, fp32 accumulation: ~2^-16 relative, 16x the MFMA rate)
        self.precision = 0
        # arithmetic of the AUTOREGRESSIVE part of sample() (per generated frame: the window columns of cond_transform, gic and
        # the Ks reverse cells). At full depth (K = 16 x 56 generated frames) plain fp32 torch is itself 2.4e-5 from the fp64
        # oracle, and every frame's error feeds all later frames through the prev_p1_face window: this part has to be
        # fp32-grade whatever the static part (window encoders, non-autoregressive cond_transform columns) uses - with
        # three bf16 products here the sampler lands at 4.7e-5 - 5.6e-5 (gate: 1.5 x the fp32 floor). In bf16x3 engine mode it
        # is, at three-product cost (value 9, the default; profiles/round3_sample_precision.md): the per-frame GEMMs
        # (lfi_gemm_desc.precision 9) and the reverse cells' recurrent products as three FP16 products of two-piece operands
        # (11 + 11 mantissa bits, 2^-22 relative; their operands - h, standardised faces, flow activations, features, weights -
        # sit far inside fp16's range): 2.2e-5 at 94 ms per 1024 x 300 call, against 122 ms with everything
        # on the f32-input MFMA (value 0, LFI_SAMPLE_FRAME_PRECISION=f32), 99 ms with six bf16 products in the GEMMs (value 5,
        # =bf16x6: no range caveat) and 93 ms with three bf16 products (value 1, =bf16x3: 4.7e-5). None = by engine mode (9 / 0).
        self.sample_frame_precision = {"bf16x3": 1, "f32": 0, "bf16x6": 5, "fp16x3": 9}.get(os.environ.get("LFI_SAMPLE_FRAME_PRECISION", ""))
        self._mask_calls = 0
        self._enc_stash_f16 = {}   # modality -> did the last stashing forward write its gate stash as fp16 (build_features)
        self._sample_fp16_unsafe = False   # set once a sampling call in fp16 pieces came back non-finite (_check_last_sample)
        # GEMM class -> bf16x3 products to drop (bit 0: a_lo b_hi, bit 1: a_hi b_lo; 3 = plain bf16 operands, one product).
        # Measurement switch only: profiles/precision_sweep.md (tools/precision_sweep.py, final widths, against the fp64
        # oracle) shows NO class that keeps the test bounds with fewer than three products - the forward classes cost
        # 1e-5 ... 4e-5 of the 1e-4 NLL gate EACH, the backward classes leave the NLL alone but put single gradient tensors at
        # 1.3e-3 ... 3.3e-3 relative L2 (gate 2e-3; the encoders' dW_hh, the mildest, still 1.9e-3 on its smallest tensor) -
        # so the default is empty: three products everywhere. LFI_PASS_SKIP="cls=bits,..." sets entries for timing runs.
        # bf16 products per k-step in the BACKWARD GEMM classes (BWD_CLASSES): 3, 2 (the A operand - always a gradient: dgi,
        # dpre, dgh, scattered dgi - rounded to bf16, i.e. the a_lo b_hi product not issued) or "auto" = 2 once a step holds
        # >= 8192 frames, else 3. Evidence (profiles/precision_sweep_b256.md, profiles/round3_loss_curve_ab.md): at the benchmark's
        # 14 336 frames every backward class with two products keeps the per-tensor gradient gate (worst tensor 8.7e-4 of 2e-3
        # relative L2 against the fp64 oracle; three products: 4.7e-4), the per-frame NLL is untouched (forward classes keep three),
        # and a 200-step loss curve is indistinguishable from the three-product one (parameter distance 8.5e-6 against 1.4e-3 for
        # another dropout seed). At a few hundred frames single small tensors exceed the gate (1.3e-3 .. 1.9e-3 per class), hence
        # the threshold. hparams `engine_backward_products` / LFI_BWD_PRODUCTS override.
        self.backward_products = self.check_backward_products(os.environ.get("LFI_BWD_PRODUCTS", "auto"))
        self._bwd_skip = 0   # skip bits of the backward classes for the backward pass in progress (set by backward())
        self.tile_pin = {k.strip(): int(v) for k, v in (it.split("=") for it in os.environ.get("LFI_TILE_PIN", "").split(",") if it)}
        self.pass_skip = {}
        env = os.environ.get("LFI_PASS_SKIP", "")
        if env.strip() == "none":
            self.pass_skip = {}
        for item in filter(None, [] if env.strip() == "none" else env.split(",")):
            name, bits = item.split("=")
            self.pass_skip[name.strip()] = int(bits)

    def _skip_bits(self, cls):
        """bf16x3 products to drop for GEMM class `cls` (bit 0: a_lo b_hi, bit 1: a_hi b_lo): an explicit pass_skip entry
        (measurement runs), else the backward pass's product count for the backward classes, else none."""
        if cls in self.pass_skip:
            return self.pass_skip[cls] & 3
        return self._bwd_skip if cls in BWD_CLASSES else 0

    @staticmethod
    def check_backward_products(value):
        """2, 3 or "auto" (LFI_BWD_PRODUCTS / hparams `engine_backward_products`); anything else - e.g. 1, which no kernel
        implements and which used to run silently as 3 - raises."""
        if isinstance(value, str) and value.strip().lower() == "auto":
            return "auto"
        try:
            n = int(value)
        except (TypeError, ValueError):
            n = None
        if n not in (2, 3) or (not isinstance(value, str) and n != value):
            raise ValueError("backward products must be 2, 3 or 'auto', got %r" % (value,))
        return n

    def _want_stash_f16(self, frames):
        """The window encoders' gate stash as fp16? LFI_ENC_STASH_F16=0 / 1 forces; by default wherever the backward GEMM classes
        take two products (backward_product_count)."""
        env = os.environ.get("LFI_ENC_STASH_F16")
        if env in ("0", "1"):
            return env == "1" and bool(self.precision & 1)
        return self.backward_product_count(frames) == 2

    def backward_product_count(self, frames):
        bp = self.check_backward_products(self.backward_products)
        if not (self.precision & 1):
            return 3
        return (2 if frames >= 8192 else 3) if bp == "auto" else int(bp)

    # ------------------------------------------------------------------ per-kernel timing (HIP events on the launch stream)
    def enable_timing(self, on=True, only=None):
        """Per-kernel HIP events around the tagged launches (bench.py). only: an iterable of tags to restrict them to - every
        event record is a marker packet in the stream's queue and costs the step ~6 us of dispatch gap (18 of them per step
        with every tag on: 0.1 ms of an 8 ms step)."""
        self.timers = {} if on else None
        self._timer_tags = None if only is None else set(only)

    def _tic(self, tag):
        if self.timers is None or tag is None or (self._timer_tags is not None and tag not in self._timer_tags):
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def _toc(self, tag, ev):
        if ev is not None:
            ev[1].record()
            self.timers.setdefault(tag, []).append(ev)

    def timing_summary(self):
        """{tag: (launches, mean milliseconds)}; synchronises."""
        torch.cuda.synchronize()
        return {t: (len(evs), sum(a.elapsed_time(b) for a, b in evs) / len(evs)) for t, evs in (self.timers or {}).items()}

    # ------------------------------------------------------------------ views
    def view(self, name, buf=None):
        off, shape = self.layout[name]
        buf = self.params if buf is None else buf
        return buf[off:off + math.prod(shape)].view(shape)

    def fview(self, name, buf=None):
        return self.view("flow." + name, buf)

    def _flow_dims(self, B, N):
        s = self.spec
        return FlowDims(B, N, s.C, s.H, s.D, s.Ks, s.affine, 1 if s.rnn_type == "lstm" else 0, s.scale_eps, self.precision)

    def _flow_params(self):
        s = self.spec
        p = FlowParams()
        for name in ("an_bias", "an_logs", "w_ih", "w_hh", "b_ih", "b_hh", "w_fl", "b_fl", "l_fl"):
            setattr(p, name, self.fview(name).data_ptr())
        if s.lu:
            p.inv_l, p.inv_u, p.inv_logs = (self.fview(n).data_ptr() for n in ("inv_l", "inv_u", "inv_logs"))
            p.inv_p, p.inv_sign = self.inv_p.data_ptr(), self.inv_sign.data_ptr()
        else:
            p.inv_w = self.fview("inv_w").data_ptr()
        return p

    def _flow_grads(self):
        s = self.spec
        g = FlowGrads()
        for name in ("an_bias", "an_logs", "w_ih", "w_hh", "b_ih", "b_hh", "w_fl", "b_fl", "l_fl"):
            setattr(g, name, self.fview(name, self.grads).data_ptr())
        if s.lu:
            g.inv_l, g.inv_u, g.inv_logs = (self.fview(n, self.grads).data_ptr() for n in ("inv_l", "inv_u", "inv_logs"))
        else:
            g.inv_w = self.fview("inv_w", self.grads).data_ptr()
        return g

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, floats, zero=False):
        t = self._ws.get(name)
        if t is None or t.numel() < floats:
            # always born zeroed: padding columns of the folded feature matrix are never written and must stay zero
            t = torch.zeros(max(int(floats), 1), dtype=torch.float32, device=self.device)
            self._ws[name] = t
            self._order_birth()
        elif zero:
            t.zero_()
        return t

    def _order_birth(self):
        """A buffer's zero fill runs on the stream that is current when the buffer is born; work that was FORKED earlier onto the second
        stream is not ordered behind it. Round 6 found what that does (tools/dp_gloo_check.py, DP_CHECK_EAGER_TWICE): in the first
        forward pass of a new engine the feature matrix `cond` is born on the main stream after the fork, the smallest window encoder
        (and, since this round, the prev_p1_face gather) writes its columns of it on the second stream - and when the fill happened to
        run late (two ranks sharing one card: reliably; a card of one's own: never observed) it wiped them: ActNorm's data-dependent
        init, which is exactly that first pass, then saw zero features, and the run continued from other parameters - identical on both
        ranks, different from run to run. So: whoever is forked waits for the birth."""
        if os.environ.get("LFI_NO_BIRTH_ORDER") == "1":      # test hook: the behaviour before the fix (tests/test_gpu_parity.py)
            return
        if torch.cuda.is_current_stream_capturing():         # (inside a capture the fill is a node of the graph; a stream outside the
            return                                           # capture must not be made to wait on it)
        cur = torch.cuda.current_stream(self.device)
        for st in [self._side_stream] + list((self.__dict__.get("_partial_streams") or {}).values()):
            if st is not None and st != cur:
                st.wait_stream(cur)

    def draw_masks(self, B, N, seed, key_dev=None):
        """{modality: (N, B, hist)} dropout multipliers of the window encoders in ONE launch (lfi_dropout_masks); the call
        counter makes successive calls independent, (seed, counter) reproduces a call. key_dev: a device (seed, offset) pair the
        kernel reads instead (a captured training step: the caller sets it before every replay and keeps the counter)."""
        segs = [e for e in self.spec.encoders if e.dropout > 0]
        if not segs:
            return None
        if key_dev is None:
            self._mask_calls += 1
        total = sum(N * B * e.hist for e in segs)
        flat = self._buf("dropout_masks", total)
        outs, ns, keeps, masks, off = (C.c_void_p * 4)(), (C.c_long * 4)(), (C.c_float * 4)(), {}, 0
        for i, e in enumerate(segs):
            n = N * B * e.hist
            view = flat[off:off + n].view(N, B, e.hist)
            outs[i], ns[i], keeps[i] = view.data_ptr(), n, 1.0 - e.dropout
            masks[e.name] = view
            off += n
        if key_dev is not None:
            check(self.L.lfi_dropout_masks_dev(len(segs), outs, ns, keeps, key_dev.data_ptr(), _stream()), "lfi_dropout_masks_dev")
        else:
            check(self.L.lfi_dropout_masks(len(segs), outs, ns, keeps, int(seed) & (2 ** 64 - 1), self._mask_calls, _stream()),
                  "lfi_dropout_masks")
        return masks

    def has_dropout(self):
        return any(e.dropout > 0 for e in self.spec.encoders)

    # ------------------------------------------------------------------ second stream
    # Launch-latency-bound and HBM-streaming side work (parameter preparation; the encoders' window scatter + dW_ih) runs on a
    # second stream next to the MFMA-bound products it does not depend on (same-box A/B: -0.05 .. -0.14 ms per step).
    # _fork(): the side stream, ordered after everything enqueued on the current stream so far (None with LFI_NO_OVERLAP=1:
    # everything in line); _join(): the current stream waits for it. Side work gets its own scratch buffers (ws=...).
    def _fork(self):
        if os.environ.get("LFI_NO_OVERLAP") == "1":
            return None
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        self._side_stream.wait_stream(torch.cuda.current_stream(self.device))
        return self._side_stream

    def close(self):
        """Destroys the partial-chip streams this engine created (after their work has finished). Called when the module re-binds
        to a new engine (.to() / .float(): glow/models.py _bind), at interpreter exit, and from __del__."""
        streams, self._partial_streams = self.__dict__.get("_partial_streams") or {}, {}
        for st in streams.values():
            st.synchronize()
            self.L.lfi_stream_destroy(C.c_void_p(st.cuda_stream))

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown: the runtime may be gone already
            pass

    def _fork_partial(self, cus_per_xcd):
        """A second stream that owns `cus_per_xcd` CUs of every XCD (lfi_stream_create_partial), ordered after the current stream
        like _fork()'s; the caller joins it through events (sample())."""
        st = self._partial_streams.get(cus_per_xcd)
        if st is None:
            ptr = C.c_void_p()
            if self.L.lfi_stream_create_partial(int(cus_per_xcd), C.byref(ptr)) != 0:
                # a driver that refuses CU masks: the ordinary second stream does the same work, in turns with the chain
                if not self._partial_refused:
                    warnings.warn("lets_face_it_amd: no partial-chip stream (%s); the sampler's static part runs on an ordinary "
                                  "second stream" % self.L.lfi_last_error().decode(), RuntimeWarning)
                self._partial_refused = True
                return self._fork()
            st = self._partial_streams[cus_per_xcd] = torch.cuda.ExternalStream(ptr.value, device=self.device)
            if len(self._partial_streams) == 1:
                ref = weakref.ref(self)
                atexit.register(lambda: ref() is not None and ref().close())
        st.wait_stream(torch.cuda.current_stream(self.device))
        return st

    @staticmethod
    def _on(side):
        return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

    def _join(self):
        if self._side_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side_stream)

    # ------------------------------------------------------------------ low-level wrappers
    def gemm(self, M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, ldc, bias=None, act=0, slope=0.01, G=None, ldg=0,
             batch=1, sA=0, sB=0, sC=0, sBias=0, sG=0, accumulate=0, splitk=1, a_off=0, b_off=0, c_off=0, bias_off=0,
             tag=None, ws="scratch.gemm_splitk", cls=None, colsum_into=None, a_bf16=False):
        """Offsets are in floats relative to the tensors' data pointers. cls: GEMM class name for tools/precision_sweep.py
        (self.pass_skip maps a class to the bf16x3 products to drop; empty in normal operation).
        colsum_into: optional (ncols,) tensor that receives the column sums of the stored result (all batch entries side by
        side, as they lie in C): taken from per-tile partial sums the GEMM epilogue leaves, without a second pass over C.
        Returns False when this product cannot do that (the caller then sums C itself)."""
        g = GemmDesc()
        g.M, g.N, g.K = M, N, K
        g.A, g.lda, g.a_kcontig = A.data_ptr() + (2 if a_bf16 else 4) * a_off, lda, akc
        g.a_bf16 = 1 if a_bf16 else 0
        g.B, g.ldb, g.b_kcontig = Bm.data_ptr() + 4 * b_off, ldb, bkc
        g.C, g.ldc = Cm.data_ptr() + 4 * c_off, ldc
        g.bias = None if bias is None else bias.data_ptr() + 4 * bias_off
        g.G, g.ldg = ptr(G), ldg
        g.batch, g.strideA, g.strideB, g.strideC, g.strideBias, g.strideG = batch, sA, sB, sC, sBias, sG
        g.accumulate, g.act, g.slope = accumulate, act, slope
        g.splitk = splitk
        g.precision = self.precision | (self._skip_bits(cls) << 8 if (self.precision & 1) else 0)
        if self.tile_pin and (self.precision & 1):   # LFI_TILE_PIN="cls=128|256,...": pin a GEMM class's tile shape (measurement)
            g.precision |= {128: 0x20, 256: 0x10}.get(self.tile_pin.get(cls), 0)
        if splitk > 1 or splitk == 0:  # 0: the library picks the split (and the tile shape) that fills the chip
            g.work = self._buf(ws, batch * max(splitk, 8) * M * N).data_ptr()
        part, prow = None, 0
        if colsum_into is not None and os.environ.get("LFI_NO_COLPART") != "1":
            prow = int(self.L.lfi_gemm_colpart_rows(C.byref(g)))
            if prow > 0:
                part = self._buf("scratch.colpart", prow * ldc)
                g.colsum_part, g.ld_part = part.data_ptr(), ldc
        ev = self._tic(tag)
        check(self.L.lfi_gemm_f32(C.byref(g), _stream()), "lfi_gemm_f32")
        self._toc(tag, ev)
        if part is not None:
            self.colsum(part, ldc, 0, prow, colsum_into.numel(), 1, colsum_into, 0, ws="scratch.colsum.part")
        return part is not None

    # ---- bf16 hi / lo operand planes in MFMA fragment order (include/lfi.h, lfi_planes_from_f32 / lfi_gemm_planes)
    def planes(self, name, X, ldx, rows, cols, x_off=0):
        """Split the fp32 matrix X (rows x cols, row pitch ldx) once into planes kept in workspace `name`. -> (tensor, nkt)"""
        # + one 256-row panel of slack: a batch entry that starts inside the buffer (lfi_pgemm_desc.a_stride / b_stride) reads
        # whole 256-row panels from its own first row (rows past its M / N only feed outputs that are never stored)
        elems = self.L.lfi_planes_elems(rows, cols) + 256 * ((cols + 15) // 16 * 16) * 2
        buf = self._ws.get(name)
        if buf is None or buf.numel() < elems:
            buf = torch.empty(elems, dtype=torch.bfloat16, device=self.device)
            self._ws[name] = buf
        check(self.L.lfi_planes_from_f32(X.data_ptr() + 4 * x_off, ldx, rows, cols, buf.data_ptr(), _stream()),
              "lfi_planes_from_f32")
        return buf, (cols + 15) // 16

    def plane_buf(self, name, elems):
        """An (uninitialised) bf16 workspace for planes that a kernel's epilogue writes."""
        buf = self._ws.get(name)
        if buf is None or buf.numel() < elems:
            buf = torch.empty(int(elems), dtype=torch.bfloat16, device=self.device)
            self._ws[name] = buf
        return buf

    def gemm_planes(self, M, N, K, Ap, a_nkt, Bp, b_nkt, Cm, ldc, bias=None, act=0, slope=0.01, G=None, ldg=0, batch=1,
                    a_stride=0, b_stride=0, sC=0, sBias=0, sG=0, accumulate=0, c_off=0, bias_off=0, tag=None, cls=None,
                    a_fmt=0, b_fmt=0, a_off=0, b_off=0, splitk=1, ws="scratch.pgemm_splitk", store=True,
                    Cr=None, cr_nkt=0, cr_col0=0, Gr=None, gr_nkt=0, gr_col0=0, colsum_into=None, hi_only=False, tile=0):
        """lfi_gemm_planes. a_fmt / b_fmt: 0 = row use of the operand's planes (k = the matrix' columns), 1 = transposed use (k = its
        rows). a_nkt / b_nkt: column tiles per row tile of the plane buffers. a_off / b_off: bf16 elements into the plane buffers (a
        column-tile offset: ct * 1024; a row-tile offset: rt * nkt * 1024). Cr: bf16 tensor that receives the result as planes; Gr:
        planes whose hi plane's sign stands in for G (act 2). colsum_into: as gemm(). Returns True when the column sums were taken in
        the epilogue."""
        g = PGemmDesc()
        g.skip = self._skip_bits(cls)
        g.M, g.N, g.K = M, N, K
        g.Ap, g.a_nkt, g.a_stride = Ap.data_ptr() + 2 * a_off, a_nkt, a_stride
        g.Bp, g.b_nkt, g.b_stride = Bp.data_ptr() + 2 * b_off, b_nkt, b_stride
        g.C, g.ldc = (None if Cm is None else Cm.data_ptr() + 4 * c_off), ldc
        g.bias = None if bias is None else bias.data_ptr() + 4 * bias_off
        g.G, g.ldg = ptr(G), ldg
        g.batch, g.strideC, g.strideBias, g.strideG = batch, sC, sBias, sG
        g.accumulate, g.act, g.slope = accumulate, act, slope
        g.a_fmt, g.b_fmt = a_fmt, b_fmt
        g.splitk = splitk
        g.store_f32 = 1 if store else 0
        if splitk > 1:
            g.work = self._buf(ws, batch * splitk * M * N).data_ptr()
        g.Cr, g.cr_nkt, g.cr_col0 = ptr(Cr), cr_nkt, cr_col0
        g.Gr, g.gr_nkt, g.gr_col0 = ptr(Gr), gr_nkt, gr_col0
        g.out_hi_only, g.tile = (1 if hi_only else 0), tile
        part, prow = None, 0
        if colsum_into is not None and os.environ.get("LFI_NO_COLPART") != "1":
            prow = int(self.L.lfi_gemm_planes_colpart_rows(C.byref(g)))
            if prow > 0:
                part = self._buf("scratch.colpart", prow * ldc)
                g.colsum_part, g.ld_part = part.data_ptr(), ldc
        ev = self._tic(tag)
        check(self.L.lfi_gemm_planes(C.byref(g), _stream()), "lfi_gemm_planes")
        self._toc(tag, ev)
        if part is not None:
            self.colsum(part, ldc, 0, prow, colsum_into.numel(), 1, colsum_into, 0, ws="scratch.colsum.part")
        return part is not None

    @staticmethod
    def _fill_splitk(M, N, K, batch=1):
        """Split K so that a long-K product with few output tiles still gives every CU ~3 workgroups (256 CUs)."""
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        return max(1, min(16, 768 // max(tiles, 1), K // 1024))

    @staticmethod
    def _long_k_splitk(M, N, K, kmin=1024):
        """K split of a long-K product with a handful of output tiles (the BPTT weight gradients: K = steps x windows): as many
        splits as fill one round of co-resident workgroups of the tile shape the library's plan will pick (256 x 256: one per CU,
        rate 1.12; 128 x 128: two per CU), each keeping at least kmin of K."""
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        wide = 1.12 * M * N / (t256 * 65536.0) >= M * N / (t128 * 16384.0)
        sk = 256 // t256 if wide else 512 // t128
        return max(1, min(sk, K // kmin))

    @staticmethod
    def _tail_splitk(M, N, K, batch=1, slots=512):
        """Smallest K split (<= 8) that wastes < 15 % of the last round of `slots` co-resident workgroups."""
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        best, best_eff = 1, 0.0
        for sk in (1, 2, 3, 4, 6, 8):
            if K // sk < 512:
                break
            wg = tiles * sk
            eff = wg / (math.ceil(wg / slots) * slots)
            if eff > best_eff + 0.02:
                best, best_eff = sk, eff
            if eff > 0.85:
                break
        return best

    def colsum(self, X, ldx, strideX, rows, cols, batch, out, strideOut, scale=1.0, accumulate=0, x_off=0, ws="scratch.colsum"):
        w = self._buf(ws, self.L.lfi_colsum_work_floats(rows, cols, batch))
        check(self.L.lfi_colsum_f32(X.data_ptr() + 4 * x_off, ldx, strideX, rows, cols, batch, out.data_ptr(), strideOut,
                                    scale, accumulate, w.data_ptr(), _stream()), "lfi_colsum_f32")

    # ---- the cond_transform / coupling-input products on operand planes end to end ("the chain", bf16x3 mode): every big
    # operand of the step's GEMMs is split into bf16 hi / lo ONCE, by the kernel that produces it - c by the cond_transform
    # product's epilogue, d pre-activation by the dpre product's, dgi by the backward walk - and never exists as fp32
    def _chain_fwd_ok(self):
        s = self.spec
        return (self.precision == 1 and os.environ.get("LFI_PGEMM", "1") != "0" and os.environ.get("LFI_PCHAIN", "1") != "0"
                and s.D % 32 == 0 and s.G % 32 == 0)

    def _chain_ok(self, dims, with_stash):
        if not self._chain_fwd_ok():
            return False
        if not with_stash:
            return True
        F = dims.B * dims.N
        return F % 32 == 0 and bool(self.L.lfi_flow_bwd_emits_planes(C.byref(dims)))

    @staticmethod
    def _planes_splitk(M, N, K, batch=1, kmin=1024, slots=512):
        """K split (<= 8) of a product on the 128 x 256-tile planes kernel that best fills whole rounds of its 512 co-resident
        workgroups, less 3 % per extra partial for the reduce pass."""
        tiles = ((M + 127) // 128) * ((N + 255) // 256) * batch
        best, best_score = 1, -1.0
        for sk in range(1, 9):
            if sk > 1 and K // sk < kmin:
                break
            wg = tiles * sk
            score = wg / (math.ceil(wg / slots) * slots) * (1.0 - 0.03 * (sk - 1))
            if score > best_score + 1e-9:
                best, best_score = sk, score
        return best

    def run_prep(self, with_inverse=False):
        """W = P L U (+ transposes, optional fp64 inverse) once per parameter state, not per call (modules.py:147-178)."""
        d = self._flow_dims(1, 1)
        need = self.L.lfi_flow_prep_floats(C.byref(d))
        if self.prep is None or self.prep.numel() < need:
            self.prep = torch.zeros(need, dtype=torch.float32, device=self.device)
        p = self._flow_params()
        s = self.spec
        # the two STREAMING kernels of this block (80 + 60 MB) first: on the second stream they then run in the step's first ~90 us,
        # beside the small launches in front of the window encoders, instead of under the largest encoder's recurrence, which they
        # slowed (round 5: cols_fold started 128 us into the step behind the latency-bound invconv kernels and took 300 us there)
        prep_first = os.environ.get("LFI_PREP_FOLD_FIRST", "1") == "0"
        if prep_first:
            check(self.L.lfi_flow_prep(C.byref(d), C.byref(p), self.prep.data_ptr(), 1 if with_inverse else 0, _stream()),
                  "lfi_flow_prep")
        check(self.L.lfi_cols_fold(self.fview("wct").data_ptr(), s.E, s.Ks * s.D, self.fold_a.data_ptr(),
                                   self.fold_b.data_ptr(), s.Ef, self.wct_f.data_ptr(), s.ldf, _stream()), "lfi_cols_fold")
        if self.precision == 1:
            # bf16 hi / lo planes of the folded weights for the cond_transform forward product: split once per parameter state
            # here instead of once per row tile (56 times) inside the GEMM
            # (over all ldf columns - the padding columns are zero - so that the k-tile count is even whatever the widths and the
            # products take the v_mfma_f32_16x16x32_bf16 kernels, which consume k-tiles in pairs)
            self._wct_planes = self.planes("wct_planes", self.wct_f, s.ldf, s.Ks * s.D, s.ldf)
        if not prep_first:
            check(self.L.lfi_flow_prep(C.byref(d), C.byref(p), self.prep.data_ptr(), 1 if with_inverse else 0, _stream()),
                  "lfi_flow_prep")
        if self.precision == 1:
            if self._chain_fwd_ok():
                # W_c = W_ih[:, Ch:] of every flow step, (Ks G x D): row use in gic = c W_c^T (sums over D), transposed use in
                # d pre-activation = dgi W_c (sums over the gate rows). (The folded cond_transform weights above serve the
                # feature-gradient product the same way: it sums over their rows, the Ks D output units.)
                self._wc_r = self.planes("wc_r", self.prep, s.D, s.Ks * s.G, s.D, x_off=self._wc_offset())

    # ------------------------------------------------------------------ conditioning
    def _check_input(self, x, name, B, Tmin, dim):
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3 and x.shape[0] == B
                and x.shape[1] >= Tmin and x.shape[2] == dim):
            raise ValueError("%s: expected contiguous float32 GPU tensor (B=%d, T>=%d, %d), got %s %s on %s"
                             % (name, B, Tmin, dim, tuple(x.shape), x.dtype, x.device))

    def build_features(self, data, faces, B, T, masks, cond, with_stash, skip_p1=False, sampling=False, windows=False, frame0=0,
                       side=None):
        """FeatureEncoder.forward for every timestep at once (models.py:127-145): fills cond (F x ldf, folded layout).

        windows=True: `data` holds ONE conditioning window per modality, (B, hist, dim) each, as create_conditioning
        cuts them (models.py:598-615), and `data["frame_nb"]` the counter itself: one timestep, N = 1."""
        s = self.spec
        N = 1 if windows else T - s.start
        F = N * B
        # side: a forked second stream (forward()): the SMALLEST recurrent encoder's whole chain (pads, x W_ih^T, the recurrence: ~60 us
        # of small launches at final_model.yaml's p1_speech) runs there, next to the large ones; the caller joins
        small = None
        if side is not None and os.environ.get("LFI_ENC_FWD_SMALL_ON_SIDE", "1") != "0":
            rec = sorted((e for e in s.encoders if e.enc in ("rnn", "lstm") and not (e.name == "p1_face" and skip_p1)),
                         key=lambda e: e.hist * e.hid * e.hid)
            small = rec[0] if len(rec) >= 3 else None
        # (round 6) pure window gathers (enc: none - at final_model.yaml the prev_p1_face history, 23 us) touch nothing the recurrent
        # encoders read: on the second stream too, instead of in front of the first recurrence (6.364 against 6.392 ms per step, same box;
        # LFI_ENC_GATHER_ON_SIDE=0: in line)
        gather_side = side is not None and os.environ.get("LFI_ENC_GATHER_ON_SIDE", "1") != "0"
        for e in s.encoders:
            with (self._on(side) if (e is small or (gather_side and e.enc == "none")) else contextlib.nullcontext()):
                self._build_feature(e, data, faces, B, T, N, F, masks, cond, with_stash, skip_p1, sampling, windows, frame0)

    def _build_feature(self, e, data, faces, B, T, N, F, masks, cond, with_stash, skip_p1, sampling, windows, frame0):
        s = self.spec
        st = _stream()
        p1 = e.name == "p1_face"
        if p1 and skip_p1:
            return
        if e.enc == "frame_nb":
            # forward / invert count from batch["frame_nb"] + 2 * start (models.py:539-542,623-625); inference from
            # ones (:572-575); +2 per timestep either way
            base = None
            if not sampling:
                base = data.get("frame_nb")
                if base is None:
                    raise KeyError("Conditioning.use_frame_nb is set but the batch has no 'frame_nb' entry")
                if not (base.is_cuda and base.dtype == torch.float32 and base.numel() == B):
                    raise ValueError("frame_nb: expected a float32 GPU tensor of shape (B, 1), got %s %s on %s"
                                     % (tuple(base.shape), base.dtype, base.device))
                base = base.contiguous()
            # (frame0: a run of sampled frames that starts frame0 frames into its sequence, see sample())
            check(self.L.lfi_fill_frame_nb(ptr(base), 2.0 * frame0 if sampling else (0.0 if windows else 2.0 * s.start), B, N,
                                           cond.data_ptr(), s.ldf, e.fcol, st), "lfi_fill_frame_nb")
            return
        # prev_p1_face is the window [t - hist, t) of the model's own output (models.py:601-603); every other modality
        # (t - hist, t] of its input stream (:607-610): same kernels, window end shifted by one frame
        x = faces if p1 else data.get(e.name)
        if x is None:
            raise KeyError("batch is missing modality %r" % e.name)
        self._check_input(x, e.name, B, e.hist if windows else T, e.in_dim)
        Tx = x.shape[1]
        incl = 0 if p1 else 1
        start = (e.hist - incl) if windows else s.start   # a lone window ends at its own last row
        if windows and Tx != e.hist:
            raise ValueError("%s: a conditioning window has %d frames, got %d" % (e.name, e.hist, Tx))
        mk = None if masks is None else masks.get(e.name)
        if mk is not None and not (tuple(mk.shape) == (N, B, e.hist) and mk.is_contiguous()
                                   and mk.dtype == torch.float32 and mk.is_cuda):
            raise ValueError("mask for %s must be a contiguous float32 GPU tensor (N, B, hist)" % e.name)
        if e.enc == "none":
            check(self.L.lfi_gather_windows(x.data_ptr(), B, Tx, e.in_dim, N, start, e.hist, incl, ptr(mk),
                                            cond.data_ptr(), s.ldf, e.fcol, st), "lfi_gather_windows")
            return
        hid = e.hid
        if e.enc == "mlp":  # Linear(hist * in -> hid) + LeakyReLU on the flattened (masked) window (models.py:70-71)
            win = self._buf("enc_win." + e.name, F * e.win, zero=False)
            check(self.L.lfi_gather_windows(x.data_ptr(), B, Tx, e.in_dim, N, start, e.hist, incl, ptr(mk),
                                            win.data_ptr(), e.win, 0, st), "lfi_gather_windows")
            self.gemm(F, hid, e.in_dim * e.hist, win, e.win, 1, self.view("enc.%s.mlp_weight" % e.name),
                      e.in_dim * e.hist, 1, cond, s.ldf, bias=self.view("enc.%s.mlp_bias" % e.name), act=1, slope=0.01,
                      c_off=e.fcol)
            return
        # input projection hoisted over the B*Tx distinct frames (no bias: it is added after the dropout mask)
        lstm = e.enc == "lstm"
        G = e.ng * hid
        xp = self._buf("xp." + e.name, B * Tx * G)
        xa, wa, ldi = self._aligned_input(e, x, B * Tx, G)
        self.gemm(B * Tx, G, e.in_dim, xa, ldi, 1, wa, ldi, 1, xp, G, cls="enc_xproj")
        # gate stash: r, z, n, W_hn h + b_hn (GRU; only kept for a backward pass) / i, f, g, o, c (LSTM: it is the cell state)
        d = EncDesc(B, Tx, N, start - 1 + incl, e.hist, hid, s.ldf, e.fcol, self.precision, 0, 1 if lstm else 0)
        # fp16 gate stash (r, z, n, W_hn h + b_hn as four halves per hidden unit): where the backward pass runs its two-product
        # arithmetic anyway (gate DERIVATIVES rounded to bf16) and the row-layout kernels take the shape
        s16 = bool(with_stash and not lstm and self._want_stash_f16(F) and self.L.lfi_encode_windows_stash_f16_ok(C.byref(d)))
        d.stash_f16 = 1 if s16 else 0
        self._enc_stash_f16[e.name] = s16
        gates = self._buf("enc_gates." + e.name, e.hist * F * (5 if lstm else (2 if s16 else 4)) * hid) if (with_stash or lstm) else None
        hseq = self._buf("enc_hseq." + e.name, e.hist * F * hid)
        work = self._buf("scratch.enc_fwd." + e.name, self.L.lfi_encode_windows_work_floats(C.byref(d)))
        ev = self._tic("enc_fwd." + e.name)
        check(self.L.lfi_encode_windows_fwd(
            C.byref(d), xp.data_ptr(), self.view("enc.%s.weight_hh" % e.name).data_ptr(),
            self.view("enc.%s.bias_ih" % e.name).data_ptr(), self.view("enc.%s.bias_hh" % e.name).data_ptr(),
            ptr(mk), cond.data_ptr(), ptr(gates), hseq.data_ptr(), work.data_ptr(), st), "lfi_encode_windows_fwd")
        self._toc("enc_fwd." + e.name, ev)   # (includes the small weight-fragment kernel in front of the recurrence)

    def _aligned_input(self, e, x, rows, G):
        """Input stream and W_ih of a recurrent window encoder with 16-byte aligned rows. BASELINE's 50-d faces / 27-d speech
        (and the corpus' 30-d speech) give row strides that are not multiples of 4 floats, which keeps the input projection
        and its weight gradient off the vector-load / bf16x3 GEMM paths (measured: 45 + 133 us per modality on the exact
        f32 kernel). Two small strided copies per call (rows x in_dim and G x in_dim) into zero-padded buffers fix that."""
        if e.in_dim % 4 == 0 or os.environ.get("LFI_NO_XPAD") == "1":
            return x, self.view("enc.%s.weight_ih" % e.name), e.in_dim
        ldi = (e.in_dim + 3) // 4 * 4
        xa = self._buf("xpad." + e.name, rows * ldi)
        check(self.L.lfi_pad_rows(x.data_ptr(), rows, e.in_dim, e.in_dim, xa.data_ptr(), ldi, _stream()), "lfi_pad_rows")
        wa = self._buf("wihpad." + e.name, G * ldi)
        check(self.L.lfi_pad_rows(self.view("enc.%s.weight_ih" % e.name).data_ptr(), G, e.in_dim, e.in_dim, wa.data_ptr(), ldi,
                                  _stream()), "lfi_pad_rows")
        return xa, wa, ldi

    def _project(self, cond, F, chain=False, with_stash=False):
        """c = LeakyReLU(cond Wct^T + b) for all Ks steps (one GEMM), gic = c W_ih[:, Ch:]^T + b_ih (batched GEMM).
        chain: c leaves the first product as operand planes only (gic and the backward mask use them by rows, dW_c transposed)
        -> (c planes, gic); otherwise (c as fp32, gic)."""
        s = self.spec
        KD = s.Ks * s.D
        if chain:
            cp, nkc = self.planes("cond_planes", cond, s.ldf, F, s.ldf)   # (ldf columns: an even k-tile count; the padding is zero)
            wp, nkw = self._wct_planes
            nkKD = KD // 16
            c_r = self.plane_buf("c_r", self.L.lfi_planes_elems(F, KD) + 256 * KD * 2)
            self.gemm_planes(F, KD, s.ldf, cp, nkc, wp, nkw, None, KD, bias=self.fview("bct"), act=1, slope=0.01, store=False,
                             Cr=c_r, cr_nkt=nkKD, tag="gemm_cond_fwd", cls="cond_fwd")
            gic = self._buf("gic", s.Ks * F * s.G)
            wr, nkwr = self._wc_r
            self.gemm_planes(F, s.G, s.D, c_r, nkKD, wr, nkwr, gic, s.G, bias=self.fview("b_ih"), batch=s.Ks,
                             a_stride=(s.D // 16) * 1024, b_stride=(s.G // 32) * nkwr * 1024, sC=F * s.G, sBias=s.G,
                             tag="gemm_gic", cls="gic")
            self._cond_planes = (cp, nkc)
            return c_r, gic
        cbuf = self._buf("c", F * KD)
        if self.precision == 1 and os.environ.get("LFI_PGEMM", "1") != "0":
            # operands pre-split into bf16 hi / lo planes in MFMA fragment order, streamed to LDS by LDS-DMA: same products
            # and accumulation order as the fp32-operand bf16x3 kernel (bit-identical results; LFI_PGEMM=0 keeps that one)
            cp, nkc = self.planes("cond_planes", cond, s.ldf, F, s.ldf)
            wp, nkw = self._wct_planes
            self.gemm_planes(F, KD, s.ldf, cp, nkc, wp, nkw, cbuf, KD, bias=self.fview("bct"), act=1, slope=0.01,
                             tag="gemm_cond_fwd", cls="cond_fwd")
        else:
            self.gemm(F, KD, s.Ef, cond, s.ldf, 1, self.wct_f, s.ldf, 1, cbuf, KD, bias=self.fview("bct"), act=1, slope=0.01,
                      tag="gemm_cond_fwd", cls="cond_fwd")
        gic = self._buf("gic", s.Ks * F * s.G)
        self.gemm(F, s.G, s.D, cbuf, KD, 1, self.prep, s.D, 1, gic, s.G, bias=self.fview("b_ih"),
                  batch=s.Ks, sA=s.D, sB=s.G * s.D, sC=F * s.G, sBias=s.G, b_off=self._wc_offset(), cls="gic")
        return cbuf, gic

    # ------------------------------------------------------------------ forward / backward
    @translate_oom
    def forward(self, batch, masks=None, with_stash=True, init_actnorm=None):
        """Teacher-forced NLL of a batch dict of (B, T, dim) tensors. Returns z (N, B, C), nll (N, B)."""
        s = self.spec
        x = batch["p1_face"]
        B, T = x.shape[0], x.shape[1]
        N = T - s.start
        if N <= 0:
            raise ValueError("sequence length %d does not exceed the longest history %d" % (T, s.start))
        F = N * B
        # W = P L U, its transposes / bf16 images and the folded cond_transform weights (~0.11 ms of small launches) depend on
        # the parameters only: on the second stream they run under the window encoders
        side = self._fork()
        with self._on(side):
            self.run_prep()
        cond = self._buf("cond", F * s.ldf)
        self.build_features(batch, x, B, T, masks, cond, with_stash, side=side)
        self._join()
        dims = self._flow_dims(B, N)
        chain = self._chain_ok(dims, with_stash)
        cbuf, gic = self._project(cond, F, chain, with_stash)
        if init_actnorm is not None:
            self._actnorm_init_walk(x, B, T, F, gic, dims, init_actnorm)
        stash = self._buf("flow_stash", self.L.lfi_flow_stash_floats(C.byref(dims)))
        z = torch.empty(N, B, s.C, dtype=torch.float32, device=self.device)
        nll = torch.empty(N, B, dtype=torch.float32, device=self.device)
        p = self._flow_params()
        check(self.L.lfi_flow_seq_fwd(C.byref(dims), C.byref(p), self.prep.data_ptr(), x.data_ptr(), T, s.start,
                                      gic.data_ptr(), stash.data_ptr(), z.data_ptr(), nll.data_ptr(), _stream()),
              "lfi_flow_seq_fwd")
        ctx = _Ctx()
        ctx.batch, ctx.masks, ctx.B, ctx.T, ctx.N, ctx.F = batch, masks, B, T, N, F
        ctx.cond, ctx.cbuf, ctx.gic, ctx.stash, ctx.dims, ctx.with_stash, ctx.chain = cond, cbuf, gic, stash, dims, with_stash, chain
        ctx.enc_stash_f16 = dict(self._enc_stash_f16)
        self._last = ctx
        return z, nll

    def _actnorm_init_walk(self, x, B, T, F, gic, dims, allreduce):
        """Data-dependent ActNorm init from the first timestep, layer by layer (modules.py:32-43,69-70).

        allreduce: callable(sums fp64 tensor) -> number of ranks summed (1 when not distributed)."""
        s = self.spec
        st = _stream()
        xa = self._buf("init_xa", B * s.C)
        xb = self._buf("init_xb", B * s.C)
        hd = self._buf("init_h", B * s.H)
        cd = self._buf("init_c", B * s.H)
        xa[:B * s.C].view(B, s.C).copy_(x[:, s.start, :])
        sums = torch.zeros(2 * s.C, dtype=torch.float64, device=self.device)
        p = self._flow_params()
        for k in range(s.Ks):
            check(self.L.lfi_actnorm_init_stats(xa.data_ptr(), B, s.C, sums.data_ptr(), st), "lfi_actnorm_init_stats")
            world = allreduce(sums)
            check(self.L.lfi_actnorm_init_apply(sums.data_ptr(), float(B * world), s.C, s.actnorm_scale,
                                                self.fview("an_bias")[k].data_ptr(), self.fview("an_logs")[k].data_ptr(),
                                                st), "lfi_actnorm_init_apply")
            check(self.L.lfi_flow_step(C.byref(dims), C.byref(p), self.prep.data_ptr(), k, B, xa.data_ptr(), s.C, None, None,
                                       gic.data_ptr() + 4 * k * F * s.G, xb.data_ptr(), s.C, hd.data_ptr(), cd.data_ptr(),
                                       None, 0, st), "lfi_flow_step")
            xa, xb = xb, xa
        self.run_prep()  # the constant log-det term depends on the new actnorm logs

    @translate_oom
    def backward(self, gscale, after_flow=None):
        """Gradients of gscale * sum(nll) w.r.t. every parameter, written into self.grads (overwritten).

        after_flow: optional callable, invoked once every gradient of the flow block (self.grads[self.flow_offset:]) has
        been enqueued and before the window encoders' BPTT starts (the trainer launches that bucket's all-reduce there)."""
        ctx = self._last
        if ctx is None or not ctx.with_stash:
            raise _lib.LfiError("backward() needs a preceding forward(with_stash=True)")
        s = self.spec
        B, T, N, F = ctx.B, ctx.T, ctx.N, ctx.F
        KD = s.Ks * s.D
        st = _stream()
        dims, p = ctx.dims, self._flow_params()
        self._bwd_skip = 1 if self.backward_product_count(F) == 2 else 0
        bst = self._buf("flow_bstash", self.L.lfi_flow_bstash_floats(C.byref(dims)))
        if ctx.chain:
            return self._backward_chain(ctx, gscale, after_flow, bst)
        check(self.L.lfi_flow_seq_bwd(C.byref(dims), C.byref(p), self.prep.data_ptr(), ctx.stash.data_ptr(), gscale,
                                      bst.data_ptr(), st), "lfi_flow_seq_bwd")
        work = self._buf("scratch.pg", self.L.lfi_flow_param_grads_work_floats(C.byref(dims)))
        g = self._flow_grads()
        # the flow's bias / ActNorm gradients (column sums over the backward stash: HBM streams) and the thin dW product + LU
        # gradient kernel: on the second stream, next to the weight-gradient products and the dpre / cond_transform products
        # (same-box A/B: -0.2 ms per step)
        side = self._fork()
        pg_dims = dims
        if self._skip_bits("flow_pgrads") and (self.precision & 1):
            pg_dims = self._flow_dims(B, N)
            pg_dims.gemm_precision = self.precision | (self._skip_bits("flow_pgrads") << 8)
        check(self.L.lfi_flow_param_grads(C.byref(pg_dims), C.byref(p), self.prep.data_ptr(), ctx.stash.data_ptr(),
                                          bst.data_ptr(), ctx.cbuf.data_ptr(), KD, gscale, C.byref(g), 0, work.data_ptr(),
                                          st, side.cuda_stream if side is not None else None), "lfi_flow_param_grads")
        # d pre-activation of cond_transform, in place over c: dpre = (dgi[k] W_ih[k][:, Ch:]) * leaky'(c)
        dgi_off = (self.L.lfi_flow_bstash_ptr(C.byref(dims), bst.data_ptr(), 1) - bst.data_ptr()) // 4
        # (its epilogue also leaves per-tile column sums of dpre: the cond_transform bias gradient without a 470 MB pass)
        bct_done = self.gemm(F, s.D, s.G, bst, s.G, 1, self.prep, s.D, 0, ctx.cbuf, KD, act=2, slope=0.01, G=ctx.cbuf, ldg=KD,
                             batch=s.Ks, sA=F * s.G, sB=s.G * s.D, sC=s.D, sG=s.D, a_off=dgi_off, b_off=self._wc_offset(),
                             cls="dpre", colsum_into=self.fview("bct", self.grads))
        dpre = ctx.cbuf
        # cond_transform weight / bias gradients for all steps at once
        dwf = self._buf("dwct_f", KD * s.ldf)
        # (all ldf columns: the padding columns of cond are zero, and a 4-float granular N keeps the split-K reduce on 16-byte rows)
        self.gemm(KD, s.ldf, F, dpre, KD, 0, ctx.cond, s.ldf, 0, dwf, s.ldf, tag="gemm_cond_wgrad", splitk=0, cls="cond_wgrad")
        # both copies of a duplicated input column receive the folded column's gradient
        check(self.L.lfi_cols_fold(dwf.data_ptr(), s.ldf, KD, self.unfold.data_ptr(), None, s.E,
                                   self.fview("wct", self.grads).data_ptr(), s.E, st), "lfi_cols_fold")
        # (not on the second stream: streaming dpre for the bias sums next to the weight-gradient product that reads it too was
        # measured 0.1 ms slower per step than in line)
        if not bct_done:
            self.colsum(dpre, KD, 0, F, KD, 1, self.fview("bct", self.grads), 0)
        self._join()
        if after_flow is not None:
            after_flow()
        # gradient of the feature matrix, columns of the trainable encoders only (raw windows are data)
        rnn = [e for e in s.encoders if e.enc in ("rnn", "lstm", "mlp")]
        if rnn:
            col0 = min(e.fcol for e in rnn)
            W = s.Ef - col0
            ldd = (W + 3) // 4 * 4
            dcond = self._buf("dcond", F * ldd)
            # 112 x 5 tiles = 1.09 rounds of the 512 resident workgroups: split K so the tail round is full too
            self.gemm(F, W, KD, dpre, KD, 1, self.wct_f, s.ldf, 0, dcond, ldd, b_off=col0, tag="gemm_cond_dgrad",
                      splitk=0, cls="cond_dgrad")
            self._encoders_backward(rnn, ctx, dcond, ldd, col0)

    def _backward_chain(self, ctx, gscale, after_flow, bst):
        """backward() with every big GEMM operand on planes (see _chain_fwd_ok): the walk leaves dgi as operand planes,
        dW_c = dgi^T c, d pre-activation = dgi W_c (masked by the sign of c's planes, emitted as planes in their place), the
        cond_transform weight gradient and the feature gradient all run on lfi_gemm_planes - every set of planes written once and
        used by rows in one product, transposed in another."""
        s = self.spec
        B, T, N, F = ctx.B, ctx.T, ctx.N, ctx.F
        KD, G, D, Ks = s.Ks * s.D, s.G, s.D, s.Ks
        st = _stream()
        dims, p = ctx.dims, self._flow_params()
        c_r = ctx.cbuf
        nkKD, nkG, nkD = KD // 16, G // 16, D // 16
        dgi_p = self.plane_buf("dgi_planes", self.L.lfi_planes_elems(Ks * F, G) + 256 * G * 2)
        # with two products in their consumers (A rounded to bf16) the lo planes of dgi and of d pre-activation are never
        # fetched: the walk / the dpre epilogue then write the hi planes only (half the plane traffic)
        dgi_hi = all(self._skip_bits(c) & 1 for c in ("dpre", "flow_pgrads"))
        dpre_hi = all(self._skip_bits(c) & 1 for c in ("cond_wgrad", "cond_dgrad"))
        # two-product thin weight-gradient products round dgi | dgh to bf16 on arrival: the walk then leaves those ROWS as bf16 (half the
        # bytes written by the walk and read by the products; gemm_precision bit 16, same bit in both calls; LFI_FLOW_G16=0: fp32 rows)
        pg_skip = self._skip_bits("flow_pgrads")
        g16 = bool((pg_skip & 1) and not (pg_skip & 2) and (self.precision & 1) and os.environ.get("LFI_FLOW_G16", "1") != "0")
        bdims = dims
        if g16:
            bdims = self._flow_dims(B, N)
            bdims.gemm_precision = dims.gemm_precision | (1 << 16)
        check(self.L.lfi_flow_seq_bwd_planes(C.byref(bdims), C.byref(p), self.prep.data_ptr(), ctx.stash.data_ptr(), gscale,
                                             bst.data_ptr(), dgi_p.data_ptr(), 1 if dgi_hi else 0, st), "lfi_flow_seq_bwd_planes")
        work = self._buf("scratch.pg", self.L.lfi_flow_param_grads_work_floats(C.byref(dims)))
        g = self._flow_grads()
        side = self._fork()
        # w_ih[k][:, Ch:] (G x D) = dgi[k]^T c[:, k D:(k + 1) D]: the MFMA-bound flow weight gradient; both operands' planes in
        # transposed use (the sum runs over the F frames = the rows of dgi and of c): step k = row tiles [k F / 32, ..) of dgi's
        # planes and column tiles [k D / 16, ..) of c's
        self.gemm_planes(G, D, F, dgi_p, nkG, c_r, nkKD, self.fview("w_ih", self.grads), s.I, batch=Ks, a_fmt=1, b_fmt=1,
                         a_stride=(F // 32) * nkG * 1024, b_stride=nkD * 1024, sC=G * s.I, c_off=s.Ch,
                         splitk=self._planes_splitk(G, D, F, Ks), tag="gemm_dwc", cls="flow_pgrads")
        pg_dims = dims
        if self._skip_bits("flow_pgrads") and (self.precision & 1):
            pg_dims = self._flow_dims(B, N)
            pg_dims.gemm_precision = self.precision | (self._skip_bits("flow_pgrads") << 8) | ((1 << 16) if g16 else 0)
        check(self.L.lfi_flow_param_grads(C.byref(pg_dims), C.byref(p), self.prep.data_ptr(), ctx.stash.data_ptr(),
                                          bst.data_ptr(), None, KD, gscale, C.byref(g), 0, work.data_ptr(),
                                          st, side.cuda_stream if side is not None else None), "lfi_flow_param_grads")
        # d pre-activation of cond_transform = (dgi[k] W_c[k]) * leaky'(c), written as planes IN PLACE of c's (same blocks, read
        # for the mask and rewritten by the same workgroup; dW_c above was c's last reader); W_c's planes in transposed use (the sum
        # runs over its G gate rows); the epilogue also leaves per-pass column sums: the cond_transform bias gradient
        wr, nkwr = self._wc_r
        bct_done = self.gemm_planes(F, D, G, dgi_p, nkG, wr, nkwr, None, KD, act=2, slope=0.01, batch=Ks, b_fmt=1,
                                    a_stride=(F // 32) * nkG * 1024, b_stride=(G // 32) * nkwr * 1024, sC=D, store=False,
                                    Gr=c_r, gr_nkt=nkKD, Cr=c_r, cr_nkt=nkKD, hi_only=dpre_hi,
                                    colsum_into=self.fview("bct", self.grads), tag="gemm_dpre", cls="dpre")
        if not bct_done:
            raise _lib.LfiError("the dpre product on planes did not take the column-sum epilogue")
        dpre_p = c_r
        # cond_transform weight gradient for all steps at once: dwf (Ks D x ldf) = dpre^T cond, both planes in transposed use
        self._join()
        cp, nkc = self._cond_planes
        dwf = self._buf("dwct_f", KD * s.ldf)
        self.gemm_planes(KD, s.ldf, F, dpre_p, nkKD, cp, nkc, dwf, s.ldf, a_fmt=1, b_fmt=1,
                         splitk=self._planes_splitk(KD, s.ldf, F), tag="gemm_cond_wgrad", cls="cond_wgrad")
        # (the unfold on the second stream beside the feature gradient was measured in round 6: 6.397 against 6.392 ms, nothing)
        check(self.L.lfi_cols_fold(dwf.data_ptr(), s.ldf, KD, self.unfold.data_ptr(), None, s.E,
                                   self.fview("wct", self.grads).data_ptr(), s.E, st), "lfi_cols_fold")
        if after_flow is not None:
            after_flow()
        rnn = [e for e in s.encoders if e.enc in ("rnn", "lstm", "mlp")]
        if rnn:
            # gradient of the feature matrix, from the 32-column block that holds the first trainable encoder's first column:
            # dpre's planes by rows, the folded weights' planes transposed (the sum runs over their Ks D rows)
            col0 = min(e.fcol for e in rnn) // 32 * 32
            W = s.Ef - col0
            ldd = (W + 3) // 4 * 4
            dcond = self._buf("dcond", F * ldd)
            wp, nkw = self._wct_planes
            self.gemm_planes(F, W, KD, dpre_p, nkKD, wp, nkw, dcond, ldd, b_fmt=1, b_off=(col0 // 16) * 1024,
                             splitk=self._planes_splitk(F, W, KD), tag="gemm_cond_dgrad", cls="cond_dgrad")
            self._encoders_backward(rnn, ctx, dcond, ldd, col0)

    def _encoders_backward(self, rnn, ctx, dcond, ldd, col0):
        """BPTT of every trainable window encoder. Largest recurrence first on the main stream (each one's window scatter + dW_ih on
        the second stream under its dW_hh product and the next encoder's recurrence); the SMALLEST recurrent encoder's whole chain
        (p1_speech at final_model.yaml: 2 steps x 128 units, ~0.2 ms of small launches that used to trail the step) goes to the second
        stream first, under the largest encoder's recurrence. LFI_ENC_BWD_SMALL_ON_SIDE=0 keeps everything in line."""
        order = self._bptt_order(rnn)
        recurrent = [e for e in order if e.enc != "mlp"]
        small = None
        if (len(recurrent) >= 3 and os.environ.get("LFI_ENC_BWD_SMALL_ON_SIDE", "1") != "0"
                and os.environ.get("LFI_ENC_BWD_OVERLAP", "1") != "0"):
            side = self._fork()
            if side is not None:
                small = recurrent[-1]
                with self._on(side):
                    self._encoder_backward(small, ctx, dcond, ldd, small.fcol - col0, inline=True)
        for e in order:
            if e is small:
                continue
            if e.enc == "mlp":
                self._mlp_backward(e, ctx, dcond, ldd, e.fcol - col0)
            else:
                self._encoder_backward(e, ctx, dcond, ldd, e.fcol - col0)
        self._join()

    @staticmethod
    def _bptt_order(encoders):
        """Largest recurrence first: each encoder's window scatter + dW_ih run on the second stream under the NEXT encoder's
        BPTT kernel, so what is left uncovered at the end of the step is the LAST encoder's - which should be the smallest
        (the encoders' gradients are independent of each other; the order changes no value)."""
        if os.environ.get("LFI_ENC_BWD_ORDER", "1") == "0":
            return list(encoders)
        return sorted(encoders, key=lambda e: -(e.hist * e.hid * e.hid))

    def _mlp_backward(self, e, ctx, dcond, lddcond, col):
        """Linear + LeakyReLU window encoder: dpre = dfeat * leaky'(feat); dW = dpre^T window, db = colsum(dpre)."""
        s = self.spec
        F = ctx.F
        K = e.in_dim * e.hist
        check(self.L.lfi_leaky_grad(dcond.data_ptr() + 4 * col, lddcond, ctx.cond.data_ptr() + 4 * e.fcol, s.ldf, F, e.hid,
                                    0.01, _stream()), "lfi_leaky_grad")
        win = self._ws["enc_win." + e.name]
        gname = "enc.%s." % e.name
        self.gemm(e.hid, K, F, dcond, lddcond, 0, win, e.win, 0, self.view(gname + "mlp_weight", self.grads), K,
                  splitk=max(1, min(32, F // 1024)), a_off=col)
        self.colsum(dcond, lddcond, 0, F, e.hid, 1, self.view(gname + "mlp_bias", self.grads), 0, x_off=col)

    def _encoder_backward(self, e, ctx, dcond, lddcond, col, inline=False):
        """BPTT of one window encoder + its weight / bias gradients. inline=True: every launch on the CURRENT stream, with split-K
        workspaces of their own (the caller runs this encoder's whole chain on the second stream next to another encoder's)."""
        s = self.spec
        B, N, F = ctx.B, ctx.N, ctx.F
        x = ctx.batch[e.name]
        Tx, hid, G3 = x.shape[1], e.hid, e.ng * e.hid
        st = _stream()
        d = EncDesc(B, Tx, N, s.start - (1 if e.name == "p1_face" else 0), e.hist, hid, lddcond, col, self.precision, 0,
                    1 if e.enc == "lstm" else 0,
                    # two-product BPTT recurrence + bf16 gradient stash: only together with a two-product (A rounded) dW_hh
                    # product, the stash's other reader
                    1 if ((self._skip_bits("enc_bptt") & 1) and (self._skip_bits("enc_dwhh") & 3) == 1) else 0,
                    1 if ctx.enc_stash_f16.get(e.name) else 0)
        gates = self._ws["enc_gates." + e.name]
        hseq = self._ws["enc_hseq." + e.name]
        compact = bool(self.L.lfi_encode_windows_compact_dgi(C.byref(d)))   # fused GRU backward: dgi = its n block only
        dgi = self._buf("enc_dgi." + e.name, e.hist * F * (hid if compact else G3))
        dgh = self._buf("enc_dgh." + e.name, e.hist * F * G3)
        work = self._buf("scratch.enc_bwd." + e.name, self.L.lfi_encode_windows_work_floats(C.byref(d)))
        whh = self.view("enc.%s.weight_hh" % e.name)
        prow = self.L.lfi_encode_windows_bias_rows(C.byref(d))
        part = self._buf("enc_bias_part." + e.name, prow * 4 * hid) if prow else None
        check(self.L.lfi_encode_windows_bwd(C.byref(d), dcond.data_ptr(), lddcond, whh.data_ptr(), gates.data_ptr(),
                                            hseq.data_ptr(), dgi.data_ptr(), dgh.data_ptr(), ptr(part), work.data_ptr(), st),
              "lfi_encode_windows_bwd")
        mk = None if ctx.masks is None else ctx.masks.get(e.name)
        dxp = self._buf("dxp." + e.name, B * Tx * G3)
        gname = "enc.%s." % e.name
        rows = B * Tx
        xa, ldi = self._ws.get("xpad." + e.name), (e.in_dim + 3) // 4 * 4   # the padded copy the forward pass made
        if e.in_dim % 4 == 0 or xa is None or os.environ.get("LFI_NO_XPAD") == "1":
            xa, ldi = x, e.in_dim
        # the input side (window scatter: one HBM pass over dgi / dgh, then the thin dW_ih product) on the second stream, next
        # to the MFMA-bound dW_hh product of the hidden side - and to the next modality's backward recurrence
        side = self._fork() if (os.environ.get("LFI_ENC_BWD_OVERLAP", "1") != "0" and not inline) else None
        with self._on(side):
            check(self.L.lfi_encode_windows_scatter(C.byref(d), dgi.data_ptr(), dgh.data_ptr(), ptr(mk), dxp.data_ptr(),
                                                    _stream()), "lfi_encode_windows_scatter")
            self.gemm(G3, e.in_dim, rows, dxp, G3, 0, xa, ldi, 0, self.view(gname + "weight_ih", self.grads), e.in_dim,
                      splitk=self._long_k_splitk(G3, e.in_dim, rows, kmin=512),
                      ws="scratch.gemm_splitk.side3" if inline else "scratch.gemm_splitk.side", cls="enc_dwih")
        gbi, gbh = self.view(gname + "bias_ih", self.grads), self.view(gname + "bias_hh", self.grads)
        if part is not None:  # per-workgroup partial sums of (d r, d z, d n, d n * r) left by the fused backward kernel
            check(self.L.lfi_encode_windows_bias_grads(part.data_ptr(), prow, hid, gbi.data_ptr(), gbh.data_ptr(), st),
                  "lfi_encode_windows_bias_grads")
        else:
            self.colsum(dgi, G3, 0, e.hist * F, G3, 1, gbi, 0)
        if e.hist > 1:
            kk = (e.hist - 1) * F
            # (a bf16 gradient stash: dgh holds bf16 values in the same [hist][F][3 hid] order - the two-product kernel's A operand)
            g16 = bool(self.L.lfi_encode_windows_grad_stash_bf16(C.byref(d)))
            self.gemm(G3, hid, kk, dgh, G3, 0, hseq, hid, 0, self.view(gname + "weight_hh", self.grads), hid,
                      splitk=self._long_k_splitk(G3, hid, kk), a_off=F * G3, cls="enc_dwhh", a_bf16=g16,
                      ws="scratch.gemm_splitk.side2" if inline else "scratch.gemm_splitk")
        else:
            self.view(gname + "weight_hh", self.grads).zero_()
        if part is None:
            self.colsum(dgh, G3, 0, e.hist * F, G3, 1, gbh, 0)

    @translate_oom
    def encode_condition(self, condition, masks=None):
        """FeatureEncoder.forward on ONE timestep's conditioning dict (models.py:127-145): {"prev_p1_face": (B, h1, C),
        "p2_face": (B, h, C), "p1_speech" / "p2_speech": (B, h, S), "frame_nb": (B, 1)} -> (B, E) features in the
        reference's layout (a GRU encoder's output twice)."""
        s = self.spec
        data = {("p1_face" if k == "prev_p1_face" else k): v for k, v in condition.items()}
        faces = data["p1_face"]
        B = faces.shape[0]
        cond = self._buf("step_cond", B * s.ldf, zero=True)
        self.build_features(data, faces, B, 0, masks, cond, with_stash=False, windows=True)
        return cond[:B * s.ldf].view(B, s.ldf).index_select(1, self.unfold.long())

    # ------------------------------------------------------------------ module-level FlowNet / FlowStep / Glow calls
    @translate_oom
    def flow_net(self, x, cond, h_prev, c_prev, reverse, init_actnorm=None):
        """FlowNet.encode / decode (models.py:444-462) on ONE (B, C) batch with explicit recurrent state.

        cond: (B, E) feature vector in the reference's layout (what FeatureEncoder.forward returns). h_prev / c_prev:
        per flow step a (B, H) tensor or None (= the zero state of the first call after init_rnn_hidden).
        init_actnorm: None, or the all-reduce callable of the data-dependent ActNorm init (modules.py:32-43): each layer
        is initialised from its own input before it runs, as ActNorm2d.forward does on its first training-mode call.
        -> out (B, C), logdet (B,) (actnorm + invconv + coupling; the reverse pass returns the negated sum),
           h_new (Ks, B, H), c_new (Ks, B, H) or None."""
        s = self.spec
        B = x.shape[0]
        for t, name, w in ((x, "input", s.C), (cond, "condition", s.E)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape == (B, w)):
                raise ValueError("%s: expected a float32 GPU tensor of shape (%d, %d), got %s %s on %s"
                                 % (name, B, w, tuple(t.shape), t.dtype, t.device))
        x, cond = x.contiguous(), cond.contiguous()
        st = _stream()
        self.run_prep(with_inverse=bool(reverse))
        KD = s.Ks * s.D
        cb = self._buf("step_c", B * KD)
        # the UNfolded cond_transform weights: a caller's feature vector need not repeat the GRU halves
        self.gemm(B, KD, s.E, cond, s.E, 1, self.fview("wct"), s.E, 1, cb, KD, bias=self.fview("bct"), act=1, slope=0.01)
        gic = self._buf("step_gic", s.Ks * B * s.G)
        self.gemm(B, s.G, s.D, cb, KD, 1, self.prep, s.D, 1, gic, s.G, bias=self.fview("b_ih"),
                  batch=s.Ks, sA=s.D, sB=s.G * s.D, sC=B * s.G, sBias=s.G, b_off=self._wc_offset())
        dims = self._flow_dims(B, 1)
        p = self._flow_params()
        lstm = s.rnn_type == "lstm"
        f32 = dict(dtype=torch.float32, device=self.device)
        h_new = torch.empty(s.Ks, B, s.H, **f32)
        c_new = torch.empty(s.Ks, B, s.H, **f32) if lstm else None
        ld = torch.zeros(B, **f32)
        bufs = (torch.empty(B, s.C, **f32), torch.empty(B, s.C, **f32))
        sums = torch.zeros(2 * s.C, dtype=torch.float64, device=self.device) if init_actnorm is not None else None
        src = x
        order = range(s.Ks - 1, -1, -1) if reverse else range(s.Ks)
        for i, k in enumerate(order):
            if sums is not None:
                sums.zero_()
                check(self.L.lfi_actnorm_init_stats(src.data_ptr(), B, s.C, sums.data_ptr(), st), "lfi_actnorm_init_stats")
                world = init_actnorm(sums)
                check(self.L.lfi_actnorm_init_apply(sums.data_ptr(), float(B * world), s.C, s.actnorm_scale,
                                                    self.fview("an_bias")[k].data_ptr(), self.fview("an_logs")[k].data_ptr(),
                                                    st), "lfi_actnorm_init_apply")
            hp = None if h_prev is None else h_prev[k]
            cp = None if (c_prev is None or not lstm) else c_prev[k]
            for t in (hp, cp):
                if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.shape == (B, s.H) and t.is_contiguous()):
                    raise ValueError("recurrent state of flow step %d: expected a contiguous float32 GPU (%d, %d) tensor" % (k, B, s.H))
            dst = bufs[i & 1]
            check(self.L.lfi_flow_step(C.byref(dims), C.byref(p), self.prep.data_ptr(), k, B, src.data_ptr(), s.C, ptr(hp),
                                       ptr(cp), gic.data_ptr() + 4 * k * B * s.G, dst.data_ptr(), s.C, h_new[k].data_ptr(),
                                       ptr(None if c_new is None else c_new[k]), ld.data_ptr(), 1 if reverse else 0, st),
                  "lfi_flow_step")
            src = dst
        if sums is not None:
            self.run_prep(with_inverse=bool(reverse))  # the constant log-det term depends on the new actnorm logs
        const = self.prep[self._ldconst_offset()]
        return src, (ld - const if reverse else ld + const), h_new, c_new

    # ------------------------------------------------------------------ optimiser
    @staticmethod
    def adam_step_floats(lr, beta1, beta2, step_count):
        """(step_size, 1 / sqrt(1 - beta2^t)) exactly as lfi_adam_clip_step derives them from its float arguments (doubles, then
        rounded to fp32): what a captured step reads from the device block lfi_set_step_params fills."""
        import numpy as np
        b1, b2, lr64 = float(np.float32(beta1)), float(np.float32(beta2)), float(np.float32(lr))
        bc1, bc2 = 1.0 - b1 ** int(step_count), 1.0 - b2 ** int(step_count)
        return float(np.float32(lr64 / bc1)), float(np.float32(1.0 / math.sqrt(bc2)))

    def _claim_optimizer(self, name):
        """The state buffers mean different things per optimiser (optimizer_state): a state loaded from a checkpoint of one
        optimiser must not be stepped by another (Adam's moments were silently reused as a momentum buffer)."""
        have = self.__dict__.get("_opt_name")
        if have is not None and have != name:
            raise ValueError("the engine holds optimiser state of %r (checkpoint or earlier steps); stepping it with %r would "
                             "reinterpret its buffers - start from fresh state (load_optimizer_state({})) or keep Optim.name" % (have, name))
        self._opt_name = name

    @translate_oom
    def optimizer_step(self, lr, beta1, beta2, eps, clip=0.0, gmul=1.0, hyper_dev=None, weight_decay=0.0, amsgrad=False):
        """clip_grad_norm_(clip) + Adam on the flat buffers (lets_face_it_glow.py:61-72; final_model.yaml:126).
        hyper_dev: device pointer to (step_size, 1 / sqrt(1 - beta2^t)) - a captured step; the caller keeps step_count.
        weight_decay / amsgrad: torch.optim.Adam's (the reference forwards Optim["args"]["adam"] verbatim); the running maximum of
        the second moment lives in `opt_aux`."""
        self._claim_optimizer("adam")
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        if amsgrad and self.__dict__.get("opt_aux") is None:
            self.opt_aux = torch.zeros_like(self.params)
        st = _stream()
        if clip and clip > 0:
            check(self.L.lfi_grad_sumsq(self.grads.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                        self.sumsq_work.data_ptr(), st), "lfi_grad_sumsq")
        if weight_decay or amsgrad:
            if hyper_dev is None:
                self.step_count += 1
            check(self.L.lfi_adam_clip_step_ex(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                               self.adam_v.data_ptr(), ptr(self.opt_aux if amsgrad else None), self.n_params,
                                               self.sumsq.data_ptr(), float(clip or 0.0), gmul, lr, beta1, beta2, eps,
                                               float(weight_decay), self.step_count, hyper_dev, st), "lfi_adam_clip_step_ex")
            return
        if hyper_dev is not None:
            check(self.L.lfi_adam_clip_step_dev(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                                self.adam_v.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                                float(clip or 0.0), gmul, beta1, beta2, eps, hyper_dev, st), "lfi_adam_clip_step_dev")
            return
        self.step_count += 1
        check(self.L.lfi_adam_clip_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                        self.adam_v.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                        float(clip or 0.0), gmul, lr, beta1, beta2, eps, self.step_count, st),
              "lfi_adam_clip_step")

    @translate_oom
    def optimizer_step_sgd(self, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, clip=0.0, gmul=1.0):
        """clip_grad_norm_(clip) + torch.optim.SGD on the flat buffers (lets_face_it_glow.py:61-72 with Optim.name = "sgd";
        final_model.yaml:98-99: momentum 0.9). The momentum buffer lives in `adam_m` (one optimiser per model)."""
        self._claim_optimizer("sgd")
        # torch.optim.SGD's "first step" (buf = g) is "no momentum buffer yet", not "step 1": a reference checkpoint carries a
        # buffer and no step count (the imported buffer was overwritten with g on the first resumed step)
        first = momentum and (self.adam_m is None or not self.__dict__.get("_momentum_inited", False))
        if momentum and self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
        st = _stream()
        if clip and clip > 0:
            check(self.L.lfi_grad_sumsq(self.grads.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                        self.sumsq_work.data_ptr(), st), "lfi_grad_sumsq")
        self.step_count += 1
        check(self.L.lfi_sgd_clip_step(self.params.data_ptr(), self.grads.data_ptr(), ptr(self.adam_m if momentum else None),
                                       self.n_params, self.sumsq.data_ptr(), float(clip or 0.0), gmul, lr, float(momentum),
                                       float(dampening), float(weight_decay), 1 if nesterov else 0, 1 if first else 2, st),
              "lfi_sgd_clip_step")
        if momentum:
            self._momentum_inited = True

    @translate_oom
    def optimizer_step_rmsprop(self, lr, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False, clip=0.0, gmul=1.0):
        """clip_grad_norm_(clip) + torch.optim.RMSprop on the flat buffers (Optim.name = "rmsprop"; final_model.yaml:96-97: eps 1e-8).
        Square average in `adam_v`, momentum buffer in `adam_m`, the centered variant's gradient average in `opt_aux`."""
        self._claim_optimizer("rmsprop")
        if self.adam_v is None:
            self.adam_v = torch.zeros_like(self.params)
        if momentum and self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
        if centered and self.__dict__.get("opt_aux") is None:
            self.opt_aux = torch.zeros_like(self.params)
        st = _stream()
        if clip and clip > 0:
            check(self.L.lfi_grad_sumsq(self.grads.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                        self.sumsq_work.data_ptr(), st), "lfi_grad_sumsq")
        self.step_count += 1
        check(self.L.lfi_rmsprop_clip_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_v.data_ptr(),
                                           ptr(self.adam_m if momentum else None), ptr(self.opt_aux if centered else None),
                                           self.n_params, self.sumsq.data_ptr(), float(clip or 0.0), gmul, lr, float(alpha), float(eps),
                                           float(weight_decay), float(momentum), st), "lfi_rmsprop_clip_step")

    def optimizer_state(self):
        """The optimiser's state buffers and step count (checkpoints, engine re-binds). Adam: first / second moments, `opt_aux` =
        amsgrad's running maximum; SGD: `adam_m` = momentum buffer; RMSprop: `adam_v` = square average, `adam_m` = momentum buffer,
        `opt_aux` = centered gradient average. `optimizer` names whose state this is (None before the first step) - another
        optimiser refuses to step it. Buffers are None before the first step."""
        aux = self.__dict__.get("opt_aux")
        return {"step_count": int(self.step_count),
                "optimizer": self.__dict__.get("_opt_name"),
                "momentum_inited": bool(self.__dict__.get("_momentum_inited", False)),
                "opt_aux": None if aux is None else aux.detach().clone(),
                # the dropout-mask stream is keyed on (seed, call counter): a resumed / re-bound run must not replay the masks
                # of steps 1..k
                "mask_calls": int(self._mask_calls),
                "adam_m": None if self.adam_m is None else self.adam_m.detach().clone(),
                "adam_v": None if self.adam_v is None else self.adam_v.detach().clone()}

    def load_optimizer_state(self, state):
        self.step_count = int(state.get("step_count", 0))
        self._mask_calls = int(state.get("mask_calls", self._mask_calls))
        self._opt_name = state.get("optimizer")     # None (older checkpoints, fresh state): the first step claims it
        # (checkpoints from before round 6 carry no flag: a momentum buffer that exists has been written)
        self._momentum_inited = bool(state.get("momentum_inited", state.get("adam_m") is not None))
        for name in ("adam_m", "adam_v", "opt_aux"):
            t = state.get(name)
            if t is None:
                setattr(self, name, None)
                continue
            if t.numel() != self.n_params:
                raise ValueError("optimizer state holds %d floats, the model has %d parameters" % (t.numel(), self.n_params))
            setattr(self, name, t.to(device=self.device, dtype=torch.float32).reshape(-1).clone())

    def grad_norm(self):
        check(self.L.lfi_grad_sumsq(self.grads.data_ptr(), self.n_params, self.sumsq.data_ptr(),
                                    self.sumsq_work.data_ptr(), _stream()), "lfi_grad_sumsq")
        return float(self.sumsq.sqrt().item())

    # ------------------------------------------------------------------ sampling / inversion
    @translate_oom
    def sample(self, seq_len, data, noise, masks=None):
        """SeqGlow.inference (models.py:567-596); see _sample. With the static part on a partial-chip stream (LFI_SAMPLE_STATIC_CUS)
        the whole call runs on a private non-blocking stream between two joins with the caller's: hipExtStreamCreateWithCUMask makes a
        BLOCKING stream, which takes turns with the legacy default stream - and that is the stream most callers are on."""
        caller = torch.cuda.current_stream(self.device)
        # (also without a partial-chip stream when the caller is on the legacy default stream: the per-run graphs are not replayed
        # there - with fewer than four hardware queues per process, GPU_MAX_HW_QUEUES=3, hipGraphLaunch into the NULL stream was seen to
        # segfault inside the HIP runtime on the training step's graph)
        if self._sample_static_cus(seq_len - self.spec.start) <= 0 and caller != torch.cuda.default_stream(self.device):
            return self._sample(seq_len, data, noise, masks)
        if self._sample_stream is None:
            self._sample_stream = torch.cuda.Stream(device=self.device)
        own = self._sample_stream
        own.wait_stream(caller)
        with torch.cuda.stream(own):
            out = self._sample(seq_len, data, noise, masks)
        caller.wait_stream(own)
        out.record_stream(caller)
        return out

    def _sample(self, seq_len, data, noise, masks=None):
        """SeqGlow.inference (models.py:567-596) with the prior noise given: (seq_len - start, B, C), already * eps.

        The generated frames are produced in a few RUNS (LFI_SAMPLE_RUNS, default 4 from 64 frames up, 6 from 96 when the static part
        has its own CUs): everything of a run that
        does not depend on generated frames - window encoders, the non-autoregressive cond_transform columns - is queued on the
        second stream behind the previous run's chain of dependent reverse cells, so that only the first run's static part stands in
        front of the first frame. On an ORDINARY second stream that does not shorten the call (a chain launch holds every CU: the
        call takes static + chain whatever the number of runs, profiles/round5_sampler_one_run_kernel_stats.md); on the partial-chip
        stream (LFI_SAMPLE_STATIC_CUS, the default on 256-CU devices: the static part owns half of every XCD beside the chain) it
        does: 55.0 -> 50.0 ms per call on one box, and six runs instead of four another 46.7 -> 45.5."""
        s = self.spec
        seed = data["p1_face"]
        B = seed.shape[0]
        nframes = seq_len - s.start
        if nframes <= 0:
            raise ValueError("seq_len %d does not exceed the longest history %d" % (seq_len, s.start))
        self._check_input(seed, "p1_face", B, s.start, s.C)
        if tuple(noise.shape) != (nframes, B, s.C) or not noise.is_contiguous() or noise.dtype != torch.float32:
            raise ValueError("noise must be a contiguous float32 (%d, %d, %d) tensor" % (nframes, B, s.C))
        self._check_last_sample()
        F = nframes * B
        KD = s.Ks * s.D
        main = torch.cuda.current_stream(self.device)
        # ---- range guard of the fp16-piece arithmetic: ONE reduction over the caller's tensors, the noise and the parameters,
        # read back through pinned memory on the side stream; the host waits for it only when it picks the per-frame arithmetic,
        # with the first run's static part already queued behind it (no drained queue, no per-tensor sync)
        fp, guard = self.sample_frame_precision, None
        if fp is None:
            fp = 0
            if self.precision == 1:
                fp = 5 if self._sample_fp16_unsafe else None
                if fp is None:
                    guard = self._launch_range_guard([self.params, noise] + [v for v in data.values() if torch.is_tensor(v)
                                                                                and v.dtype == torch.float32 and v.is_cuda])
        ev_static = self._tic("sample_static")
        faces = self._buf("sample_faces", B * seq_len * s.C)[:B * seq_len * s.C].view(B, seq_len, s.C)
        faces.zero_()
        faces[:, :s.start].copy_(seed[:, :s.start])
        self.run_prep(with_inverse=True)
        cond = self._buf("cond", F * s.ldf)
        e1 = s.encoders[0]             # prev_p1_face: the only autoregressive input
        c1 = (e1.fdim + 3) // 4 * 4    # first column after its block (blocks start on 4-float boundaries)
        pre = self._buf("pre_static", F * KD)
        planes_ok = s.Ef > c1 and self.precision == 1 and os.environ.get("LFI_PGEMM", "1") != "0"
        wp = nkw = None
        if planes_ok:
            wp, nkw = self.planes("wct_planes_static", self.wct_f, s.ldf, KD, s.Ef - c1, x_off=c1)
        part = self._sample_static_cus(nframes)
        runs = self._sample_runs(nframes, part > 0)

        def static(o, n):
            """Frames [o, o + n) of the call: features of everything but prev_p1_face, then through the static cond_transform columns
            (no activation yet) into their rows of `pre`."""
            sub = {k: (v[:, o:o + s.start + n].contiguous() if (torch.is_tensor(v) and v.dim() == 3 and k != "p1_face") else v)
                   for k, v in data.items()} if (o > 0 or n < nframes) else data
            mk = None if masks is None else {k: v[o:o + n].contiguous() for k, v in masks.items()}
            rows = n * B
            cnd = cond[o * B * s.ldf:]
            prs = pre[o * B * KD:]
            self.build_features(sub, None, B, s.start + n, mk, cnd, with_stash=False, skip_p1=True, sampling=True, frame0=o)
            if planes_ok:
                # the static columns of cond_transform for the run's frames (F x Ks D x 640 at final widths: the largest product of a
                # sampling call) on pre-split planes, as the training step's cond_transform forward
                cp, nkc = self.planes("cond_planes", cnd, s.ldf, rows, s.Ef - c1, x_off=c1)
                self.gemm_planes(rows, KD, s.Ef - c1, cp, nkc, wp, nkw, prs, KD, bias=self.fview("bct"), cls="cond_fwd")
            elif s.Ef > c1:
                self.gemm(rows, KD, s.Ef - c1, cnd, s.ldf, 1, self.wct_f, s.ldf, 1, prs, KD, bias=self.fview("bct"), a_off=c1, b_off=c1)
            else:
                prs[:rows * KD].view(rows, KD).copy_(self.fview("bct").reshape(1, KD).expand(rows, KD))

        static(*runs[0])
        self._toc("sample_static", ev_static)
        events = [None]
        if len(runs) > 1:
            side = self._fork_partial(part) if part > 0 else self._fork()
            if side is None:
                for o, n in runs[1:]:
                    static(o, n)
                    events.append(None)
            else:
                with self._on(side):
                    for o, n in runs[1:]:
                        static(o, n)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        events.append(ev)
        if guard is not None:
            fp = 9 if self._read_range_guard(guard) <= 1e3 else 5
        dims = self._flow_dims(B, nframes)
        dims.gemm_precision = int(fp)
        h = self._buf("sample_h", s.Ks * B * s.H, zero=True)
        cs = self._buf("sample_c", s.Ks * B * s.H, zero=True) if s.rnn_type == "lstm" else None
        work = self._buf("scratch.sample", self.L.lfi_flow_sample_work_floats(C.byref(dims)))
        nz = self._buf("sample_noise", nframes * B * s.C)   # engine-owned copy: the captured graph reads a stable address
        nz[:nframes * B * s.C].view_as(noise).copy_(noise)
        p = self._flow_params()
        hist1 = e1.hist
        # an encoded prev_p1_face window (enc: mlp / rnn / lstm) is re-encoded for every generated frame inside the sampler
        p1 = P1Enc()
        p1.kind, p1.hid, p1.col = {"none": 0, "mlp": 1, "rnn": 2, "lstm": 3}[e1.enc], e1.hid, e1.fcol
        if e1.enc == "mlp":
            p1.w1, p1.b1 = self.view("enc.p1_face.mlp_weight").data_ptr(), self.view("enc.p1_face.mlp_bias").data_ptr()
        elif e1.enc in ("rnn", "lstm"):
            for leaf in ENC_LEAVES:
                setattr(p1, {"weight_ih": "w_ih", "weight_hh": "w_hh", "bias_ih": "b_ih", "bias_hh": "b_hh"}[leaf],
                        self.view("enc.p1_face." + leaf).data_ptr())
        p1work = self._buf("scratch.sample_p1", self.L.lfi_flow_sample_p1_work_floats(C.byref(dims), C.byref(p1), hist1))

        def launch(o, n):
            check(self.L.lfi_flow_sample_seq_from(C.byref(dims), C.byref(p), self.prep.data_ptr(), self.wct_f.data_ptr(), s.ldf, hist1,
                                                  pre.data_ptr() + 4 * o * B * KD, nz.data_ptr() + 4 * o * B * s.C, faces.data_ptr(),
                                                  seq_len, s.start + o, n, o, h.data_ptr(), ptr(cs), C.byref(p1), p1work.data_ptr(),
                                                  work.data_ptr(), _stream()), "lfi_flow_sample_seq_from")

        # The per-frame chain (2 small GEMMs + Ks reverse cells, ~19 launches x nframes) is launch-bound on the host at
        # small batch: from the second call of a shape on, every run of it is replayed as ONE hipGraph (captured once; every buffer
        # it touches is engine-owned and keeps its address). LFI_NO_GRAPH=1 keeps eager launches.
        key = (B, seq_len, self.precision, int(dims.gemm_precision), faces.data_ptr(), pre.data_ptr(), nz.data_ptr(), h.data_ptr(),
               self.prep.data_ptr(), tuple(runs))
        graphs = self._sample_graphs.get(key)
        if graphs is None and os.environ.get("LFI_NO_GRAPH") != "1" and self._sample_seen.get(key):
            torch.cuda.synchronize()
            graphs = []
            for o, n in runs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    launch(o, n)
                graphs.append(g)
            self._sample_graphs = {key: graphs}   # one shape at a time: buffers are shared between shapes
        elif graphs is None:
            self._sample_seen = {key: True}
        ev = self._tic("sample_graph")
        for i, (o, n) in enumerate(runs):
            if events[i] is not None:
                main.wait_event(events[i])
            if graphs is not None:
                graphs[i].replay()
            else:
                launch(o, n)
        self._toc("sample_graph" if graphs is not None else "sample_chain_eager", ev)
        self._join()
        out = faces[:, s.start:].clone()
        if fp == 9:
            self._watch_sample_output(out)
        return out

    # ---- helpers of sample()
    def _sample_static_cus(self, nframes):
        """CUs of every XCD that the static part of runs 2.. gets beside the chain (0: an ordinary second stream, where the two
        only take turns). LFI_SAMPLE_STATIC_CUS overrides; by default HALF of each XCD on the 256-CU card, where the chain's 256
        one-per-CU workgroups then go in exactly two rounds (14 or 18 of 32 measured 7 / 2 ms worse per call)."""
        if os.environ.get("LFI_NO_OVERLAP") == "1" or nframes <= 0 or len(self._sample_runs(nframes)) < 2 or self._partial_refused:
            return 0
        want = os.environ.get("LFI_SAMPLE_STATIC_CUS")
        if want is not None:
            return max(int(want), 0)
        return 16 if torch.cuda.get_device_properties(self.device).multi_processor_count == 256 else 0

    @staticmethod
    def _sample_runs(nframes, beside=False):
        """[(first frame, frames)] of the runs a sampling call is cut into (see sample()). beside: the static part of runs 2.. has its
        own share of the chip (_sample_static_cus): shorter runs then - six from 96 frames up (46.7 -> 45.5 ms per 1024 x 300 call
        against four, same box; seven the same, eight slower) - because only the first run's static part stands in front of the call."""
        want = os.environ.get("LFI_SAMPLE_RUNS")
        if want and "," in want:  # an explicit list of run lengths; a last run takes what is left
            runs, o = [], 0
            for n in (int(v) for v in want.split(",")):
                n = min(n, nframes - o)
                if n > 0:
                    runs.append((o, n))
                    o += n
            if o < nframes:
                runs.append((o, nframes - o))
            return runs
        nruns = int(want) if want else ((6 if beside and nframes >= 96 else 4) if nframes >= 64 else 1)
        nruns = max(1, min(nruns, nframes))
        base, extra, runs, o = nframes // nruns, nframes % nruns, [], 0
        for i in range(nruns):
            n = base + (1 if i < extra else 0)
            runs.append((o, n))
            o += n
        return runs

    def _pinned_word(self, name):
        t = self.__dict__.get(name)
        if t is None:
            t = torch.zeros(1, dtype=torch.int32).pin_memory()
            setattr(self, name, t)
        return t

    def _launch_range_guard(self, tensors):
        """One launch: bit pattern of max |v| over `tensors` (lfi_absmax_f32), copied to pinned memory on the guard stream. ->
        (pinned word, event)."""
        main = torch.cuda.current_stream(self.device)
        gs = self.__dict__.get("_guard_stream")
        if gs is None:
            gs = self._guard_stream = torch.cuda.Stream(device=self.device)
        tensors = [t.contiguous() for t in tensors if t.numel() > 0]
        if len(tensors) > 8:
            raise ValueError("range guard: %d float tensors in one sampling call (at most 8)" % len(tensors))
        word = self._buf_i32("sample_guard_word", 1)
        n = len(tensors)
        ptrs, ns = (C.c_void_p * max(n, 1))(), (C.c_long * max(n, 1))()
        for i, t in enumerate(tensors):
            ptrs[i], ns[i] = t.data_ptr(), t.numel()
        gs.wait_stream(main)          # the tensors are complete
        pinned = self._pinned_word("_guard_pinned")
        with torch.cuda.stream(gs):
            check(self.L.lfi_absmax_f32(n, ptrs, ns, word.data_ptr(), gs.cuda_stream), "lfi_absmax_f32")
            pinned.copy_(word, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(gs)
        for t in tensors:
            t.record_stream(gs)
        return pinned, ev

    @staticmethod
    def _read_range_guard(guard):
        """max |v| as a float (inf / nan when an input held one: both fail `<= limit`)."""
        import struct
        pinned, ev = guard
        ev.synchronize()
        return struct.unpack("<f", struct.pack("<I", int(pinned.item()) & 0xffffffff))[0]

    def _buf_i32(self, name, n):
        t = self._ws.get(name)
        if t is None or t.numel() < n:
            t = torch.zeros(n, dtype=torch.int32, device=self.device)
            self._ws[name] = t
            self._order_birth()
        return t

    def _watch_sample_output(self, out):
        """fp16 pieces turn a value beyond 65504 - e.g. a generated frame of a diverging checkpoint fed back through the
        prev_p1_face window - into inf - inf = NaN where fp32 would stay finite. The generated frames are scanned asynchronously
        (same guard kernel); a non-finite result is reported at the NEXT engine call (_check_last_sample: a warning, and the engine
        stays with the six-product bf16 form, which has no range caveat, from then on)."""
        pinned = self._pinned_word("_watch_pinned")
        main = torch.cuda.current_stream(self.device)
        word = self._buf_i32("sample_watch_word", 1)
        ptrs, ns = (C.c_void_p * 1)(out.data_ptr()), (C.c_long * 1)(out.numel())
        check(self.L.lfi_absmax_f32(1, ptrs, ns, word.data_ptr(), main.cuda_stream), "lfi_absmax_f32")
        pinned.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        self._sample_watch = (pinned, ev)

    def _check_last_sample(self):
        w = self.__dict__.get("_sample_watch")
        if w is None or not w[1].query():      # never a host wait: a scan still in flight is looked at by a later call
            return
        self._sample_watch = None
        amax = self._read_range_guard(w)
        if not (amax <= 65504.0):     # inf or NaN in the frames of the previous fp16-piece sampling call
            import warnings
            self._sample_fp16_unsafe = True
            warnings.warn("the previous sampling call produced non-finite frames (max |x| = %r) in its fp16-piece arithmetic: this "
                          "engine samples with six bf16 products (no range caveat) from now on; re-run that call" % amax)

    @translate_oom
    def invert(self, z_seq, batch, masks=None):
        """SeqGlow.invert (models.py:617-645): teacher-forced reverse pass. z_seq (N, B, C) -> x (N, B, C), logdet (N, B)."""
        s = self.spec
        x = batch["p1_face"]
        B, T = x.shape[0], x.shape[1]
        N = z_seq.shape[0]
        F = N * B
        Tn = s.start + N
        # The inverse pass is ill-conditioned where the forward pass is not: an error of 2^-16 in the coupling nets' conditioning
        # input (three bf16 products) comes back amplified by the chain of inverse couplings and inverse 1x1 convolutions - at 96 flow
        # steps 1e-2 of x against 2.5e-3 for plain fp32 (tests/test_gpu_deep_parity.py) - so `invert`, a validation call
        # (Validation.check_invertion, mimicry_logger.py:241-251), runs its window encoders and conditioning products in the exact
        # f32 arithmetic whatever the engine mode.
        keep_precision, self.precision = self.precision, 0
        try:
            self.run_prep(with_inverse=True)
            cond = self._buf("cond", F * s.ldf)
            self.build_features(batch, x, B, Tn, masks, cond, with_stash=False)
            _, gic = self._project(cond, F)
        finally:
            self.precision = keep_precision
        dims = self._flow_dims(B, N)
        p = self._flow_params()
        st = _stream()
        out = torch.empty(N, B, s.C, dtype=torch.float32, device=self.device)
        h = self._buf("sample_h", s.Ks * B * s.H, zero=True)
        cs = self._buf("sample_c", s.Ks * B * s.H, zero=True)
        z_seq = z_seq.contiguous()
        ldconst = self.prep[self._ldconst_offset()]
        if self.L.lfi_flow_seq_rev_ok(C.byref(dims)):
            # ONE persistent launch for the whole (timestep, flow step) grid: the reverse twin of the forward walk
            ld = torch.empty(N, B, dtype=torch.float32, device=self.device)
            work = self._buf("scratch.invert", self.L.lfi_flow_seq_rev_work_floats(C.byref(dims)))
            check(self.L.lfi_flow_seq_rev(C.byref(dims), C.byref(p), self.prep.data_ptr(), z_seq.data_ptr(), gic.data_ptr(),
                                          out.data_ptr(), ld.data_ptr(), h.data_ptr(), cs.data_ptr(), work.data_ptr(), st),
                  "lfi_flow_seq_rev")
            return out, ld - ldconst
        ld = torch.zeros(N, B, dtype=torch.float32, device=self.device)
        xa = self._buf("init_xa", B * s.C)
        xb = self._buf("init_xb", B * s.C)
        for n in range(N):
            src, ldx = z_seq.data_ptr() + 4 * n * B * s.C, s.C
            for k in range(s.Ks - 1, -1, -1):
                dst = out.data_ptr() + 4 * n * B * s.C if k == 0 else (xa if (k & 1) else xb).data_ptr()
                hk = h.data_ptr() + 4 * k * B * s.H
                ck = cs.data_ptr() + 4 * k * B * s.H
                check(self.L.lfi_flow_step(C.byref(dims), C.byref(p), self.prep.data_ptr(), k, B, src, ldx,
                                           hk if n > 0 else None, ck if n > 0 else None,
                                           gic.data_ptr() + 4 * (k * F + n * B) * s.G, dst, s.C, hk, ck,
                                           ld.data_ptr() + 4 * n * B, 1, st), "lfi_flow_step")
                src = dst
        return out, ld - ldconst

    def _wc_offset(self):
        s = self.spec
        cc = s.Ks * s.C * s.C
        return 3 * cc + s.Ks * s.Ch * s.G + s.Ks * s.H * s.G + s.Ks * s.H * s.Cout

    def _ldconst_offset(self):
        s = self.spec
        return self._wc_offset() + s.Ks * s.G * s.D

    def logdet_const(self):
        return self.prep[self._ldconst_offset()]
