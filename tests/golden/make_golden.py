"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Runs only in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

It imports ``glow_pytorch.glow`` unmodified from /root/reference/code through
five stub modules, runs the reference in fp64 on seeded
inputs, asserts that ``oracle/seqglow_oracle.py`` reproduces every output to
1e-10, and writes inputs + expected outputs as ``*.npz`` (data only).
"""
import argparse
import copy
import json
import os
import sys
import types

sys.dont_write_bytecode = True   # importing /root/reference must not leave __pycache__ behind there (it is read-only by contract)

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"


def import_reference():
    for n in ["jsmin", "toml", "pytorch_lightning", "optuna", "h5py"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["jsmin"].jsmin = lambda s: s
    pl = sys.modules["pytorch_lightning"]
    pl.Trainer = type("Trainer", (), {"add_argparse_args": staticmethod(lambda p: p)})
    pl.LightningModule = torch.nn.Module
    pl.LightningDataModule = object
    pl.seed_everything = lambda s: None
    # `misc` stays a namespace whose `shared` is a stub (the real one reads config.toml through `toml`), while
    # `misc.utils` (get_face_indicies, misc/utils.py:38-43) is the reference's own file
    misc = types.ModuleType("misc")
    misc.__path__ = [os.path.join(REF, "misc")]
    sys.modules["misc"] = misc
    ms = types.ModuleType("misc.shared")
    ms.DATA_DIR = "/data"
    ms.BASE_DIR = "/data"
    ms.RANDOM_SEED = 1234
    sys.modules["misc.shared"] = ms
    misc.shared = ms
    # modules generate_motion_from_model.py imports that do not exist in the reference tree itself (SURVEY.md par. 2 #11)
    for n, attrs in (("data_segments", ()), ("data_segments.find_test_segments", ("get_frames",)), ("visualize", ()),
                     ("visualize.faces", ("get_vert", "render_double_face_video"))):
        if n not in sys.modules:
            mod = types.ModuleType(n)
            for a in attrs:
                setattr(mod, a, None)
            sys.modules[n] = mod
    sys.path.insert(0, REF)
    from glow_pytorch.glow import models, modules, utils  # noqa
    return models, modules, utils


def base_hparams():
    with open(os.path.join(REF, "glow_pytorch/hparams/final_model.yaml")) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def make_config(name):
    hp = base_hparams()
    c, g = hp["Conditioning"], hp["Glow"]
    if name in ("tiny", "tiny_lstm", "tiny_additive", "framenb", "dense"):
        # BASELINE.json configs[0]: 2-step/1-level, 16-d, seq_len 20, batch 4
        c["cond_dim"] = 32
        c["p1_face"].update(history=2, dim=16)
        c["p1_speech"].update(history=2, hidden_dim=16)
        c["p2_face"].update(history=4, dim=16, hidden_dim=24)
        c["p2_speech"].update(history=4, hidden_dim=24)
        hp["Data"]["speech_dim"] = 8
        g.update(K=2, L=1, hidden_channels=32)
        hp["Train"]["seq_len"] = 20
        dims = dict(B=4, T=20)
        if name == "tiny_lstm":
            g["rnn_type"] = "lstm"
        if name == "tiny_additive":
            g["flow_coupling"] = "additive"
        if name == "dense":  # InvertibleConv1x1 with a dense weight: slogdet / inverse branch (modules.py:151-161)
            g["LU_decomposed"] = False
        if name == "framenb":  # Conditioning.use_frame_nb: one frame-counter column appended to the features
            c["use_frame_nb"] = True
    elif name == "odd":
        # odd channel count (C=15 -> z1 7, z2 8, coupling output C+1), L=2, one modality switched off
        c["cond_dim"] = 40
        c["p1_face"].update(history=3, dim=15)
        c["p1_speech"].update(history=0, hidden_dim=16)
        c["p2_face"].update(history=5, dim=15, hidden_dim=20)
        c["p2_speech"].update(history=6, hidden_dim=12)
        hp["Data"]["speech_dim"] = 7
        g.update(K=2, L=2, hidden_channels=24)
        hp["Train"]["seq_len"] = 14
        dims = dict(B=6, T=14)
    elif name == "mid":
        # final_model.yaml histories/dropouts at BASELINE's synthetic dims (C=50, S=27); K=3, short T and
        # narrow hidden widths so the fixture stays small (full widths are checked against the oracle on the GPU)
        c["cond_dim"] = 48
        c["p1_face"]["dim"] = 50
        c["p2_face"].update(dim=50, hidden_dim=40)
        c["p1_speech"]["hidden_dim"] = 20
        c["p2_speech"]["hidden_dim"] = 36
        hp["Data"]["speech_dim"] = 27
        g.update(K=3, hidden_channels=44)
        hp["Train"]["seq_len"] = 28
        dims = dict(B=8, T=28)
    elif name in ("mlp", "p1enc"):
        # the encoder types the reference's hparam search draws for EVERY modality, p1_face included
        # (hparam_tuning_configs/large_hparam_search.py:45-72): "mlp", a raw window with dropout, an encoded p1_face window
        c["cond_dim"] = 32
        hp["Data"]["speech_dim"] = 8
        g.update(K=2, L=1, hidden_channels=32)
        hp["Train"]["seq_len"] = 20
        dims = dict(B=4, T=20)
        if name == "mlp":
            c["p1_face"].update(history=3, dim=16)
            c["p2_face"].update(history=4, dim=16, hidden_dim=24, enc="mlp")
            c["p1_speech"].update(history=2, hidden_dim=16, enc="none")          # raw window, dropout 0.5 kept
            c["p2_speech"].update(history=4, hidden_dim=20, enc="mlp")
        else:
            c["p1_face"].update(history=3, dim=16, hidden_dim=12, enc="rnn", dropout=0.4)
            c["p2_face"].update(history=4, dim=16, hidden_dim=24)
            c["p1_speech"].update(history=2, hidden_dim=16, enc="mlp")
            c["p2_speech"].update(history=3, hidden_dim=20, enc="none")
    elif name == "lstmenc":
        # "enc: lstm" window encoders (ModalityEncoder's nn.LSTM branch, models.py:27-33,65-69), with dropout on one of them
        c["cond_dim"] = 32
        hp["Data"]["speech_dim"] = 8
        g.update(K=2, L=1, hidden_channels=32)
        hp["Train"]["seq_len"] = 20
        dims = dict(B=4, T=20)
        c["p1_face"].update(history=3, dim=16)
        c["p2_face"].update(history=5, dim=16, hidden_dim=24, enc="lstm")
        c["p1_speech"].update(history=2, hidden_dim=16)
        c["p2_speech"].update(history=4, hidden_dim=20, enc="lstm")
    elif name == "p1lstm":
        # an LSTM-encoded prev_p1_face window: re-encoded for every generated frame in SeqGlow.inference
        c["cond_dim"] = 32
        hp["Data"]["speech_dim"] = 8
        g.update(K=2, L=1, hidden_channels=32)
        hp["Train"]["seq_len"] = 20
        dims = dict(B=4, T=20)
        c["p1_face"].update(history=4, dim=16, hidden_dim=12, enc="lstm", dropout=0.3)
        c["p2_face"].update(history=4, dim=16, hidden_dim=24)
        c["p1_speech"].update(history=2, hidden_dim=16, enc="lstm")
        c["p2_speech"].update(history=3, hidden_dim=20, enc="none")
    elif name == "p1mlp":
        c["cond_dim"] = 32
        hp["Data"]["speech_dim"] = 8
        g.update(K=2, L=1, hidden_channels=32)
        hp["Train"]["seq_len"] = 20
        dims = dict(B=4, T=20)
        c["p1_face"].update(history=3, dim=16, hidden_dim=20, enc="mlp", dropout=0.3)
        c["p2_face"].update(history=4, dim=16, hidden_dim=24)
        c["p1_speech"].update(history=0, hidden_dim=16)
        c["p2_speech"].update(history=4, hidden_dim=24)
    else:
        raise KeyError(name)
    dims["C"] = c["p1_face"]["dim"]
    dims["S"] = hp["Data"]["speech_dim"]
    return hp, dims


class MaskFeed(torch.nn.Module):
    """Stands in for nn.Dropout inside the reference's ModalityEncoder: replays preset masks."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0

    def forward(self, ones):
        m = self.masks[self.i].to(ones.dtype)
        self.i += 1
        assert m.shape == ones.shape
        return m


def clean(model):
    """Drop the non-leaf tensors the reference caches on its modules (f_seq.hidden, FlowStep.scale)."""
    model.glow.init_rnn_hidden()
    for mod in model.modules():
        if mod.__class__.__name__ == "FlowStep":
            mod.scale = None
    return model


def to_double(model):
    m = copy.deepcopy(clean(model)).double()
    for mod in m.modules():
        if mod.__class__.__name__ == "InvertibleConv1x1" and mod.LU:
            mod.l_mask = mod.l_mask.double()
            mod.eye = mod.eye.double()
    return m


class NoFloatCast:
    """Neutralise the .float() casts (models.py:321, modules.py:175-176) while running the fp64 reference."""

    def __enter__(self):
        self.orig = torch.Tensor.float
        torch.Tensor.float = lambda t, *a, **k: t

    def __exit__(self, *a):
        torch.Tensor.float = self.orig


def perturb(model, gen):
    """SURVEY.md finding 5: LinearZeros is zero at init, so the conditioning path is dead until perturbed."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "final_linear" in name:
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)
            elif "actnorm" in name:
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith("invconv.log_s"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)
            elif name.endswith("invconv.weight"):  # dense mode: off the orthogonal init, so that log|det W| != 0
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)


def sd_numpy(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def close(a, b, tol, what):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
    assert err < tol, "%s: oracle vs reference rel err %.3e" % (what, err)
    return err


def lstm_shim(models):
    """Reference LSTM branch crashes on (None, None) (models.py:209-213); pass None on the first step instead."""
    def fwd(self, z, condition):
        x = torch.cat((z, self.cond_transform(condition)), dim=1)
        if self.rnn_type == "gru":
            self.hidden = self.rnn(x, self.hidden)
        else:
            self.hidden, self.cell = self.rnn(x, None if self.hidden is None else (self.hidden, self.cell))
        return self.final_linear(self.hidden)
    models.f_seq.forward = fwd


def build(name, models, modules, oracle):
    hp, dims = make_config(name)
    B, T, C, S = dims["B"], dims["T"], dims["C"], dims["S"]
    ns = argparse.Namespace(**copy.deepcopy(hp))
    torch.manual_seed(1234)
    np.random.seed(1234)
    model = models.SeqGlow(ns)
    gen = torch.Generator().manual_seed(4321)
    perturb(model, gen)
    model.glow.set_actnorm_init(True)
    m64 = to_double(model)
    sd32 = {k: v.clone() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd32.items()}
    batch32 = oracle.synthetic_batch(B, T, C, S, seed=1234)
    use_nb = bool(hp["Conditioning"]["use_frame_nb"])

    def with_frame_nb(b, first):
        # MimicryDataset would supply the index of the window's first frame; any (B, 1) float works for parity. Kept small
        # (a real counter of several thousand saturates LeakyReLU(cond_transform) and hides every other column)
        if use_nb:
            b["frame_nb"] = (torch.arange(b["p1_face"].shape[0], dtype=torch.float32).view(-1, 1) * 0.25 + first) * 0.1
        return b

    batch32 = with_frame_nb(batch32, 1.0)
    batch64 = {k: v.double() for k, v in batch32.items()}
    start = oracle.longest_history(hp["Conditioning"])
    N = T - start
    out = {"hparams_json": np.array(json.dumps(hp)), "dims": np.array([B, T, C, S, N, start])}
    for k, v in sd_numpy(sd32).items():
        out["sd/" + k] = v
    for k, v in batch32.items():
        out["batch/" + k] = v.numpy()

    # ---- 1. eval-mode forward (dropout off)
    m64.eval()
    with NoFloatCast(), torch.no_grad():
        z_seq, loss, losses = m64(batch64)
    z_ref, l_ref = torch.stack(z_seq), torch.stack(losses)
    z_o, loss_o, l_o = oracle.seqglow_forward(hp, sd64, batch64)
    close(z_o, z_ref, 1e-10, name + " eval z")
    close(l_o, l_ref, 1e-10, name + " eval nll")
    close(loss_o, loss, 1e-10, name + " eval loss")
    out.update({"eval/z": z_ref.numpy(), "eval/nll": l_ref.numpy(), "eval/loss": loss.numpy()})
    # the fp32 reference itself, as the "reference noise floor"
    model.eval()
    with torch.no_grad():
        _, loss32, losses32 = model(batch32)
    out["eval/nll_ref_fp32"] = torch.stack(losses32).numpy()

    # ---- 2. train-mode forward + backward with injected dropout masks
    masks = {}
    for mname in ("p1_face", "p2_face", "p1_speech", "p2_speech"):
        cfg = hp["Conditioning"][mname]
        if cfg["history"] and cfg["dropout"] > 0:
            keep = 1.0 - cfg["dropout"]
            mk = (torch.rand(N, B, cfg["history"], generator=gen) < keep).double() / keep
            masks[mname] = mk
            out["mask/" + mname] = mk.numpy()
    m64.train()
    for mname, mk in masks.items():
        getattr(m64.feature_encoder, mname + "_encoder").dropout = MaskFeed(list(mk))
    m64.zero_grad()
    with NoFloatCast():
        z_seq, loss, losses = m64(batch64)
        loss.sum().backward()
    close(oracle.seqglow_forward(hp, sd64, batch64, masks)[2], torch.stack(losses), 1e-10, name + " train nll")
    out.update({"train/nll": torch.stack(losses).detach().numpy(), "train/loss": loss.detach().numpy(),
                "train/z": torch.stack(z_seq).numpy()})
    sdg = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
           for k, v in sd64.items()}
    oracle.seqglow_forward(hp, sdg, batch64, masks)[1].sum().backward()
    for pname, p in m64.named_parameters():
        close(sdg[pname].grad, p.grad, 1e-9, name + " grad " + pname)
        out["grad/" + pname] = p.grad.numpy()

    # ---- 3. one optimiser step on those gradients (Adam + clip, final_model.yaml)
    opt_hp = hp["Optim"]["args"]["adam"]
    m_step = copy.deepcopy(clean(m64))
    opt = torch.optim.Adam(m_step.parameters(), lr=1e-3, betas=tuple(opt_hp["betas"]), eps=opt_hp["eps"])
    for p_new, p_old in zip(m_step.parameters(), m64.parameters()):
        p_new.grad = p_old.grad.clone()
    gn = torch.nn.utils.clip_grad_norm_(m_step.parameters(), 0.5)
    opt.step()
    out["adam/lr"], out["adam/clip"], out["adam/grad_norm"] = np.array(1e-3), np.array(0.5), gn.numpy()
    for pname, p in m_step.named_parameters():
        out["adam/" + pname] = p.detach().numpy()
    names = [n_ for n_, _ in m64.named_parameters()]
    ps = [sd64[n_].clone() for n_ in names]
    gs = [p.grad.clone() for p in m64.parameters()]
    ms_, vs_ = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    gno = oracle.adam_clip_step(ps, gs, ms_, vs_, 1, 1e-3, opt_hp["betas"][0], opt_hp["betas"][1], opt_hp["eps"], 0.5)
    close(gno, gn, 1e-10, name + " grad norm")
    for n_, p in zip(names, ps):
        close(p, dict(m_step.named_parameters())[n_], 1e-10, name + " adam " + n_)

    # ---- 4. negative-example step with a fixed permutation (lets_face_it_glow.py:40-50)
    perm = torch.randperm(B, generator=gen)
    mixed = dict(batch64)
    for mname in ("p2_face", "p2_speech"):
        if hp["Conditioning"][mname]["history"] > 0:
            mixed[mname] = batch64[mname][perm]
    m64.eval()
    with NoFloatCast(), torch.no_grad():
        _, nloss, nlosses = m64(mixed)
    close(oracle.training_loss(hp, sd64, batch64, None, perm), nloss * -0.1, 1e-10, name + " negative step")
    out.update({"neg/perm": perm.numpy(), "neg/nll": torch.stack(nlosses).numpy(), "neg/loss": (nloss * -0.1).numpy()})

    # ---- 5. sampling with injected prior noise (models.py:567-596)
    seq_len = T + 6
    eps_std = 0.7
    long_batch = with_frame_nb(oracle.synthetic_batch(B, seq_len, C, S, seed=99), 3.0)
    data32 = {k: v for k, v in long_batch.items()}
    data32["p1_face"] = data32["p1_face"][:, :start].clone()
    data64 = {k: v.double() for k, v in data32.items()}
    noise = torch.randn(seq_len - start, B, C, generator=gen).double() * eps_std
    feed = list(noise)
    orig_sample = modules.GaussianDiag.sample
    modules.GaussianDiag.sample = staticmethod(lambda shape, eps_std=1: feed.pop(0).to(shape.dtype))
    m64.hparams.Infer["eps"] = eps_std
    with NoFloatCast(), torch.no_grad():
        samp = m64.inference(seq_len, data64)
    modules.GaussianDiag.sample = orig_sample
    close(oracle.seqglow_inference(hp, sd64, seq_len, data64, noise), samp, 1e-10, name + " inference")
    # the fp32 reference itself on the same injected noise: its distance from the fp64 result is the floor any fp32
    # implementation of this autoregressive chain sits on (SURVEY.md par. 7, "hard parts")
    feed32 = list(noise.float())
    modules.GaussianDiag.sample = staticmethod(lambda shape, eps_std=1: feed32.pop(0))
    model.hparams.Infer["eps"] = eps_std
    model.eval()
    clean(model)
    with torch.no_grad():
        samp32 = model.inference(seq_len, data32)
    modules.GaussianDiag.sample = orig_sample
    out["infer/out_ref_fp32"] = samp32.numpy()
    for k, v in data32.items():
        out["infer/data/" + k] = v.numpy()
    out.update({"infer/noise": noise.numpy(), "infer/out": samp.numpy(), "infer/seq_len": np.array(seq_len)})

    # ---- 6. invert (models.py:617-645)
    # (the reference's invert() fails for additive coupling: its log-det stays a 0-dim tensor and the
    #  in-place `objective += ...` of models.py:564 cannot broadcast; no fixture for that case)
    if hp["Glow"]["flow_coupling"] == "affine":
        with NoFloatCast(), torch.no_grad():
            rec, bl = m64.invert(list(z_ref), batch64)
        rec_o, bl_o = oracle.seqglow_invert(hp, sd64, z_ref, batch64)
        close(rec_o, torch.stack(rec), 1e-10, name + " invert x")
        close(bl_o, bl, 1e-10, name + " invert loss")
        out.update({"invert/x": torch.stack(rec).numpy(), "invert/loss": bl.numpy()})

    # ---- 7. ActNorm data-dependent init on a larger batch (modules.py:32-43)
    Bi = 32
    init_batch = {k: v.double() for k, v in with_frame_nb(oracle.synthetic_batch(Bi, T, C, S, seed=7), 2.0).items()}
    m_init = to_double(model)
    m_init.glow.set_actnorm_init(False)
    m_init.train()
    init_masks = {}
    for mname, mk in masks.items():
        cfg = hp["Conditioning"][mname]
        keep = 1.0 - cfg["dropout"]
        imk = (torch.rand(N, Bi, cfg["history"], generator=gen) < keep).double() / keep
        init_masks[mname] = imk
        getattr(m_init.feature_encoder, mname + "_encoder").dropout = MaskFeed(list(imk))
        out["init/mask/" + mname] = imk.numpy()
    with NoFloatCast(), torch.no_grad():
        _, _, init_losses = m_init(init_batch)
    sd_init = oracle.actnorm_init(hp, sd64, init_batch, init_masks)
    for k in range(oracle.n_flow_steps(hp)):
        for leaf in ("bias", "logs"):
            key = "glow.flow.layers.%d.actnorm.%s" % (k, leaf)
            close(sd_init[key], m_init.state_dict()[key], 1e-10, name + " init " + key)
            out["init/" + key] = m_init.state_dict()[key].numpy()
    close(oracle.seqglow_forward(hp, sd_init, init_batch, init_masks)[2], torch.stack(init_losses), 1e-10,
          name + " init nll")
    out["init/nll"] = torch.stack(init_losses).numpy()
    for k, v in init_batch.items():
        out["init/batch/" + k] = v.float().numpy()
    return out


class DictFile(dict):
    """What MimicryDataset needs of h5py.File (mimicry_data_module.py:33,48): `File(name, "r")[split][kind][key][who][rows]`,
    `.items()`, `len()`, use as a context manager. Backed by nested dicts of numpy arrays (`rows` is a list of ints: numpy
    fancy indexing, which is what h5py does for an increasing index list)."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def build_mimicry(utils):
    """SURVEY.md par. 8(f): the reference's MimicryDataset (mimicry_data_module.py:12-81), calc_jerk (glow/utils.py:53-58),
    get_face_indicies (misc/utils.py:38-43), dictify_frames / expand_face_dim (generate_motion_from_model.py:39-51,73-87),
    run on a small seeded corpus; the numpy restatement oracle/mimicry_oracle.py is asserted equal before anything is written."""
    import random
    from oracle import mimicry_oracle as mo
    import glow_pytorch.mimicry_data_module as ref_dm
    import glow_pytorch.generate_motion_from_model as ref_gm
    from misc.utils import get_face_indicies

    data_hp = {"expression_dim": 5, "jaw_dim": 3, "neck_dim": 3, "speech_dim": 6, "file_name": "unused"}
    cond_hp = {"p1_face": {"history": 2}, "p2_face": {"history": 3}, "p1_speech": {"history": 2}, "p2_speech": {"history": 0},
               "use_frame_nb": False}
    seq_len, seed = 5, 1234
    rng = np.random.RandomState(0)
    dims = {"flame_expression": 8, "flame_jaw": 3, "flame_neck": 3, "mfcc": 4, "prosody": 2}
    store = {"train": {k: {} for k in dims}}
    for i, n in enumerate((7, 3, 12, 5, 40)):
        for kind, d in dims.items():
            store["train"][kind][str(i)] = {who: rng.randn(n, d).astype(np.float32) for who in ("agent", "interlocutor")}
    out = {"hparams_json": np.array(json.dumps({"Data": data_hp, "Conditioning": cond_hp})), "seq_len": np.array(seq_len),
           "seed": np.array(seed)}
    for kind, bins in store["train"].items():
        for key, d in bins.items():
            for who, a in d.items():
                out["store/train/%s/%s/%s" % (kind, key, who)] = a

    sys.modules["h5py"].File = lambda name, mode="r": DictFile(store)
    random.seed(seed)
    ds = ref_dm.MimicryDataset("unused", "train", data_hparams=data_hp, conditioning_hparams=cond_hp, seq_len=seq_len)
    keys = [k for k, _ in ds.indicies]
    starts = [rows[0] for _, rows in ds.indicies]
    assert all(rows == list(range(rows[0], rows[0] + seq_len)) for _, rows in ds.indicies)
    # oracle: same enumeration, then the reference's one-off shuffle (random.sample of the whole list, :43)
    enum = mo.window_index(store, "train", seq_len)
    random.seed(seed)
    assert random.sample(enum, len(enum)) == list(zip(keys, starts))
    out["index/keys"], out["index/starts"] = np.array(keys), np.array(starts, dtype=np.int64)
    hist = {m: cond_hp[m]["history"] for m in ("p1_speech", "p2_speech", "p2_face")}
    for i in range(len(ds)):
        item = ds[i]
        want = mo.get_item(store, "train", keys[i], starts[i], seq_len, data_hp["expression_dim"], hist)
        assert set(item) == set(want)
        for name, t in item.items():
            assert t.dtype == torch.float32 and np.array_equal(t.numpy(), want[name]), (i, name)
            out["item/%d/%s" % (i, name)] = t.numpy()

    g = torch.Generator().manual_seed(5)
    for j, shape in enumerate(((3, 7, 5), (16, 76, 50), (1, 4, 1))):
        x = torch.randn(*shape, generator=g)
        jr = utils.calc_jerk(x)
        assert abs(mo.calc_jerk(x.numpy()) - float(jr)) < 1e-6 * max(1.0, float(jr))
        out["jerk/%d/x" % j], out["jerk/%d/out" % j] = x.numpy(), jr.numpy()

    for j, args in enumerate(((50, 3, 3, 0), (50, 3, 3, 136), (5, 3, 3, 0), (100, 3, 3, 136))):
        idx = get_face_indicies(*args[:3], offset=args[3])
        assert idx == mo.get_face_indicies(*args[:3], offset=args[3])
        out["face_idx/%d/args" % j], out["face_idx/%d/out" % j] = np.array(args), np.array(idx)

    gm_hp = {"expression_dim": 50, "jaw_dim": 3, "neck_dim": 3, "speech_dim": 30}
    frames = torch.randn(6, 272, generator=g)
    d = ref_gm.dictify_frames(frames, gm_hp)
    d_o = mo.dictify_frames(frames.numpy(), gm_hp)
    out["gm/hparams_json"], out["gm/frames"] = np.array(json.dumps(gm_hp)), frames.numpy()
    for k, v in d.items():
        assert np.array_equal(v.numpy(), d_o[k]), k
        out["gm/dictify/" + k] = v.numpy()
    seq = torch.randn(2, 6, 56, generator=g)
    ex = ref_gm.expand_face_dim(seq, gm_hp)
    assert np.array_equal(ex.numpy(), mo.expand_face_dim(seq.numpy(), gm_hp))
    out["gm/expand/seq"], out["gm/expand/out"] = seq.numpy(), ex.numpy()
    return out


def main():
    sys.path.insert(0, ROOT)
    from oracle import seqglow_oracle as oracle
    models, modules, utils = import_reference()
    lstm_shim(models)
    names = sys.argv[1:] or ("tiny", "tiny_lstm", "tiny_additive", "odd", "mid", "mlp", "p1enc", "p1mlp", "framenb", "lstmenc", "p1lstm",
                             "dense", "mimicry")
    for name in names:
        out = build_mimicry(utils) if name == "mimicry" else build(name, models, modules, oracle)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
