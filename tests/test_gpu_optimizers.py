"""GPU: every optimiser and schedule the reference's configure_optimizers / get_scheduler can build
(glow/lets_face_it_glow.py:61-72, glow/utils.py:60-82; hparam_tuning_configs/large_hparam_search.py:9-16 draws adam / sgd / rmsprop)
runs on the fused path: clip + update as ONE launch on the flat buffers, checked against torch.optim + clip_grad_norm_ fed the
engine's own gradients, step by step."""
import copy
from argparse import Namespace

import pytest
import torch

from helpers import Fixture, report

pytestmark = pytest.mark.gpu

CASES = [
    ("sgd", {"momentum": 0.9}),                                    # final_model.yaml:98-99
    ("sgd", {}),
    ("sgd", {"momentum": 0.8, "dampening": 0.1, "weight_decay": 1e-3}),
    ("sgd", {"momentum": 0.9, "nesterov": True}),
    ("rmsprop", {"eps": 1e-8}),                                    # final_model.yaml:96-97
    ("rmsprop", {"eps": 1e-6, "alpha": 0.9, "momentum": 0.5, "weight_decay": 1e-3}),
    ("rmsprop", {"eps": 1e-8, "centered": True}),
    ("adam", {"betas": [0.9, 0.9999], "eps": 1e-8}),
    # torch.optim.Adam's remaining arguments, which the reference forwards verbatim from the YAML (lets_face_it_glow.py:61-70)
    ("adam", {"betas": [0.9, 0.999], "eps": 1e-8, "weight_decay": 1e-2}),
    ("adam", {"betas": [0.9, 0.99], "eps": 1e-8, "amsgrad": True}),
    ("adam", {"betas": [0.8, 0.99], "eps": 1e-6, "weight_decay": 1e-3, "amsgrad": True}),
]


@pytest.mark.parametrize("fx_name", ["tiny", "mid"])
@pytest.mark.parametrize("name,kwargs", CASES, ids=["%s-%d" % (c[0], i) for i, c in enumerate(CASES)])
def test_fused_optimizer_matches_torch_optim(fx_name, name, kwargs, gpu_device):
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fx = Fixture(fx_name)
    hp = Namespace(**copy.deepcopy(fx.hp))
    hp.Train["use_negative_nll_loss"] = False
    hp.gradient_clip_val = 5.0
    hp.Optim["name"] = name
    hp.Optim["args"][name] = dict(kwargs)
    hp.engine_precision = "f32"
    lm = LetsFaceItGlow(hp)
    lm.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    lm.to(gpu_device)
    lm.seq_glow.glow.set_actnorm_init(True)
    lm.train()
    lm.seq_glow.injected_masks = {k: v.to(gpu_device) for k, v in fx.masks(torch.float32).items()} if fx.masks() else None
    batch = {k: v.to(gpu_device) for k, v in fx.batch(torch.float32).items()}
    eng = lm.seq_glow._ensure_engine(gpu_device)
    ref_p = torch.nn.Parameter(eng.params.detach().clone())
    cls = {"adam": torch.optim.Adam, "sgd": torch.optim.SGD, "rmsprop": torch.optim.RMSprop}[name]
    kw = dict(kwargs)
    if "betas" in kw:
        kw["betas"] = tuple(kw["betas"])
    opt = cls([ref_p], lr=3e-3, foreach=False, **kw)
    worst = 0.0
    for step in range(4):
        lm.fused_training_step(batch, 3e-3)
        # the reference side: the SAME gradient (the engine's, as it lies in the flat buffer after the step), clipped and applied by torch
        ref_p.grad = eng.grads.detach().clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
        opt.step()
        d = (eng.params - ref_p.detach()).abs().max().item()
        scale = max(1.0, ref_p.detach().abs().max().item())
        worst = max(worst, d / scale)
        # keep the two trajectories on the same parameters so that every step compares ONE update (rounding does not compound into
        # the next step's gradient)
        with torch.no_grad():
            ref_p.copy_(eng.params)
    report("fused %s %s on %s: worst parameter difference after one update, 4 steps: %.2e" % (name, kwargs, fx_name, worst))
    assert worst < 2e-6


def test_unknown_optimizer_arguments_raise(gpu_device):
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fx = Fixture("tiny")
    hp = Namespace(**copy.deepcopy(fx.hp))
    hp.Train["use_negative_nll_loss"] = False
    hp.Optim["name"] = "sgd"
    hp.Optim["args"]["sgd"] = {"momentum": 0.9, "bogus": 1}
    lm = LetsFaceItGlow(hp)
    lm.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    lm.to(gpu_device)
    lm.seq_glow.glow.set_actnorm_init(True)
    lm.train()
    batch = {k: v.to(gpu_device) for k, v in fx.batch(torch.float32).items()}
    with pytest.raises(TypeError, match="bogus"):
        lm.fused_training_step(batch, 1e-3)


def _tiny_module(gpu_device, name, kwargs):
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fx = Fixture("tiny")
    hp = Namespace(**copy.deepcopy(fx.hp))
    hp.Train["use_negative_nll_loss"] = False
    hp.gradient_clip_val = 5.0
    hp.Optim["name"] = name
    hp.Optim["args"][name] = dict(kwargs)
    hp.engine_precision = "f32"
    lm = LetsFaceItGlow(hp)
    lm.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    lm.to(gpu_device)
    lm.seq_glow.glow.set_actnorm_init(True)
    lm.train()
    lm.seq_glow.injected_masks = {k: v.to(gpu_device) for k, v in fx.masks(torch.float32).items()} if fx.masks() else None
    return lm, {k: v.to(gpu_device) for k, v in fx.batch(torch.float32).items()}


def test_resumed_sgd_keeps_its_imported_momentum_buffer(gpu_device):
    """torch.optim.SGD's "first step" (buf = g) means "this parameter has no momentum buffer yet"; the fused step keyed
    it on its own step count, so a state imported from a reference checkpoint (a buffer, no step count) was overwritten with g on the
    first resumed step. One resumed step against torch.optim.SGD holding the same buffer."""
    lm, batch = _tiny_module(gpu_device, "sgd", {"momentum": 0.9})
    eng = lm.seq_glow._ensure_engine(gpu_device)
    buf = torch.randn_like(eng.params) * 0.1
    eng.load_optimizer_state({"step_count": 0, "optimizer": "sgd", "momentum_inited": True, "adam_m": buf.clone(), "adam_v": None,
                              "opt_aux": None})
    ref_p = torch.nn.Parameter(eng.params.detach().clone())
    opt = torch.optim.SGD([ref_p], lr=3e-3, momentum=0.9, foreach=False)
    opt.state[ref_p]["momentum_buffer"] = buf.clone()
    lm.fused_training_step(batch, 3e-3)
    ref_p.grad = eng.grads.detach().clone()
    torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
    opt.step()
    d = (eng.params - ref_p.detach()).abs().max().item()
    assert d < 2e-6, d
    assert (eng.adam_m - opt.state[ref_p]["momentum_buffer"]).abs().max().item() < 2e-6


def test_optimizer_state_of_another_optimizer_is_refused(gpu_device):
    """native checkpoints carried no optimiser tag, so resuming with another Optim.name silently reused Adam's moments as
    a momentum buffer / square average. The state names its optimiser; another one refuses to step it."""
    lm, batch = _tiny_module(gpu_device, "adam", {"betas": [0.9, 0.999], "eps": 1e-8})
    lm.fused_training_step(batch, 1e-3)
    eng = lm.seq_glow._ensure_engine(gpu_device)
    state = eng.optimizer_state()
    assert state["optimizer"] == "adam"
    lm2, batch2 = _tiny_module(gpu_device, "sgd", {"momentum": 0.9})
    eng2 = lm2.seq_glow._ensure_engine(gpu_device)
    eng2.load_optimizer_state(state)
    with pytest.raises(ValueError, match="adam"):
        lm2.fused_training_step(batch2, 1e-3)
