"""CPU: the training loop's host logic with a recording stand-in for the model (no kernels run) - mid-epoch resume keeps the
epoch's batch order (trainer.py), the validation callbacks leave Python's global RNG alone (ADVICE r3,
mimicry_logger.py), backward-product settings are validated."""
import random
from argparse import Namespace

import pytest
import torch

from lets_face_it_amd.mimicry_data_module import WindowLoader
from lets_face_it_amd.trainer import Trainer


class _Windows:
    """40 'windows': a batch is the float tensor of its window numbers (what the recording model logs)."""

    def __len__(self):
        return 40

    def batch(self, index):
        idx = torch.as_tensor(index, dtype=torch.int64)
        return {"p1_face": idx.float().reshape(-1, 1, 1).repeat(1, 30, 1)}


class _Data:
    def __init__(self, shuffle=True):
        self.shuffle = shuffle

    def train_dataloader(self):
        return WindowLoader(_Windows(), 8, shuffle=self.shuffle)


class _Recorder(torch.nn.Module):
    """Stands in for LetsFaceItGlow: records the window numbers of every step and draws from torch's global generator the way
    derange_batch does on negative steps (so the stream the loader shuffles from moves between steps)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))
        glow = Namespace(actnorm_inited=lambda: True, set_actnorm_init=lambda flag: None)
        self.seq_glow = Namespace(spec=Namespace(start=24), engine=None, glow=glow, allreduce_hook=None,
                                  _ensure_engine=lambda device: Namespace(load_optimizer_state=lambda st: None))
        self.seen, self.draws = [], []

    def fused_training_step(self, batch, lr, world, allreduce):
        self.seen.append(batch["p1_face"][:, 0, 0].long().tolist())
        self.draws.append(float(torch.rand(1)))
        with torch.no_grad():
            self.w += 1
        return torch.zeros(())


def _hparams(max_steps=None, max_epochs=2):
    return Namespace(lr=1e-3, Optim={"Schedule": {"name": None}}, max_epochs=max_epochs, max_steps=max_steps,
                     checkpoint_callback=True)


@pytest.mark.parametrize("stop_at", [2, 5, 7])
def test_mid_epoch_resume_continues_the_same_batch_order(tmp_path, stop_at):
    """A run stopped by max_steps inside an epoch and resumed from its checkpoint sees exactly the batches (and the RNG
    draws) an uninterrupted run sees: the epoch's permutation is re-drawn from the epoch-start generator state kept in the
    checkpoint, not from the checkpoint-time state."""
    torch.manual_seed(7)
    ref = _Recorder()
    Trainer(_hparams(), device="cpu", log_every=10 ** 9, checkpoint_dir="").fit(ref, _Data())
    assert len(ref.seen) == 10 and sorted(sum(ref.seen[:5], [])) == list(range(40))

    torch.manual_seed(7)
    a = _Recorder()
    Trainer(_hparams(max_steps=stop_at), device="cpu", log_every=10 ** 9, checkpoint_dir=str(tmp_path)).fit(a, _Data())
    assert a.seen == ref.seen[:stop_at]

    torch.manual_seed(999)      # a fresh process: whatever the generator holds is replaced by the checkpoint's
    b = _Recorder()
    tr = Trainer(_hparams(), device="cpu", log_every=10 ** 9, checkpoint_dir="")
    tr.resume(b, str(tmp_path / "last.ckpt"))
    assert tr.global_step == stop_at and tr.batches_into_epoch == (stop_at if stop_at <= 5 else stop_at - 5)
    tr.fit(b, _Data())
    assert a.seen + b.seen == ref.seen
    assert a.draws + b.draws == ref.draws
    assert float(b.w.detach()[0]) == 10.0    # weights came back with the checkpoint and were stepped 10 - stop_at more times


def test_window_loader_skip_batches_starts_inside_the_order():
    torch.manual_seed(3)
    full = [b["p1_face"][:, 0, 0].long().tolist() for b in WindowLoader(_Windows(), 8)]
    torch.manual_seed(3)
    ld = WindowLoader(_Windows(), 8)
    ld.skip_batches = 2
    assert [b["p1_face"][:, 0, 0].long().tolist() for b in ld] == full[2:]
    assert ld.skip_batches == 0      # one-shot


def test_logger_render_draws_do_not_move_the_global_python_rng(monkeypatch):
    """Under data parallelism the callbacks run on rank 0 only; the negative-example branch of every rank reads Python's global
    `random`, so the logger must not draw from it."""
    from lets_face_it_amd import mimicry_logger as ml
    monkeypatch.setattr(ml, "calc_jerk", lambda x: x.abs().mean())
    rendered = []
    B, T, Cc = 6, 40, 4
    batch = {"p1_face": torch.randn(B, T, Cc), "p2_face": torch.randn(B, T, Cc)}

    class SG:
        def __call__(self, b):
            return [torch.zeros(B, Cc)] * 5, torch.ones(1), None

        def inference(self, seq_len, data):
            return torch.zeros(B, 10, Cc)

        def invert(self, z_seq, data):
            return [torch.zeros(B, Cc)] * 5, -torch.ones(1)

    hp = Namespace(Validation={"inference": True, "seq_len": 34, "render": True, "check_invertion": True,
                               "scale_logging": False, "wrong_context_test": False},
                   Conditioning={m: {"history": 24} for m in ("p1_face", "p2_face", "p1_speech", "p2_speech")})
    pl = Namespace(hparams=hp, seq_glow=SG(), log=lambda *a, **k: None)
    random.seed(11)
    before = random.getstate()
    logger = ml.MimicryLogger(render_hook=lambda name, a, b, m: rendered.append(name), seed=5)
    logger.on_validation_batch_end(None, pl, None, batch, 0)
    assert rendered == ["video", "test_reconstr"]
    assert random.getstate() == before


def test_backward_products_setting_is_validated():
    from lets_face_it_amd.engine import GlowEngine
    assert GlowEngine.check_backward_products("auto") == "auto"
    assert GlowEngine.check_backward_products("2") == 2 and GlowEngine.check_backward_products(3) == 3
    for bad in (1, "1", 4, "two", None, 2.5):
        with pytest.raises(ValueError):
            GlowEngine.check_backward_products(bad)


def test_sampler_run_split_covers_every_frame_once(monkeypatch):
    """GlowEngine.sample cuts the generated frames into runs (static part of run i + 1 on the second stream under run i's chain):
    the runs tile [0, nframes) in order, differ by at most one frame, and short sequences stay one run."""
    from lets_face_it_amd.engine import GlowEngine
    monkeypatch.delenv("LFI_SAMPLE_RUNS", raising=False)
    for n in (1, 5, 63, 64, 276, 277, 1000):
        runs = GlowEngine._sample_runs(n)
        assert runs[0][0] == 0 and sum(c for _, c in runs) == n
        assert all(runs[i][0] + runs[i][1] == runs[i + 1][0] for i in range(len(runs) - 1))
        assert max(c for _, c in runs) - min(c for _, c in runs) <= 1
        assert len(runs) == (4 if n >= 64 else 1)
    monkeypatch.setenv("LFI_SAMPLE_RUNS", "3")
    assert [c for _, c in GlowEngine._sample_runs(10)] == [4, 3, 3]
    monkeypatch.setenv("LFI_SAMPLE_RUNS", "9")
    assert len(GlowEngine._sample_runs(4)) == 4      # never more runs than frames


def test_lr_at_reproduces_every_reference_schedule():
    """Trainer.lr_at in closed form == the torch schedulers get_scheduler builds (glow/utils.py:60-82), stepped once per epoch as
    Lightning does: StepLR, LambdaLR(e // val), MultiplicativeLR(e // val), and no schedule."""
    import copy
    from argparse import Namespace
    from torch.optim.lr_scheduler import LambdaLR, MultiplicativeLR, StepLR
    from helpers import Fixture
    from lets_face_it_amd.trainer import Trainer
    base = Fixture("tiny").hp
    for name, args in (("step", {"gamma": 0.73, "step_size": 3}), ("lambda", {"val": 4}), ("multiplicative", {"val": 1}),
                       ("multiplicative", {"val": 3}), (None, None)):
        hp = copy.deepcopy(base)
        hp["lr"] = 2e-4
        hp["Optim"]["Schedule"]["name"] = name
        if name:
            hp["Optim"]["Schedule"]["args"][name] = args
        tr = Trainer(Namespace(**hp), device="cpu", checkpoint_dir="")
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=2e-4)
        sched = {"step": lambda: StepLR(opt, **args), "lambda": lambda: LambdaLR(opt, lr_lambda=[lambda e: e // args["val"]]),
                 "multiplicative": lambda: MultiplicativeLR(opt, lr_lambda=[lambda e: e // args["val"]]), None: lambda: None}[name]()
        for epoch in range(12):
            want = opt.param_groups[0]["lr"]
            assert abs(tr.lr_at(epoch) - want) <= 1e-12 * max(1.0, abs(want)) + 1e-18, (name, args, epoch, tr.lr_at(epoch), want)
            opt.step()
            if sched is not None:
                sched.step()
    hp = copy.deepcopy(base)
    hp["Optim"]["Schedule"]["name"] = "cosine"
    import pytest
    with pytest.raises(NotImplementedError):
        Trainer(Namespace(**hp), device="cpu", checkpoint_dir="").lr_at(0)


@pytest.mark.parametrize("name,kwargs", [("adam", {}), ("adam", {"amsgrad": True}), ("sgd", {"momentum": 0.9}), ("sgd", {}),
                                         ("rmsprop", {}), ("rmsprop", {"momentum": 0.5, "centered": True})])
def test_reference_optimizer_state_maps_onto_the_flat_buffers(name, kwargs):
    """a Lightning / reference checkpoint's `optimizer_states[0]` is whatever configure_optimizers built
    (glow/lets_face_it_glow.py:61-72: Adam, SGD or RMSprop) - their per-parameter state keys differ (exp_avg / exp_avg_sq /
    max_exp_avg_sq, momentum_buffer, square_avg / momentum_buffer / grad_avg) and SGD keeps no step count. Each lands in the flat
    buffer the fused step of the SAME optimiser reads it from, tagged with the optimiser's name."""
    torch.manual_seed(0)
    shapes = [(3, 4), (5,), (2, 2, 2)]
    n = sum(int(torch.tensor(s).prod()) for s in shapes)
    flat = torch.randn(n)
    params, off = [], 0
    for s in shapes:
        k = int(torch.tensor(s).prod())
        params.append(torch.nn.Parameter(flat[off:off + k].view(s)))
        off += k
    assert all(p.data_ptr() == flat.data_ptr() + 4 * o for p, o in zip(params, (0, 12, 17)))
    model = Namespace(parameters=lambda: iter(params))
    eng = Namespace(params=flat, n_params=n)
    cls = {"adam": torch.optim.Adam, "sgd": torch.optim.SGD, "rmsprop": torch.optim.RMSprop}[name]
    opt = cls(params, lr=1e-2, **kwargs)
    for _ in range(3):
        for p in params:
            p.grad = torch.randn_like(p)
        opt.step()
    st = Trainer._flat_optimizer_state(model, eng, opt.state_dict(), name)
    assert st["optimizer"] == name

    def cat(key):
        return torch.cat([opt.state[p][key].reshape(-1) for p in params])

    if name == "adam":
        assert torch.equal(st["adam_m"], cat("exp_avg")) and torch.equal(st["adam_v"], cat("exp_avg_sq")) and st["step_count"] == 3
        assert (st["opt_aux"] is None) == (not kwargs.get("amsgrad"))
        if kwargs.get("amsgrad"):
            assert torch.equal(st["opt_aux"], cat("max_exp_avg_sq"))
    elif name == "sgd":
        if kwargs.get("momentum"):
            assert torch.equal(st["adam_m"], cat("momentum_buffer")) and st["momentum_inited"]
        else:
            assert st["adam_m"] is None and not st["momentum_inited"]
        assert st["adam_v"] is None and st["opt_aux"] is None
    else:
        assert torch.equal(st["adam_v"], cat("square_avg")) and st["step_count"] == 3
        if kwargs.get("momentum"):
            assert torch.equal(st["adam_m"], cat("momentum_buffer")) and torch.equal(st["opt_aux"], cat("grad_avg"))
        else:
            assert st["adam_m"] is None and st["opt_aux"] is None
    # the wrong optimiser's reader refuses the state instead of guessing
    other = "sgd" if name != "sgd" else "adam"
    if opt.state_dict()["state"] and any(len(v) for v in opt.state_dict()["state"].values()):
        with pytest.raises(KeyError):
            Trainer._flat_optimizer_state(model, eng, opt.state_dict(), other)
