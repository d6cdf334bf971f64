"""CPU: the host-side mirror of the reference surface (no kernels run): construction parity, hparams, helpers."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import Fixture, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = os.path.join(ROOT, "lets_face_it_amd", "hparams")


def perturb_like_fixture(model):
    gen = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "final_linear" in name:
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)
            elif "actnorm" in name:
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith("invconv.log_s"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)


@pytest.mark.parametrize("name", ["tiny", "tiny_additive", "odd", "mid"])
def test_same_seeds_give_the_reference_initialisation(name):
    """Construction order and RNG use match the reference: seed_everything(1234) -> identical initial weights and keys."""
    from lets_face_it_amd.glow.models import SeqGlow
    fx = Fixture(name)
    torch.manual_seed(1234)
    np.random.seed(1234)
    m = SeqGlow(Namespace(**fx.hp))
    perturb_like_fixture(m)
    ref = fx.state_dict(torch.float32)
    sd = m.state_dict()
    assert sorted(sd) == sorted(ref)
    for k in ref:
        assert sd[k].shape == ref[k].shape, k
        assert torch.equal(sd[k], ref[k]), k


def test_model_spec_matches_survey_dims():
    from lets_face_it_amd.engine import ModelSpec
    from lets_face_it_amd.glow.utils import load_hparams_file
    s = ModelSpec(Namespace(**load_hparams_file(os.path.join(HP, "final_model.yaml"))))
    assert (s.C, s.S, s.D, s.H, s.Ks, s.E, s.start, s.Cout, s.G, s.I) == (56, 30, 512, 128, 16, 1560, 24, 56, 384, 540)
    s = ModelSpec(Namespace(**load_hparams_file(os.path.join(HP, "final_model_synthetic.yaml"))))
    assert (s.C, s.S, s.E, s.Ch, s.C2) == (50, 27, 1530, 25, 25)
    s = ModelSpec(Namespace(**load_hparams_file(os.path.join(HP, "no_speech.yaml"))))
    assert [e.name for e in s.encoders] == ["p1_face", "p2_face", "p1_speech"] and s.start == 5


def test_parameter_count_matches_reference():
    """17 647 488 parameters for final_model.yaml, 17 342 112 at C=50/S=27."""
    from lets_face_it_amd.glow.models import SeqGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    for f, n in (("final_model.yaml", 17647488), ("final_model_synthetic.yaml", 17342112)):
        m = SeqGlow(Namespace(**load_hparams_file(os.path.join(HP, f))))
        assert sum(p.numel() for p in m.parameters()) == n


def test_get_hparams_overrides(tmp_path):
    from lets_face_it_amd.glow.utils import get_hparams
    hp, name = get_hparams([os.path.join(HP, "final_model.yaml"), "--max_epochs", "3", "--lr=0.5", "--gpus", "8"])
    assert name == "final_model.yaml" and hp.max_epochs == 3 and hp.lr == 0.5 and hp.gpus == 8
    assert hp.Glow["rnn_type"] == "gru" and hp.dataset_root == os.environ.get("LFI_DATA_DIR", "/data")
    js = tmp_path / "x.json"
    js.write_text('{"Glow": {"K": 2}, // comment\n "lr": 1e-3 /* c */ }')
    from lets_face_it_amd.glow.utils import load_hparams_file
    assert load_hparams_file(str(js))["Glow"] == {"K": 2, "rnn_type": "gru"}


def test_helpers_match_reference_semantics():
    from lets_face_it_amd.glow import utils
    fx = Fixture("tiny")
    hp = Namespace(**fx.hp)
    assert utils.get_longest_history(hp.Conditioning) == fx.start
    assert utils.get_mismatched_modalities(hp) == (["p2_face", "p2_speech"], "p2")
    b = fx.batch(torch.float32)
    torch.manual_seed(0)
    perm = torch.randperm(fx.B)
    torch.manual_seed(0)
    mixed = utils.derange_batch(b, ["p2_face", "p2_speech"])
    assert torch.equal(mixed["p2_face"], b["p2_face"][perm]) and mixed["p1_face"] is b["p1_face"]
    with pytest.raises(RuntimeError, match="GPU only"):   # the metric is a device reduction (tests/test_data_module.py)
        utils.calc_jerk(torch.zeros(1, 6, 4))
    with pytest.raises(AssertionError):
        bad = Namespace(**Fixture("tiny").hp)
        bad.Train["seq_len"] = 4
        utils.test_params(bad)


def test_trainer_lr_schedule_is_steplr():
    from lets_face_it_amd.glow.utils import load_hparams_file
    from lets_face_it_amd.trainer import Trainer
    hp = Namespace(**load_hparams_file(os.path.join(HP, "final_model.yaml")))
    t = Trainer(hp, device="cpu")
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=hp.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, **hp.Optim["Schedule"]["args"]["step"])
    for epoch in range(10):
        assert abs(t.lr_at(epoch) - opt.param_groups[0]["lr"]) < 1e-12
        opt.step()
        sched.step()


def test_standalone_modules_refuse_cpu_tensors():
    """ActNorm2d / InvertibleConv1x1 called on their own (the reference's test_modules.py:9-29) run through the HIP library
    too: there is no torch-expression path to fall back to. (The round trips themselves: tests/test_gpu_modules.py.)"""
    from lets_face_it_amd.glow import modules
    np.random.seed(0)
    x = torch.rand(6, 54)
    an = modules.ActNorm2d(54)
    an.inited = True
    with pytest.raises(RuntimeError, match="GPU only"):
        an(x, 0.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        modules.InvertibleConv1x1(54, LU_decomposed=True)(x, 0.0)


def test_oracle_is_not_imported_by_the_product():
    import re
    pkg = os.path.join(ROOT, "lets_face_it_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f


def test_oom_is_reported_in_the_wording_the_tuning_harness_matches():
    """hparams_tuning.py:162-166 halves the batch on `str(e).startswith("CUDA out of memory")`; ROCm says "HIP out of memory"."""
    from lets_face_it_amd._lib import translate_oom

    @translate_oom
    def alloc():
        raise torch.OutOfMemoryError("HIP out of memory. Tried to allocate 20.00 GiB. GPU 0 has a total capacity of 287.98 GiB")

    with pytest.raises(RuntimeError) as ei:
        alloc()
    assert str(ei.value).startswith("CUDA out of memory") and "Tried to allocate 20.00 GiB" in str(ei.value)

    @translate_oom
    def other():
        raise RuntimeError("something else")

    with pytest.raises(RuntimeError, match="something else"):
        other()


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"), limit=300):
    import subprocess
    import sys
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")   # a HIP poison: nothing here may reach the GPU
    for k in drop:
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=limit)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """Inside a launcher (WORLD_SIZE set) `--gpus N` must equal it; the refusal comes before anything touches the GPU (the HIP
    poison proves the order: the error text is bench.py's own, not a HIP failure)."""
    r = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 4" in r.stderr, r.stderr[-500:]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver types the N = 1 line (no WORLD_SIZE): the process starts the two ranks
    under torch.distributed.run as children, they rendezvous on 127.0.0.1 (gloo, --launch-check: no GPU in this suite), and the
    parent's stdout is rank 0's ONE JSON line with n_gpus = 2."""
    import json
    r = _run_bench(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launch_check"] is True and rec["rank_sum"] == rec["expected_rank_sum"] == 3.0
    assert "--nproc-per-node 2" in r.stderr and "--master-addr 127.0.0.1" in r.stderr


def test_bench_gpus_n_relays_a_failing_rank():
    """The real workload on a box without GPUs: every rank dies when it asks for its device; the parent prints no JSON line and
    exits non-zero with the launcher's code."""
    r = _run_bench(["--gpus", "2", "--quick", "--steps", "1", "--warmup", "0"], {"LFI_DIST_BACKEND": "gloo"}, limit=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")], r.stdout[-500:]


def test_trainer_picks_rccl_on_cuda(monkeypatch):
    """Trainer.setup_distributed(): backend "nccl" (= RCCL on ROCm) for a cuda device, gloo for the CPU rehearsal; rendezvous
    on 127.0.0.1; HSA_ENABLE_IPC_MODE_LEGACY=0 kept in the environment (dmabuf IPC only on these hosts)."""
    from argparse import Namespace
    import torch.distributed as dist
    from lets_face_it_amd import trainer as tr_mod
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: calls.append((backend, kw)))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", str(d))))
    hp = Namespace(**Fixture("tiny").hp)
    t = tr_mod.Trainer(hp)                       # default device: cuda:<LOCAL_RANK>
    assert str(t.device) == "cuda:1"
    t.setup_distributed()
    assert calls[0] == ("set_device", "cuda:1")
    assert calls[1][0] == "nccl" and calls[1][1] == {"rank": 1, "world_size": 2}
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    calls.clear()
    tr_mod.Trainer(hp, device="cpu").setup_distributed()
    assert calls[0][0] == "gloo"


class _FakeWindows:
    """len() + batch(): what WindowLoader needs of a dataset (no GPU)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def batch(self, idx):
        return {"idx": torch.as_tensor(idx).clone()}


@pytest.mark.parametrize("n,batch,world", [(50, 7, 3), (48, 7, 2), (5, 4, 8), (64, 8, 4), (1, 3, 2)])
def test_window_loader_gives_every_rank_the_same_steps(n, batch, world):
    """every rank must run the same number of equally sized batches per epoch, else the per-step gradient
    all-reduces pair up across different steps and hang. DistributedSampler semantics: shared permutation, wrapped to a
    multiple of the world size, rank r takes r, r + world, ..."""
    from lets_face_it_amd.mimicry_data_module import WindowLoader
    ds = _FakeWindows(n)
    per_rank = []
    for rank in range(world):
        ld = WindowLoader(ds, batch, shuffle=True, rank=rank, world_size=world, seed=11)
        ld.set_epoch(3)
        got = [b["idx"] for b in ld]
        assert len(got) == len(ld)
        per_rank.append(got)
    sizes = [[int(b.numel()) for b in got] for got in per_rank]
    assert all(sz == sizes[0] for sz in sizes)                       # same step count, same batch sizes
    total = -(-n // world) * world
    allidx = torch.cat([torch.cat(g) for g in per_rank])
    assert allidx.numel() == total and set(allidx.tolist()) == set(range(n))    # everything seen, <= world - 1 repeats
    again = [b["idx"] for b in WindowLoader(ds, batch, True, 0, world, seed=11)]            # epoch 0: another permutation
    if n > 3:
        assert not all(torch.equal(a, b) for a, b in zip(again, per_rank[0]))
    # single process: DataLoader(drop_last=False) semantics, ragged last batch kept
    one = [b["idx"] for b in WindowLoader(ds, batch, shuffle=False)]
    assert torch.equal(torch.cat(one), torch.arange(n)) and len(one) == -(-n // batch)


def test_window_encoder_forward_variant_selection(monkeypatch):
    """lfi_encode_windows_fwd_variant (host logic only: no GPU): which forward kernel a descriptor takes by shape and switches -
    the round-5 kernel with the epilogue under the matrix phase only for hid = 256 launches that fill the chip and, by default, only
    where no stash is written (glow/models.py:55-80)."""
    import ctypes as C
    from lets_face_it_amd import _lib
    from lets_face_it_amd._lib import EncDesc
    L = _lib.lib()
    for k in ("LFI_ENC_T16", "LFI_ENC_R64", "LFI_ENC_M16", "LFI_ENC_WIDE"):
        monkeypatch.delenv(k, raising=False)
    big = EncDesc(256, 80, 56, 24, 24, 256, 896, 256, 1, 0, 0, 1, 1)       # p2_face at the headline shape
    small = EncDesc(256, 80, 56, 24, 2, 128, 896, 768, 1, 0, 0, 1, 1)      # p1_speech: hid 128
    ragged = EncDesc(40, 80, 56, 24, 24, 256, 896, 256, 1, 0, 0, 1, 1)     # 2 240 windows: too few 64-window workgroups for the chip
    assert L.lfi_encode_windows_fwd_variant(C.byref(big), 1, 0) == 5 and L.lfi_encode_windows_fwd_variant(C.byref(big), 1, 1) == 4
    assert L.lfi_encode_windows_fwd_variant(C.byref(small), 1, 1) in (2, 3) and L.lfi_encode_windows_fwd_variant(C.byref(ragged), 1, 0) == 2
    monkeypatch.setenv("LFI_ENC_T16", "2")
    assert L.lfi_encode_windows_fwd_variant(C.byref(big), 1, 1) == 5
    monkeypatch.setenv("LFI_ENC_T16", "0")
    assert L.lfi_encode_windows_fwd_variant(C.byref(big), 0, 0) == 4
    monkeypatch.setenv("LFI_ENC_M16", "0")
    assert L.lfi_encode_windows_fwd_variant(C.byref(big), 0, 0) == 3
    monkeypatch.setenv("LFI_ENC_R64", "0")
    assert L.lfi_encode_windows_fwd_variant(C.byref(big), 0, 0) == 2
    f32 = EncDesc(256, 80, 56, 24, 24, 256, 896, 256, 0, 0, 0, 0, 0)       # exact-product mode: the accumulator-layout fused kernel
    assert L.lfi_encode_windows_fwd_variant(C.byref(f32), 0, 0) == 1


def test_sampling_run_lengths(monkeypatch):
    """engine._sample_runs: how a sampling call's generated frames are cut into runs (the static part of run i + 1 is computed
    beside run i's chain): four equal runs from 64 frames up, LFI_SAMPLE_RUNS = a count or an explicit list of lengths."""
    from lets_face_it_amd.engine import GlowEngine
    monkeypatch.delenv("LFI_SAMPLE_RUNS", raising=False)
    assert GlowEngine._sample_runs(276) == [(0, 69), (69, 69), (138, 69), (207, 69)]
    assert GlowEngine._sample_runs(63) == [(0, 63)]
    assert GlowEngine._sample_runs(66) == [(0, 17), (17, 17), (34, 16), (50, 16)]
    # with the static part on its own share of the chip: six runs from 96 frames up
    assert GlowEngine._sample_runs(276, beside=True) == [(0, 46), (46, 46), (92, 46), (138, 46), (184, 46), (230, 46)]
    assert GlowEngine._sample_runs(95, beside=True) == GlowEngine._sample_runs(95) and len(GlowEngine._sample_runs(95)) == 4
    assert GlowEngine._sample_runs(63, beside=True) == [(0, 63)]
    monkeypatch.setenv("LFI_SAMPLE_RUNS", "3")
    assert GlowEngine._sample_runs(10) == [(0, 4), (4, 3), (7, 3)]
    assert GlowEngine._sample_runs(2) == [(0, 1), (1, 1)]
    monkeypatch.setenv("LFI_SAMPLE_RUNS", "80,70,66")
    assert GlowEngine._sample_runs(276) == [(0, 80), (80, 70), (150, 66), (216, 60)]
    assert GlowEngine._sample_runs(100) == [(0, 80), (80, 20)]
    for runs in (GlowEngine._sample_runs(276), GlowEngine._sample_runs(100)):
        assert sum(n for _, n in runs) in (276, 100) and all(runs[i][0] + runs[i][1] == runs[i + 1][0] for i in range(len(runs) - 1))
