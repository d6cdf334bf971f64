"""The window sampler and I/O helpers around the hot path (SURVEY.md par. 8f): CPU tests pin the numpy restatement
(oracle/mimicry_oracle.py) to tests/golden/mimicry.npz — outputs of the reference's own MimicryDataset, calc_jerk,
get_face_indicies, dictify_frames and expand_face_dim, captured by tests/golden/make_golden.py — and to hand-built
expectations; the GPU tests compare lets_face_it_amd.mimicry_data_module / generate_motion / mimicry_logger with both."""
import random
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import mimicry_oracle as mo

DATA_HP = {"expression_dim": 5, "jaw_dim": 3, "neck_dim": 3, "speech_dim": 6, "file_name": "unused"}
COND_HP = {"p1_face": {"history": 2}, "p2_face": {"history": 3}, "p1_speech": {"history": 2}, "p2_speech": {"history": 0},
           "use_frame_nb": False}


def make_store(lens=(7, 3, 12), seed=0):
    """A tiny HDF5-shaped tree: bins of different lengths, expression wider than expression_dim (the file stores 100)."""
    rng = np.random.RandomState(seed)
    dims = {"flame_expression": 8, "flame_jaw": 3, "flame_neck": 3, "mfcc": 4, "prosody": 2}
    store = {"train": {k: {} for k in dims}}
    for i, n in enumerate(lens):
        for kind, d in dims.items():
            store["train"][kind][str(i)] = {who: rng.randn(n, d).astype(np.float32) for who in ("agent", "interlocutor")}
    return store


class MimicryGolden:
    """tests/golden/mimicry.npz: the corpus tree, the reference dataset's shuffled window list and every item, jerk values,
    FLAME index lists and the generate_motion layout helpers' outputs."""

    def __init__(self):
        import json
        import os
        self.raw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mimicry.npz"))
        hp = json.loads(str(self.raw["hparams_json"]))
        self.data_hp, self.cond_hp = hp["Data"], hp["Conditioning"]
        self.seq_len, self.seed = int(self.raw["seq_len"]), int(self.raw["seed"])
        self.store = {}
        for name in self.raw.files:
            parts = name.split("/")
            if parts[0] == "store":
                self.store.setdefault(parts[1], {}).setdefault(parts[2], {}).setdefault(parts[3], {})[parts[4]] = self.raw[name]
        # np.savez keeps insertion order per key, but the tree above is rebuilt in file order: bins sorted as written
        self.keys = [str(k) for k in self.raw["index/keys"]]
        self.starts = [int(v) for v in self.raw["index/starts"]]
        self.hist = {m: self.cond_hp[m]["history"] for m in ("p1_speech", "p2_speech", "p2_face")}

    def item(self, i):
        pre = "item/%d/" % i
        return {k[len(pre):]: self.raw[k] for k in self.raw.files if k.startswith(pre)}


@pytest.fixture(scope="module")
def mg():
    return MimicryGolden()


def test_oracle_dataset_matches_reference_capture(mg):
    """mimicry_oracle.window_index + the reference's one-off random.sample shuffle (mimicry_data_module.py:33-43) and
    get_item (:45-78) against what the reference's MimicryDataset produced on the same tree and seed."""
    enum = mo.window_index(mg.store, "train", mg.seq_len)
    random.seed(mg.seed)
    order = random.sample(enum, len(enum))
    assert order == list(zip(mg.keys, mg.starts)) and len(order) == 3 + 8 + 1 + 36
    for i, (key, start) in enumerate(order):
        got = mo.get_item(mg.store, "train", key, start, mg.seq_len, mg.data_hp["expression_dim"], mg.hist)
        want = mg.item(i)
        assert set(got) == set(want) == {"p1_face", "p1_speech", "p2_face"}
        for name in want:
            assert got[name].dtype == np.float32 and np.array_equal(got[name], want[name]), (i, name)


def test_oracle_helpers_match_reference_capture(mg):
    r = mg.raw
    for j in range(3):
        assert abs(mo.calc_jerk(r["jerk/%d/x" % j]) - float(r["jerk/%d/out" % j])) < 1e-6 * max(1.0, float(r["jerk/%d/out" % j]))
    for j in range(4):
        a = [int(v) for v in r["face_idx/%d/args" % j]]
        assert mo.get_face_indicies(a[0], a[1], a[2], offset=a[3]) == [int(v) for v in r["face_idx/%d/out" % j]]
    import json
    hp = json.loads(str(r["gm/hparams_json"]))
    d = mo.dictify_frames(r["gm/frames"], hp)
    for k in ("p1_face", "p1_speech", "p2_face", "p2_speech"):
        assert np.array_equal(d[k], r["gm/dictify/" + k]), k
    assert np.array_equal(mo.expand_face_dim(r["gm/expand/seq"], hp), r["gm/expand/out"])


def test_window_index_enumeration():
    store = make_store()
    idx = mo.window_index(store, "train", 5)
    # bins of 7, 3, 12 frames at seq_len 5: 3 + 0 + 8 stride-1 windows, bin order then start order
    assert idx == [("0", s) for s in range(3)] + [("2", s) for s in range(8)]
    assert mo.window_index(store, "train", 13) == []


def test_get_item_layout():
    store = make_store()
    d = mo.get_item(store, "train", "2", 4, 5, DATA_HP["expression_dim"],
                    {"p1_speech": 2, "p2_speech": 0, "p2_face": 3})
    assert set(d) == {"p1_face", "p1_speech", "p2_face"}            # p2_speech history 0: stream absent
    t = store["train"]
    assert d["p1_face"].shape == (5, 11) and d["p1_speech"].shape == (5, 6)
    np.testing.assert_array_equal(d["p1_face"][:, :5], t["flame_expression"]["2"]["agent"][4:9, :5])
    np.testing.assert_array_equal(d["p1_face"][:, 5:8], t["flame_jaw"]["2"]["agent"][4:9])
    np.testing.assert_array_equal(d["p1_face"][:, 8:], t["flame_neck"]["2"]["agent"][4:9])
    np.testing.assert_array_equal(d["p1_speech"][:, :4], t["mfcc"]["2"]["agent"][4:9])
    np.testing.assert_array_equal(d["p2_face"][:, 5:8], t["flame_jaw"]["2"]["interlocutor"][4:9])


def test_calc_jerk_oracle():
    t = np.arange(10, dtype=np.float32)
    x = np.stack([t ** 3, t ** 2], axis=-1)[None]      # third difference of t^3 is 6, of t^2 is 0
    assert abs(mo.calc_jerk(x) - 3.0) < 1e-5


def test_face_layout_helpers():
    hp = {"expression_dim": 50, "jaw_dim": 3, "neck_dim": 3, "speech_dim": 30}
    frames = np.arange(4 * 272, dtype=np.float32).reshape(4, 272)
    d = mo.dictify_frames(frames, hp)
    assert d["p1_face"].shape == (4, 56) and d["p2_speech"].shape == (4, 30)
    assert d["p1_face"][0, 50] == 100 and d["p1_face"][0, 53] == 103 and d["p2_face"][0, 0] == 136 and d["p2_face"][0, 50] == 236
    assert d["p1_speech"][0, 0] == 106 and d["p2_speech"][0, 0] == 242
    out = mo.expand_face_dim(d["p1_face"][None], hp)
    assert out.shape == (1, 4, 106)
    np.testing.assert_array_equal(out[0, :, :50], frames[:, :50])
    np.testing.assert_array_equal(out[0, :, 100:106], frames[:, 100:106])
    assert (out[0, :, 50:100] == 0).all()


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("source", ["dict", "npz"])
def test_window_sampler_matches_reference_dataset(gpu_device, tmp_path, source):
    from lets_face_it_amd.mimicry_data_module import MimicryDataset, WindowLoader
    store = make_store(lens=(7, 3, 12, 5, 40))
    src = store
    if source == "npz":
        flat = {"train/%s/%s/%s" % (k, b, w): a for k, bins in store["train"].items() for b, d in bins.items() for w, a in d.items()}
        np.savez(tmp_path / "corpus.npz", **flat)
        src = tmp_path / "corpus.npz"
    random.seed(3)
    ds = MimicryDataset(src, "train", data_hparams=DATA_HP, conditioning_hparams=COND_HP, seq_len=5, device=gpu_device)
    ref = mo.window_index(store, "train", 5)
    assert len(ds) == len(ref) == 3 + 8 + 1 + 36
    assert sorted(ds.indicies) == sorted(ref)                      # same windows, shuffled once like the reference
    hist = {"p1_speech": 2, "p2_speech": 0, "p2_face": 3}
    idx = list(range(len(ds)))
    batch = ds.batch(idx)
    assert set(batch) == {"p1_face", "p1_speech", "p2_face"}
    for i in idx:
        key, start = ds.indicies[i]
        want = mo.get_item(store, "train", key, start, 5, DATA_HP["expression_dim"], hist)
        for name, arr in want.items():
            assert torch.equal(batch[name][i].cpu(), torch.from_numpy(arr)), (i, name)     # a copy: bit-exact
    item = ds[4]
    assert item["p1_face"].shape == (5, 11) and torch.equal(item["p1_face"], batch["p1_face"][4])
    # loader: every window exactly once per epoch, ragged last batch kept (drop_last=False), ranks split the batches
    seen = []
    for rank in (0, 1):
        for b in WindowLoader(ds, 7, shuffle=True, rank=rank, world_size=2, generator=torch.Generator().manual_seed(1)):
            assert b["p1_face"].is_cuda and b["p1_face"].shape[1:] == (5, 11)
            seen.append(b["p1_face"].cpu())
    got = torch.cat(seen)
    assert got.shape[0] == len(ds)
    allw = batch["p1_face"].cpu()
    assert sorted(map(lambda t: t.flatten().tolist(), got)) == sorted(map(lambda t: t.flatten().tolist(), allw))
    with pytest.raises(IndexError):
        ds.batch([len(ds)])


@pytest.mark.gpu
def test_window_sampler_matches_reference_capture(gpu_device, mg):
    """The GPU-resident sampler against the reference's own MimicryDataset output (tests/golden/mimicry.npz): same seed ->
    the same shuffled window list, and every item bit-identical (one lfi_gather_sequences launch per modality)."""
    from lets_face_it_amd.mimicry_data_module import MimicryDataset
    random.seed(mg.seed)
    ds = MimicryDataset(mg.store, "train", data_hparams=mg.data_hp, conditioning_hparams=mg.cond_hp, seq_len=mg.seq_len,
                        device=gpu_device)
    assert ds.indicies == list(zip(mg.keys, mg.starts))
    batch = ds.batch(list(range(len(ds))))
    for i in range(len(ds)):
        want = mg.item(i)
        assert set(batch) == set(want)
        for name, arr in want.items():
            assert torch.equal(batch[name][i].cpu(), torch.from_numpy(arr)), (i, name)
    one = ds[7]
    for name, arr in mg.item(7).items():
        assert torch.equal(one[name].cpu(), torch.from_numpy(arr))


@pytest.mark.gpu
def test_jerk_and_layout_helpers_match_reference_capture(gpu_device, mg):
    import json
    from lets_face_it_amd import generate_motion as gm
    from lets_face_it_amd.glow.utils import calc_jerk
    r = mg.raw
    for j in range(3):
        got = float(calc_jerk(torch.from_numpy(r["jerk/%d/x" % j]).to(gpu_device)))
        want = float(r["jerk/%d/out" % j])
        assert abs(got - want) < 1e-6 * max(1.0, want), (j, got, want)
    hp = json.loads(str(r["gm/hparams_json"]))
    d = gm.dictify_frames(torch.from_numpy(r["gm/frames"]).to(gpu_device), hp)
    for k in ("p1_face", "p1_speech", "p2_face", "p2_speech"):
        assert torch.equal(d[k].cpu(), torch.from_numpy(r["gm/dictify/" + k])), k
    ex = gm.expand_face_dim(torch.from_numpy(r["gm/expand/seq"]).to(gpu_device), hp)
    assert torch.equal(ex.cpu(), torch.from_numpy(r["gm/expand/out"]))
    for j in range(4):
        a = [int(v) for v in r["face_idx/%d/args" % j]]
        assert gm.get_face_indicies(a[0], a[1], a[2], offset=a[3]) == [int(v) for v in r["face_idx/%d/out" % j]]


@pytest.mark.gpu
def test_window_sampler_feeds_the_model(gpu_device):
    """End to end at the boundary: batches from the sampler go straight into SeqGlow.forward (final widths, tiny corpus)."""
    from helpers import Fixture
    from lets_face_it_amd.glow.models import SeqGlow
    from lets_face_it_amd.mimicry_data_module import MimicryDataModule
    fx = Fixture("tiny")
    hp = fx.hp
    hp["Data"].update(expression_dim=10, jaw_dim=3, neck_dim=3, speech_dim=fx.S)
    hp["batch_size"] = 6
    hp["Train"]["seq_len"] = fx.T
    rng = np.random.RandomState(1)
    dims = {"flame_expression": 100, "flame_jaw": 3, "flame_neck": 3, "mfcc": fx.S - 2, "prosody": 2}
    store = {"train": {k: {str(i): {w: rng.randn(n, d).astype(np.float32) for w in ("agent", "interlocutor")}
                           for i, n in enumerate((30, 25))} for k, d in dims.items()}}
    dm = MimicryDataModule(Namespace(**hp), device=gpu_device, source=store)
    m = SeqGlow(Namespace(**hp))
    m.load_state_dict(fx.state_dict(torch.float32))
    m.to(gpu_device).eval()
    m.glow.set_actnorm_init(True)
    n = 0
    for batch in dm.train_dataloader():
        with torch.no_grad():
            _, loss, _ = m(batch)
        assert torch.isfinite(loss).all() and batch["p1_face"].shape[1:] == (fx.T, 16)
        n += batch["p1_face"].shape[0]
    assert n == (30 - fx.T + 1) + (25 - fx.T + 1)


@pytest.mark.gpu
def test_jerk_kernel_matches_reference_formula(gpu_device):
    from lets_face_it_amd.glow.utils import calc_jerk
    g = torch.Generator().manual_seed(0)
    for shape in ((3, 7, 5), (64, 276, 50), (1, 4, 1)):
        x = torch.randn(*shape, generator=g)
        got = float(calc_jerk(x.to(gpu_device)))
        want = mo.calc_jerk(x.numpy())
        assert abs(got - want) < 1e-6 * max(1.0, want), (shape, got, want)


@pytest.mark.gpu
def test_generate_motion_io_contract(gpu_device):
    """272-d frames -> model streams -> inference -> de-standardised 106-d FLAME vector, against the numpy restatement of the
    layout helpers and the engine's own sampler on the standardised streams."""
    from helpers import Fixture
    from lets_face_it_amd import generate_motion as gm
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fx = Fixture("tiny")
    hp = fx.hp
    hp["Data"].update(expression_dim=10, jaw_dim=3, neck_dim=3, speech_dim=fx.S)   # 16-d face, as the fixture's model
    model = LetsFaceItGlow(Namespace(**hp))
    model.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    model.to(gpu_device)
    model.seq_glow.glow.set_actnorm_init(True)
    g = torch.Generator().manual_seed(2)
    T = 30
    frames = torch.randn(T, 272, generator=g)
    stats = {"face_means": torch.randn(16, generator=g), "face_stds": torch.rand(16, generator=g) + 0.5,
             "speech_means": torch.randn(fx.S, generator=g), "speech_stds": torch.rand(fx.S, generator=g) + 0.5}
    d_ref = mo.dictify_frames(frames.numpy(), hp["Data"])
    d = gm.dictify_frames(frames.to(gpu_device), hp["Data"])
    for k in d_ref:
        assert torch.equal(d[k].cpu(), torch.from_numpy(d_ref[k])), k
    stats_d = {k: v.to(gpu_device) for k, v in stats.items()}
    noise = torch.randn(T - fx.start, 1, 16, generator=g).to(gpu_device)
    out = gm.generate_motion(frames.to(gpu_device), model, stats_d, eps=1, noise=noise)
    assert out.shape == (1, T - fx.start, 106)
    # the same thing by hand: standardise, sample, de-standardise, expand
    cond = {"p1_face": torch.zeros(1, fx.start, 16, device=gpu_device)}
    for k, m, s in (("p2_face", "face_means", "face_stds"), ("p1_speech", "speech_means", "speech_stds"),
                    ("p2_speech", "speech_means", "speech_stds")):
        cond[k] = ((torch.from_numpy(d_ref[k]) - stats[m]) / stats[s]).unsqueeze(0).to(gpu_device).contiguous()
    pred = model.seq_glow.inference(T, data=cond, noise=noise)
    want = mo.expand_face_dim((pred.cpu() * stats["face_stds"] + stats["face_means"]).numpy(), hp["Data"])
    assert np.abs(out.cpu().numpy() - want).max() < 1e-5
    assert (out[..., 10:100] == 0).all()


@pytest.mark.gpu
def test_mimicry_logger_metrics(gpu_device):
    """on_validation_batch_end logs what the reference's callback logs (mimicry_logger.py:154-239): jerk, the invertibility
    error (~0 for a bijection) and one mismatched-NLL probe per Mismatch entry whose modalities are all switched on."""
    from helpers import Fixture
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import calc_jerk
    from lets_face_it_amd.mimicry_logger import MimicryLogger
    fx = Fixture("tiny")
    hp = fx.hp
    hp["Validation"].update(inference=True, check_invertion=True, wrong_context_test=True, scale_logging=True, render=False,
                            seq_len=fx.T)
    model = LetsFaceItGlow(Namespace(**hp))
    model.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    model.to(gpu_device).eval()
    model.seq_glow.glow.set_actnorm_init(True)
    batch = {k: v.to(device=gpu_device, dtype=torch.float32).contiguous() for k, v in fx.batch().items()}
    out = model.validation_step(batch, 0)
    random.seed(0)
    torch.manual_seed(0)
    MimicryLogger().on_validation_batch_end(None, model, out, batch, 0, 0)
    L = model.logged
    # the reference's "error percentage" is |(backward_loss + loss) / loss| with backward_loss built from the NEGATED log-det
    # (models.py:630-645): the log-dets cancel and what is left is -2 mean(log p(z)) / ln 2, not 0 (reproduced, not "fixed")
    z = fx.get("eval/z")
    logp = (-0.5 * (z ** 2 + np.log(2 * np.pi))).sum(-1)
    want_err = abs(float(-2.0 * logp.mean() / np.log(2.0)) / float(fx.get("eval/loss")))
    assert abs(float(L["reconstruction/error_percentage"]) - want_err) < 1e-3 * want_err
    assert abs(float(L["jerk/gt_mean"]) - mo.calc_jerk(fx.batch()["p1_face"][:, -(fx.T - fx.start):].numpy())) < 1e-5
    assert float(L["jerk/generated_mean"]) > 0 and abs(float(L["jerk/generated_mean_ratio"])
                                                      - float(L["jerk/generated_mean"]) / float(L["jerk/gt_mean"])) < 1e-5
    assert abs(float(L["mismatched_nll/actual_nll"]) - float(fx.get("eval/loss"))) < 1e-3 * abs(float(fx.get("eval/loss")))
    probes = [k for k in L if k.startswith("mismatched_nll/shuffle_")]
    want = sum(all(hp["Conditioning"][m]["history"] > 0 for m in mods)
               for kind in ("shuffle_batch", "shuffle_time") for mods in hp["Mismatch"][kind].values())
    assert len(probes) == want and want >= 6
    for k in probes:
        assert torch.isfinite(L[k]).all()
        ratio = L["mismatched_nll_ratios/" + k.split("/", 1)[1]]
        assert abs(float(ratio) - (float(L["mismatched_nll/actual_nll"]) - float(L[k]))) < 1e-3
    MimicryLogger().on_validation_batch_end(None, model, out, batch, 1, 0)   # only the first batch is probed
    stats = MimicryLogger().log_scales(model)
    assert any(k.startswith("ActNorm/") for k in stats) and any(k.startswith("FlowStepScale/") for k in stats)


@pytest.mark.gpu
def test_stacked_mismatched_probes_equal_one_forward_each(gpu_device):
    """SURVEY.md par. 8 f2: the mismatched-NLL probes (mimicry_logger.py:199-238: one SeqGlow.forward per deranged batch, 10 at
    final_model.yaml) go through the engine as ONE stacked forward; every probe's NLL must be what its own forward gives."""
    from helpers import Fixture
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import derange_batch
    from lets_face_it_amd.mimicry_logger import MimicryLogger
    fx = Fixture("mid")
    model = LetsFaceItGlow(Namespace(**fx.hp))
    model.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    model.to(gpu_device).eval()
    model.seq_glow.glow.set_actnorm_init(True)
    batch = {k: v.to(device=gpu_device, dtype=torch.float32).contiguous() for k, v in fx.batch().items()}
    torch.manual_seed(3)
    probes = [derange_batch(batch, mods, shuffle_time=st)
              for st in (False, True) for mods in (["p2_face"], ["p2_speech"], ["p1_speech"], ["p2_face", "p2_speech"])]
    lg = MimicryLogger()
    with torch.no_grad():
        one_by_one = [model.seq_glow(b)[1] for b in probes]
        calls = []
        eng = model.seq_glow.engine
        fwd = eng.forward
        eng.forward = lambda *a, **k: (calls.append(a[0]["p1_face"].shape[0]), fwd(*a, **k))[1]
        stacked = lg.stacked_nll(model, probes)
        lg.max_stack_frames = 3 * fx.B * fx.T          # a cap that forces groups of three
        grouped = lg.stacked_nll(model, probes)
        eng.forward = fwd
    assert calls == [8 * fx.B, 3 * fx.B, 3 * fx.B, 2 * fx.B]
    for a, b, c in zip(one_by_one, stacked, grouped):
        assert abs(float(a) - float(b)) < 1e-5 * max(1.0, abs(float(a))) and abs(float(a) - float(c)) < 1e-5 * max(1.0, abs(float(a)))
