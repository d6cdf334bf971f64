"""GPU-resident stand-in for the reference's data module (code/glow_pytorch/mimicry_data_module.py), SURVEY.md par. 8f row 1.

The reference's MimicryDataset re-opens the HDF5 file and slices seven datasets for EVERY item (:45-78), eight DataLoader
workers collate, and the batch crosses PCIe each step; at the engine's rate (~90 batches of 256 x 80 frames per second and
GPU) that is the bottleneck. A split of the corpus is small (558 min of 25 fps frames x 4 streams < 1 GB fp32), so here it
is loaded ONCE: per modality one (rows x dim) matrix in HBM with the recording bins back to back, plus the table of valid
window starts. A batch is then one HIP gather per modality (lfi_gather_sequences): `dst[b, t] = src[starts[b] + t]`,
written straight into the (B, T, dim) tensors SeqGlow.forward takes. Same class names, constructor arguments, batch dict and
window enumeration as the reference (every stride-1 window of seq_len frames of every bin with >= seq_len frames, :33-41).

Input: the HDF5 file of feature_extraction/combine_features.py:246-265 (`/{train,val,test}/{flame_expression,flame_jaw,
flame_neck,mfcc,prosody}/{i}/{agent,interlocutor}`) when h5py is importable, an `.npz` export of the same tree (keys
"train/flame_expression/0/agent", ...), or the tree itself as nested dicts.
"""
import random
from pathlib import Path

import numpy as np
import torch

from . import _lib
from ._lib import check

KINDS = ("flame_expression", "flame_jaw", "flame_neck", "mfcc", "prosody")


def load_store(source, data_type):
    """-> {kind: {bin_key: {"agent": array, "interlocutor": array}}} for one split."""
    if isinstance(source, dict):
        return source[data_type]
    path = Path(source)
    if path.suffix == ".npz":
        tree = {}
        with np.load(path) as z:
            for name in z.files:
                parts = name.split("/")
                if len(parts) == 4 and parts[0] == data_type:
                    tree.setdefault(parts[1], {}).setdefault(parts[2], {})[parts[3]] = z[name]
        if not tree:
            raise KeyError("%s holds no '%s/...' arrays" % (path, data_type))
        return tree
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("reading %s needs h5py; export the file to .npz (keys '<split>/<kind>/<bin>/<who>') or pass the "
                           "tree as nested dicts" % path) from e
    with h5py.File(path, "r") as f:
        return {kind: {key: {who: np.asarray(ds[who]) for who in ("agent", "interlocutor")}
                       for key, ds in f[data_type][kind].items()} for kind in KINDS}


class MimicryDataset:
    """All windows of one split, resident on `device`. Indexable like the reference's Dataset (`ds[i]` -> dict of (T, dim)
    tensors) and batchable in one call (`ds.batch(indices)` -> dict of (B, T, dim) tensors, one gather launch per modality)."""

    def __init__(self, file_name, data_type, data_hparams=None, conditioning_hparams=None, seq_len=None, device="cuda"):
        self.file_name, self.data_type, self.seq_len = file_name, data_type, int(seq_len)
        self.expression_dim = data_hparams["expression_dim"]
        self.speech_dim = data_hparams["speech_dim"]
        self.p1_speech_history = conditioning_hparams["p1_speech"]["history"]
        self.p2_speech_history = conditioning_hparams["p2_speech"]["history"]
        self.p2_face_history = conditioning_hparams["p2_face"]["history"]
        self.use_frame_nb = conditioning_hparams["use_frame_nb"]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LfiError("MimicryDataset (lets_face_it_amd) keeps the corpus in GPU memory; got device %s" % device)
        self.L = _lib.lib()
        store = load_store(file_name, data_type)

        def face(key, who):   # expression[:, :expression_dim] | jaw | neck   (mimicry_data_module.py:52-60)
            return np.concatenate([np.asarray(store["flame_expression"][key][who])[:, :self.expression_dim],
                                   store["flame_jaw"][key][who], store["flame_neck"][key][who]], axis=1)

        def speech(key, who):  # mfcc | prosody   (:62-65)
            return np.concatenate([store["mfcc"][key][who], store["prosody"][key][who]], axis=1)

        streams = {"p1_face": (face, "agent")}
        if self.p1_speech_history:
            streams["p1_speech"] = (speech, "agent")
        if self.p2_speech_history:
            streams["p2_speech"] = (speech, "interlocutor")
        if self.p2_face_history:
            streams["p2_face"] = (face, "interlocutor")

        # window table in the reference's enumeration order (:33-41), then shuffled once with Python's `random` (:43)
        keys, lens, offs, tmp = [], [], [], []
        row = 0
        for key, chunk in store["prosody"].items():
            n = len(chunk["agent"])
            keys.append(key)
            lens.append(n)
            offs.append(row)
            if n >= self.seq_len:
                tmp.extend((key, s, row + s) for s in range(n - self.seq_len + 1))
            row += n
        self.rows = row
        order = random.sample(range(len(tmp)), len(tmp))
        self.indicies = [(tmp[i][0], tmp[i][1]) for i in order]          # (bin key, first frame) like the reference's list
        self._starts_host = torch.tensor([tmp[i][2] for i in order], dtype=torch.int64)
        self.starts = self._starts_host.to(self.device)
        self.data = {}
        for name, (fn, who) in streams.items():
            mat = np.concatenate([fn(k, who) for k in keys], axis=0).astype(np.float32) if keys else np.zeros((0, 1), np.float32)
            if mat.shape[0] != self.rows:
                raise ValueError("%s: %d rows, prosody has %d" % (name, mat.shape[0], self.rows))
            self.data[name] = torch.from_numpy(np.ascontiguousarray(mat)).to(self.device)

    def __len__(self):
        return len(self.indicies)

    def batch(self, index):
        """index: int64 tensor / list of window numbers -> {"p1_face": (B, T, C), ...} on the device."""
        idx = torch.as_tensor(index, dtype=torch.int64)
        if idx.numel() == 0:
            raise IndexError("empty batch")
        if int(idx.min()) < 0 or int(idx.max()) >= len(self):
            raise IndexError("window index out of range (0 .. %d)" % (len(self) - 1))
        # (through pinned memory: a pageable host-to-device copy synchronises the host with the stream, which drains the
        # training step queued behind it and leaves the GPU idle while the next one is issued - once per batch)
        if idx.device.type == "cpu" and self.device.type == "cuda":
            idx = idx.pin_memory().to(self.device, non_blocking=True)
        else:                                      # an index tensor that already lives on a device: no staging copy
            idx = idx.to(self.device)
        starts = self.starts[idx].contiguous()
        B, T = starts.numel(), self.seq_len
        st = torch.cuda.current_stream().cuda_stream
        out = {}
        for name, src in self.data.items():
            dst = torch.empty(B, T, src.shape[1], dtype=torch.float32, device=self.device)
            check(self.L.lfi_gather_sequences(src.data_ptr(), src.shape[0], src.shape[1], starts.data_ptr(), B, T,
                                              dst.data_ptr(), st), "lfi_gather_sequences")
            out[name] = dst
        return out

    def __getitem__(self, index):
        return {k: v[0] for k, v in self.batch([int(index)]).items()}


class WindowLoader:
    """Stands in for torch.utils.data.DataLoader(dataset, batch_size, shuffle, drop_last=False): iterating yields batch
    dicts already on the device.

    Under data parallelism it behaves as DataLoader + DistributedSampler do under Lightning's DDP: ONE permutation shared by
    all ranks (seeded by `seed + epoch`, set_epoch() as the sampler's), padded by wrapping around to a multiple of
    world_size, rank r taking elements r, r + world, ...; every rank therefore runs the SAME number of batches of the SAME
    sizes (the last one may be ragged, equally on all ranks). Each optimiser step issues gradient all-reduces, so ranks with
    different step counts would pair collectives of different steps and finally hang."""

    def __init__(self, dataset, batch_size, shuffle=True, rank=0, world_size=1, generator=None, seed=0):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.rank, self.world_size, self.generator = int(rank), int(world_size), generator
        self.seed, self.epoch = int(seed), 0
        self.skip_batches = 0     # the next iteration starts at this batch of the epoch's order (Trainer: mid-epoch resume)
        if not 0 <= self.rank < self.world_size:
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world_size))

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _per_rank(self):
        return (len(self.dataset) + self.world_size - 1) // self.world_size

    def __len__(self):
        return (self._per_rank() + self.batch_size - 1) // self.batch_size

    def _order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return torch.arange(n)
        if self.world_size == 1:
            return torch.randperm(n, generator=self.generator)     # the global torch RNG, as DataLoader(shuffle=True)
        g = self.generator if self.generator is not None else torch.Generator().manual_seed(self.seed + self.epoch)
        return torch.randperm(n, generator=g)                      # identical on every rank

    def __iter__(self):
        order = self._order()
        if self.world_size > 1:
            n, total = order.numel(), self._per_rank() * self.world_size
            if n == 0:
                return
            if total > n:
                order = torch.cat([order] * ((total + n - 1) // n))[:total]
            order = order[self.rank:total:self.world_size]
        skip, self.skip_batches = int(self.skip_batches), 0
        for b in range(skip * self.batch_size, order.numel(), self.batch_size):
            yield self.dataset.batch(order[b:b + self.batch_size])


class MimicryDataModule:
    """mimicry_data_module.py:84-128 with the loaders above (`workers` is accepted and unused: nothing is left to parallelise)."""

    def __init__(self, hparams, workers=8, device=None, source=None):
        self.hparams, self.workers = hparams, workers
        self.file_name = source if source is not None else Path(hparams.dataset_root) / self.hparams.Data["file_name"]
        self.device = device if device is not None else "cuda"
        self._sets = {}

    def _data_loader(self, data_type, shuffle=True, seq_len=25):
        key = (data_type, seq_len)
        if key not in self._sets:
            self._sets[key] = MimicryDataset(self.file_name, data_type, seq_len=seq_len, data_hparams=self.hparams.Data,
                                             conditioning_hparams=self.hparams.Conditioning, device=self.device)
        import os
        return WindowLoader(self._sets[key], self.hparams.batch_size, shuffle=shuffle,
                            rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")),
                            seed=int(getattr(self.hparams, "seed", 1234) or 1234))

    def train_dataloader(self):
        return self._data_loader("train", seq_len=self.hparams.Train["seq_len"])

    def val_dataloader(self):
        return self._data_loader("val", shuffle=False, seq_len=self.hparams.Validation["seq_len"])

    def test_dataloader(self):
        return self._data_loader("test", shuffle=False, seq_len=self.hparams.Test["seq_len"])
