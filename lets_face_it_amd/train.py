"""python -m lets_face_it_amd.train <hparams.yaml> [--key value ...]   (the reference's train.py:14-38)

Reads the reference's HDF5 dataset (Data.file_name under dataset_root; h5py, or an .npz export of the same tree) through the
GPU-resident lets_face_it_amd.mimicry_data_module.MimicryDataModule; `--data_module reference` keeps the reference
repository's own glow_pytorch.mimicry_data_module instead (it must be importable). For a self-contained run pass
--synthetic true, which trains on random (batch, T, dim) tensors of the configured shapes.
"""
import random

import numpy as np
import torch

from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
from lets_face_it_amd.glow.utils import get_hparams
from lets_face_it_amd.trainer import Trainer

RANDOM_SEED = 1234  # code/config.toml [project] random_seed


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class SyntheticDataModule:
    def __init__(self, hparams, steps=100):
        self.hp, self.steps = hparams, steps

    def _loader(self, seq_len, steps, seed):
        import os
        hp = self.hp
        C, S, B = hp.Conditioning["p1_face"]["dim"], hp.Data["speech_dim"], hp.batch_size
        g = torch.Generator().manual_seed(seed + 7919 * int(os.environ.get("RANK", "0")))   # another shard per rank
        for _ in range(steps):
            yield {"p1_face": torch.randn(B, seq_len, C, generator=g), "p2_face": torch.randn(B, seq_len, C, generator=g),
                   "p1_speech": torch.randn(B, seq_len, S, generator=g), "p2_speech": torch.randn(B, seq_len, S, generator=g)}

    def train_dataloader(self):
        return self._loader(self.hp.Train["seq_len"], self.steps, RANDOM_SEED)

    def val_dataloader(self):
        return self._loader(self.hp.Validation["seq_len"], 2, RANDOM_SEED + 1)


def main(argv=None):
    seed_everything(RANDOM_SEED)
    hparams, conf_name = get_hparams(argv)
    model = LetsFaceItGlow(hparams)
    if getattr(hparams, "synthetic", False):
        dm = SyntheticDataModule(hparams, steps=int(getattr(hparams, "synthetic_steps", 100)))
    elif getattr(hparams, "data_module", "") == "reference":
        try:
            from glow_pytorch.mimicry_data_module import MimicryDataModule  # the reference's data pipeline, untouched
        except ImportError as e:
            raise SystemExit("the reference's glow_pytorch.mimicry_data_module (and h5py) must be importable for "
                             "--data_module reference: %s" % e)
        dm = MimicryDataModule(hparams)
    else:
        from lets_face_it_amd.mimicry_data_module import MimicryDataModule  # same surface, corpus resident in HBM
        dm = MimicryDataModule(hparams)
    from lets_face_it_amd.mimicry_logger import MimicryLogger   # the callback the reference's train.py:33-37 registers
    trainer = Trainer(hparams, callbacks=[MimicryLogger()])
    resume = getattr(hparams, "resume_from_checkpoint", None)
    if resume:
        trainer.resume(model, resume)
    trainer.fit(model, dm)


if __name__ == "__main__":
    main()
