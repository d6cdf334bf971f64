"""End-to-end throughput of the REAL training loop (lets_face_it_amd.trainer.Trainer.fit over the GPU-resident
MimicryDataModule / WindowLoader on a synthetic corpus of the reference's HDF5 layout), next to bench.py's resident-batch step:
does anything in the loop around the step (sampler, batch gather, logging) hold the GPU up?
usage: python tools/train_loop_probe.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import numpy as np
import torch

from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
from lets_face_it_amd.glow.utils import load_hparams_file
from lets_face_it_amd.mimicry_data_module import MimicryDataModule
from lets_face_it_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hp = load_hparams_file(os.path.join(root, "lets_face_it_amd/hparams/final_model_synthetic.yaml"))
hp["max_steps"], hp["max_epochs"] = steps, 1000
rng = np.random.default_rng(0)
dims = {"flame_expression": 100, "flame_jaw": 3, "flame_neck": 3, "mfcc": 25, "prosody": 2}
hp["Data"]["expression_dim"] = hp["Conditioning"]["p1_face"]["dim"] - 6   # face = expression[:dim] | jaw (3) | neck (3)
sd = hp["Data"]["speech_dim"]
dims["mfcc"] = sd - dims["prosody"]
store = {split: {k: {str(i): {w: rng.standard_normal((3000, d)).astype(np.float32) for w in ("agent", "interlocutor")}
                     for i in range(12 if split == "train" else 1)} for k, d in dims.items()} for split in ("train", "val", "test")}
hparams = Namespace(**hp)
dm = MimicryDataModule(hparams, device="cuda:0", source=store)
model = LetsFaceItGlow(hparams)
tr = Trainer(hparams, device="cuda:0", log_every=10 ** 9, checkpoint_dir="")
tr.validate = lambda *a, **k: None
# warm-up (allocations, ActNorm init), then the timed run
hparams.max_steps = 10
tr.max_steps = 10
tr.fit(model, dm)
torch.cuda.synchronize()
tr.max_steps = 10 + steps
t0 = time.perf_counter()
tr.fit(model, dm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
B, T = hparams.batch_size, hparams.Train["seq_len"]
N = T - model.seq_glow.spec.start
done = tr.global_step - 10 if tr.global_step > 10 else steps
print("trainer loop: %d steps in %.3f s = %.3f ms per step = %.0f frames/s (batch %d, T %d)" % (
    done, dt, 1e3 * dt / done, done * B * N / dt, B, T))
