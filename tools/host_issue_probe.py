"""Host cost of issuing ONE training step (queue empty at the start, nothing awaited) next to the GPU time of that step, and
where the host time goes (cProfile, top entries): is the step launch-bound?"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from argparse import Namespace
from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
from lets_face_it_amd.glow.utils import load_hparams_file
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hp = load_hparams_file(os.path.join(root, "lets_face_it_amd/hparams/final_model_synthetic.yaml"))
dev = torch.device("cuda:0")
model = LetsFaceItGlow(Namespace(**hp)).to(dev)
model.train()
g = torch.Generator().manual_seed(0)
batch = {k: torch.randn(256, 80, d, generator=g).to(dev) for k, d in (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
for _ in range(5):
    model.fused_training_step(batch, 1e-4)
torch.cuda.synchronize()
host, gpu = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fused_training_step(batch, 1e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    gpu.append(t2 - t0)
print("host issue per step: median %.2f ms (min %.2f); issue + drain: median %.2f ms" % (
    1e3 * sorted(host)[5], 1e3 * min(host), 1e3 * sorted(gpu)[5]))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    model.fused_training_step(batch, 1e-4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
