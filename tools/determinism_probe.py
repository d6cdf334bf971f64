"""Are repeated identical training passes bit-identical? Prints which engine buffers differ between passes.
Run on the GPU box:  python tools/determinism_probe.py [f32|bf16x3] [batch] [train|eval] [passes]
(batch 256 = the headline shape, 40 = a ragged one that takes the 32-window encoder kernels; train = with dropout masks - the
same key every pass; LFI_ENC_WIDE=0 forces the accumulator-layout fused encoder kernels, the ones the SLP note in csrc/Makefile is about)"""
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lets_face_it_amd.glow.models import SeqGlow  # noqa: E402
from lets_face_it_amd.glow.utils import load_hparams_file  # noqa: E402

dev = torch.device("cuda:0")
hp = load_hparams_file(os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
torch.manual_seed(1234)
np.random.seed(1234)
m = SeqGlow(Namespace(**hp))
g = torch.Generator().manual_seed(4321)
with torch.no_grad():   # LinearZeros is zero at init: perturb so that the conditioning path is live
    for name, p in m.named_parameters():
        if "final_linear" in name:
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
        elif "actnorm" in name:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
m.to(dev).eval()
m.glow.set_actnorm_init(True)
if len(sys.argv) > 1:
    m.precision = sys.argv[1]
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if len(sys.argv) > 3 and sys.argv[3] == "train":
    m.train()
    gm = torch.Generator().manual_seed(9)
    m.injected_masks = {k: ((torch.rand(56, BATCH, h, generator=gm) >= p).float() / (1.0 - p)).to(dev)
                        for k, h, p in (("p2_face", 24, 0.6), ("p1_speech", 2, 0.5), ("p2_speech", 16, 0.3))}
PASSES = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g = torch.Generator().manual_seed(5)
batch = {k: torch.randn(BATCH, 80, d, generator=g).to(dev) for k, d in (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
snaps = []
for rep in range(PASSES):
    m.zero_grad(set_to_none=True)
    _, loss, losses = m(batch)
    loss.sum().backward()
    torch.cuda.synchronize()
    eng = m.engine
    snap = {k: v.clone() for k, v in eng._ws.items() if not k.startswith("scratch")}
    snap["grads"] = eng.grads.clone()
    snaps.append(snap)
bad = {}
for r in range(1, PASSES):
    for k in snaps[0]:
        # (bit patterns, not values: a workspace's never-written slack may hold NaN patterns, and NaN != NaN)
        bits = {2: torch.int16, 4: torch.int32, 8: torch.int64}.get(snaps[0][k].element_size())
        a, b = (snaps[0][k].view(bits), snaps[r][k].view(bits)) if bits is not None else (snaps[0][k], snaps[r][k])
        if not torch.equal(a, b):
            bad.setdefault(k, []).append(int((a != b).sum()))
print("library", os.environ.get("LFI_LIB_PATH", "(tree)"), "LFI_ENC_WIDE", os.environ.get("LFI_ENC_WIDE", "-"), "precision", m.precision, "batch", BATCH,
      "train" if m.training else "eval", "- buffers that differ between pass 0 and passes 1..%d (count of unequal elements):" % (PASSES - 1))
for k in sorted(bad):
    print("   %-28s %s" % (k, bad[k]))
if not bad:
    print("   none: all passes bit-identical")
