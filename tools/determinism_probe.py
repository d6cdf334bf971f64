"""Are repeated identical training passes bit-identical? Prints which engine buffers differ between passes.
Run on the GPU box:  python tools/determinism_probe.py [f32|bf16x3]"""
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
g = torch.Generator().manual_seed(5)
batch = {k: torch.randn(256, 80, d, generator=g).to(dev) for k, d in (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
snaps = []
for rep in range(4):
    m.zero_grad(set_to_none=True)
    _, loss, losses = m(batch)
    loss.sum().backward()
    torch.cuda.synchronize()
    eng = m.engine
    snap = {k: v.clone() for k, v in eng._ws.items() if not k.startswith("scratch")}
    snap["grads"] = eng.grads.clone()
    snaps.append(snap)
bad = {}
for r in range(1, 4):
    for k in snaps[0]:
        if not torch.equal(snaps[0][k], snaps[r][k]):
            bad.setdefault(k, []).append(int((snaps[0][k] != snaps[r][k]).sum()))
print("precision", m.precision, "- buffers that differ between pass 0 and passes 1..3 (count of unequal elements):")
for k in sorted(bad):
    print("   %-28s %s" % (k, bad[k]))
if not bad:
    print("   none: all passes bit-identical")
