"""Are repeated identical training passes bit-identical? Prints which engine buffers differ between passes.
Run on the GPU box:  python tools/determinism_probe.py [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_parity import final_model_hparams, perturbed_model, to_dev

dev = torch.device("cuda:0")
hp_ = final_model_hparams(50, 27)
g = torch.Generator().manual_seed(5)
batch = to_dev({k: torch.randn(256, 80, d, generator=g) for k, d in (('p1_face', 50), ('p2_face', 50), ('p1_speech', 27), ('p2_speech', 27))}, dev)
m = perturbed_model(hp_, dev)[0].eval()
if len(sys.argv) > 1:
    m.precision = sys.argv[1]
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
