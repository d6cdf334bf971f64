"""Phase stamps of enc_gru_fwd_ws_kernel (a -DLFI_WS_STAMPS build of lfi_encoder.hip, loaded through LFI_LIB_PATH): s_memtime
sums of wave 1 of the workgroup with ticket 40 over the iterations (chunk x history step, steps >= 1) of the LAST encoder launch
of a forward pass (p2_speech). Build: tools/build_enc_variant.sh ws_stamps "-DLFI_WS_STAMPS"."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from lets_face_it_amd import _lib
from lets_face_it_amd.glow.models import SeqGlow
from lets_face_it_amd.glow.utils import load_hparams_file
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hp = load_hparams_file(os.path.join(root, "lets_face_it_amd/hparams/final_model_synthetic.yaml"))
dev = torch.device("cuda:0")
m = SeqGlow(Namespace(**hp)).to(dev)
m.glow.set_actnorm_init(True)
m.train()
g = torch.Generator().manual_seed(0)
batch = {k: torch.randn(256, 80, d, generator=g).to(dev) for k, d in (("p1_face", 50), ("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27))}
for _ in range(3):
    m(batch)
st = torch.zeros(8192 + 64, dtype=torch.int64, device=dev)
_lib.lib().lfi_debug_set_stamps(st.data_ptr())
m(batch)
torch.cuda.synchronize()
_lib.lib().lfi_debug_set_stamps(None)
v = st.cpu()[256:265].tolist()
n = max(v[8], 1)
print("ticks per iteration (%d iterations):" % n)
tot = 0
for nm, k in (("barrier", 7), ("MFMA phase + transposes", 2), ("counted vmcnt wait", 3), ("publish + DMA issue + peek", 4), ("gate math", 5),
              ("next loads issue", 6), ("stash + state stores", 0), ("exchange stores + loop edge", 1)):
    print("  %-30s %8.0f" % (nm, v[k] / n))
    tot += v[k] / n
print("  %-30s %8.0f" % ("total", tot))
