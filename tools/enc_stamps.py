"""Phase stamps of enc_gru_fwd_r64_kernel (a -DLFI_ENC_STAMPS build of lfi_encoder.hip, loaded through LFI_LIB_PATH): s_memtime
sums of waves 0 and 5 of workgroup 100 over the history steps of the LAST encoder launch of a forward pass (p2_speech)."""
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
v = st.cpu()[256:272].tolist()
for w, o in ((0, v[:8]), (5, v[8:16])):
    n = max(o[5], 1)
    print("wave %d: cycles per history step: recurrent product (to vmcnt 0) %d | epilogue to barrier 1 %d | barrier 1 %d | epilogue tail %d | "
          "barrier 2 %d | steps %d, clock %.2f GHz" % (w, o[0] / n, o[1] / n, o[2] / n, o[3] / n, o[4] / n, n, o[6] / max(o[7], 1) * 0.1))
