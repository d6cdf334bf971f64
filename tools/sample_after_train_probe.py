#!/usr/bin/env python3
"""Round 6: bench.py's `sampling` sub-record (run in the process that has just trained) reads 50.8 ms where `bench.py --workload sample`
in a fresh process reads 45.3 on the same box. What in the process' history does that? Runs a small preamble, then the stand-alone
sampling bench in the SAME process.   python tools/sample_after_train_probe.py none|tiny_step|tiny_fwd|scratch_kernel"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
if mode == "tiny_fwd_1stream":
    os.environ["LFI_NO_OVERLAP"] = "1"
if mode == "stream_use":      # no model at all: one pool stream that has carried a kernel
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1.0)
    torch.cuda.synchronize()
if mode.startswith("primed"):
    # the sampler's two streams exist and have carried a kernel BEFORE anything forks off the default stream; every engine then uses them
    from lets_face_it_amd.engine import GlowEngine
    S1, S2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for st in (S1, S2):
        with torch.cuda.stream(st):
            torch.zeros(16, device=dev).add_(1.0)
    torch.cuda.synchronize()
    _orig = GlowEngine.__init__

    def _init(self, *a, **k):
        _orig(self, *a, **k)
        self._sample_stream, self._side_stream = S1, S2
    GlowEngine.__init__ = _init
    st = S2 if mode == "primed_fork_same" else torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    x = torch.zeros(1 << 16, device=dev)
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        x.add_(1.0)
    cur.wait_stream(st)
    x.add_(1.0)
    torch.cuda.synchronize()
if mode in ("fork_join", "fork_join_x3"):     # what GlowEngine._fork / _join do around side work, with nothing else
    for _ in range(3 if mode.endswith("x3") else 1):
        st = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        x = torch.zeros(1 << 16, device=dev)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            x.add_(1.0)
        cur.wait_stream(st)
        x.add_(1.0)
    torch.cuda.synchronize()
if mode in ("tiny_step", "tiny_fwd", "mid_step", "tiny_fwd_1stream"):
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    fx = Fixture("mid" if mode == "mid_step" else "tiny")
    hp = fx.hp
    hp["Train"]["use_negative_nll_loss"] = False
    m = LetsFaceItGlow(Namespace(**hp))
    m.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    m.to(dev).train()
    m.seq_glow.glow.set_actnorm_init(True)
    batch = {k: v.to(dev) for k, v in fx.batch(torch.float32).items()}
    if mode in ("tiny_fwd", "tiny_fwd_1stream"):
        with torch.no_grad():
            m.seq_glow(batch)
    else:
        m.fused_training_step(batch, 1e-3)
    torch.cuda.synchronize()
    del m
    os.environ.pop("LFI_NO_OVERLAP", None)
elif mode == "scratch_kernel":
    # any kernel with a private segment: torch's own sort uses scratch on this arch
    x = torch.randn(1 << 20, device=dev)
    torch.sort(x)
    torch.cuda.synchronize()
sys.argv = ["bench.py", "--workload", "sample", "--quick", "--no-gpu-state", "--steps", "5", "--warmup", "3"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
