#!/usr/bin/env python3
"""Round 6: is a replayed training step bit-reproducible on a small fixture? ONE process, the `mid` fixture (f32 engine mode, the engine's
own dropout masks), 8 optimiser steps, three ways: eager launches; the step as ONE replayed hipGraph (LetsFaceItGlow.step_graph, world 1);
the data-parallel form as TWO replayed graphs with a stand-in all-reduce between them (doubles the gradient, as a second identical rank
would; the optimiser divides by world = 2). Prints the parameter checksum of every leg, `repeats` times.
   python tools/graph_flake_probe.py [repeats] [fixture]"""
import os
import random
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import Fixture  # noqa: E402


def leg(mode, fx, dev, steps=8):
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    hp = fx.hp
    hp["gradient_clip_val"] = 20
    hp["Train"]["use_negative_nll_loss"] = False
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    m = LetsFaceItGlow(Namespace(**hp))
    m.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    m.to(dev).train()
    m.seq_glow.precision = os.environ.get("PROBE_PRECISION", "f32")
    m.step_graph = mode in ("graph", "dp_graph")
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(7)
    B = 16

    def fake_allreduce(t, async_op=False):
        t.mul_(2.0)
        return None

    for step in range(steps):
        batch = {k: torch.randn(B, fx.T, d, generator=g).to(dev) for k, d in
                 (("p1_face", fx.C), ("p2_face", fx.C), ("p1_speech", fx.S), ("p2_speech", fx.S))}
        if mode in ("dp_eager", "dp_graph"):
            m.fused_training_step(batch, 1e-3, 2, fake_allreduce)
        else:
            m.fused_training_step(batch, 1e-3)
    torch.cuda.synchronize()
    return float(m.seq_glow.engine.params.double().sum())


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    fx = Fixture(sys.argv[2] if len(sys.argv) > 2 else "mid")
    dev = torch.device("cuda:0")
    for r in range(reps):
        out = {mode: leg(mode, fx, dev) for mode in ("eager", "graph", "dp_eager", "dp_graph")}
        print("rep %d: " % r + "  ".join("%s %.12e" % kv for kv in out.items()) +
              ("   OK" if out["eager"] == out["graph"] and out["dp_eager"] == out["dp_graph"] else "   MISMATCH"), flush=True)


if __name__ == "__main__":
    main()
