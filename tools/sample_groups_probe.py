#!/usr/bin/env python3
"""Sampler, configs[3] (batch 1024 x 300 frames): do independent SAMPLE GROUPS on separate streams fill the idle time of the
per-frame dependent chain? Sequences of a batch never interact (SeqGlow.inference, glow/models.py:567-596), so G copies of the model,
each sampling B / G sequences on its own stream, compute what one call on B sequences computes:
   python tools/sample_groups_probe.py [--groups 1 2 4 8]
prints ms per (whole-batch) call for each G; G = 1 is the shipped path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from argparse import Namespace
    from lets_face_it_amd.glow.models import SeqGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    hp = load_hparams_file(os.path.join(os.path.dirname(__file__), "..", "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    dev = torch.device("cuda:0")
    B, T, C, S = a.batch, a.seq_len, 50, 27
    g = torch.Generator().manual_seed(1234)
    data = {"p1_face": torch.zeros(B, T, C, device=dev)}
    for name, d in (("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        data[name] = torch.randn(B, T, d, generator=g).to(dev).contiguous()
    nframes = T - 24
    noise = torch.randn(nframes, B, C, generator=g).to(dev).contiguous()
    torch.manual_seed(1234)
    base = SeqGlow(Namespace(**hp)).to(dev)
    base.glow.set_actnorm_init(True)
    base.eval()
    sd = base.state_dict()
    ref = None
    for G in a.groups:
        models = [base]
        for _ in range(G - 1):
            m = SeqGlow(Namespace(**hp)).to(dev)
            m.load_state_dict(sd)
            m.glow.set_actnorm_init(True)
            m.eval()
            models.append(m)
        Bg = B // G
        streams = [torch.cuda.Stream() for _ in range(G)]
        subs = [({k: v[i * Bg:(i + 1) * Bg].contiguous() for k, v in data.items()}, noise[:, i * Bg:(i + 1) * Bg].contiguous()) for i in range(G)]
        outs = [None] * G

        def call():
            cur = torch.cuda.current_stream()
            for i, m in enumerate(models):
                streams[i].wait_stream(cur)
                with torch.cuda.stream(streams[i]):
                    outs[i] = m.inference(T, subs[i][0], noise=subs[i][1])
            for s_ in streams:
                cur.wait_stream(s_)

        for _ in range(3):      # (the second call of a shape captures the per-frame sequences as hipGraphs)
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        out = torch.cat(outs, 0)
        if ref is None:
            ref = out
        print("groups %d (batch %4d each): %.2f ms per %d x %d call   max |diff| vs one group %.2e" %
              (G, Bg, e0.elapsed_time(e1) / a.reps, B, T, float((out - ref).abs().max())), flush=True)
        del models[1:]


if __name__ == "__main__":
    main()
