#!/usr/bin/env python3
"""Full-size sampling call (1024 x 300) under the environment it is started with: one run in line, four runs on an ordinary second
stream, four runs with the static part on a partial-chip stream - which of them agree bit for bit?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from test_gpu_parity import final_model_hparams, perturbed_model, to_dev  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hp = final_model_hparams(50, 27)
    m, _ = perturbed_model(hp, dev)
    m.eval()
    B, T = 1024, 300
    g = torch.Generator().manual_seed(11)
    data = {"p1_face": torch.zeros(B, T, 50)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, T, d, generator=g)
    data = to_dev(data, dev)
    noise = (torch.randn(T - 24, B, 50, generator=g) * 0.8).to(dev)
    outs = {}
    for name, runs, cus in (("one run", "1", "0"), ("4 runs, second stream", "4", "0"), ("4 runs, 16 CUs/XCD", "4", "16"),
                            ("4 runs, second stream again", "4", "0"), ("4 runs, 16 CUs/XCD again", "4", "16"), ("4 runs, 8 CUs/XCD", "4", "8")):
        os.environ["LFI_SAMPLE_RUNS"], os.environ["LFI_SAMPLE_STATIC_CUS"] = runs, cus
        for rep in range(3):
            outs["%s #%d" % (name, rep)] = m.inference(T, data, noise=noise).clone()
    ref = outs["one run #0"]
    for k, v in outs.items():
        d = (v - ref).abs().amax(dim=(0, 2))
        bad = [i for i in range(d.numel()) if float(d[i]) != 0.0]
        print("%-36s %s" % (k, "equal" if not bad else "differs from frame %d on (max %.3e)" % (bad[0], float(d.max()))))


if __name__ == "__main__":
    main()
