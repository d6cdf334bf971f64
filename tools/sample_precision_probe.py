"""Which products of SeqGlow.inference need more than bf16x3 at FULL depth? (VERDICT r2 weak #1b)

At K = 16 flow steps x 56 generated frames plain fp32 torch is itself 2.4e-5 (absolute) from the fp64 oracle; the engine's
exact-f32 mode lands on that floor, its bf16x3 mode at 5.6e-5. This probe runs the K = 16 parity case with the STATIC part
(window encoders + the non-autoregressive cond_transform columns: throughput work) and the AUTOREGRESSIVE part (per generated
frame: window columns of cond_transform, gic, the recurrent products of the 16 reverse cells: latency-bound) in either
arithmetic, and times BASELINE configs[3] (batch 1024 x 300 frames) for each combination.

    python tools/sample_precision_probe.py > profiles/round3_sample_precision.md      (GPU box)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402


def main():
    from oracle import seqglow_oracle as oracle
    from test_gpu_parity import final_model_hparams, perturbed_model, to_dev
    dev = torch.device("cuda:0")
    hp = final_model_hparams(50, 27, K=16)
    m, sd = perturbed_model(hp, dev)
    m.eval()
    B, seq_len = 8, 24 + 56
    g = torch.Generator().manual_seed(3)
    data = {"p1_face": torch.randn(B, 24, 50, generator=g)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(seq_len - 24, B, 50, generator=g) * 0.8
    ref = oracle.seqglow_inference(hp, {k: v.double() for k, v in sd.items()}, seq_len, {k: v.double() for k, v in data.items()}, noise.double())
    own = float((oracle.seqglow_inference(hp, sd, seq_len, data, noise).double() - ref).abs().max())
    # configs[3]
    Bb, Tb = 1024, 300
    big = {"p1_face": torch.zeros(Bb, Tb, 50, device=dev)}
    for name, d in (("p2_face", 50), ("p1_speech", 27), ("p2_speech", 27)):
        big[name] = torch.randn(Bb, Tb, d, generator=g).to(dev)
    bnoise = torch.randn(Tb - 24, Bb, 50, generator=g).to(dev)
    print("# sampling: arithmetic of the static and the autoregressive part vs error at full depth and time of configs[3]\n")
    print("K = 16, final widths, batch 8, 56 generated frames, injected noise; plain fp32 torch (CPU) is %.2e from the fp64 oracle; "
          "gate = max(1e-5, 1.5 x that) = %.2e. Time: batch 1024 x 300 frames (276 generated), hipGraph replay, median of 5.\n" % (own, max(1e-5, 1.5 * own)))
    print("| static part | autoregressive part | max abs err vs fp64 oracle | ms per call (1024 x 300) |")
    print("|---|---|---|---|")
    modes = {"bf16x3 GEMMs + fp16x3 cells": 1, "f32": 0, "bf16x6 GEMMs + fp16x3 cells": 5, "fp16x3 GEMMs + fp16x3 cells": 9}
    for static, frame in (("bf16x3", "bf16x3 GEMMs + fp16x3 cells"), ("bf16x3", "f32"), ("bf16x3", "bf16x6 GEMMs + fp16x3 cells"),
                          ("bf16x3", "fp16x3 GEMMs + fp16x3 cells"),
                          ("f32", "bf16x3 GEMMs + fp16x3 cells"), ("f32", "f32")):
        m.precision = static
        eng = m._ensure_engine(dev)
        eng.sample_frame_precision = modes[frame]
        out = m.inference(seq_len, to_dev(data, dev), noise=noise.to(dev))
        err = float((out.cpu().double() - ref).abs().max())
        for _ in range(3):
            m.inference(Tb, big, noise=bnoise)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.inference(Tb, big, noise=bnoise)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("| %s | %s | %.2e | %.1f |" % (static, frame, err, 1e3 * sorted(ts)[2]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
