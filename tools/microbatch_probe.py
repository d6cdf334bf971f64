"""Would two half-batches stepping CONCURRENTLY on two streams beat one full batch? The flow walks are latency-bound chains that
hold one persistent workgroup per (flow step, 16-sample tile): at batch 128 they occupy half the CUs for the same 0.55 + 0.8 ms, and
the other half-batch's throughput-bound kernels (window encoders, GEMMs) could run on the other half meanwhile. Probe: two
independent models at batch 128 on two streams vs the same two steps back to back on one stream vs one model at batch 256.
    python tools/microbatch_probe.py            (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import copy
    import random
    from argparse import Namespace
    import numpy as np
    from bench import synthetic_batch
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    hp = load_hparams_file(os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    hp["engine_precision"] = "bf16x3"
    hp["engine_backward_products"] = 2
    dev = torch.device("cuda:0")

    def make(B):
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        h = copy.deepcopy(hp); h["batch_size"] = B
        m = LetsFaceItGlow(Namespace(**h)).to(dev).train()
        return m

    def timeit(fn, n=20):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    full = make(256)
    b256 = synthetic_batch(256, 80, 50, 27, 1, dev)
    t_full = timeit(lambda: full.fused_training_step(b256, 1e-5))
    del full
    torch.cuda.empty_cache()
    a, b = make(128), make(128)
    ba, bb = synthetic_batch(128, 80, 50, 27, 2, dev), synthetic_batch(128, 80, 50, 27, 3, dev)
    t_seq = timeit(lambda: (a.fused_training_step(ba, 1e-5), b.fused_training_step(bb, 1e-5)))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def conc():
        with torch.cuda.stream(sa):
            a.fused_training_step(ba, 1e-5)
        with torch.cuda.stream(sb):
            b.fused_training_step(bb, 1e-5)

    t_conc = timeit(conc)
    print("one model, batch 256:                          %.3f ms per 256 x 56 frames" % t_full)
    print("two models, batch 128 each, back to back:      %.3f ms" % t_seq)
    print("two models, batch 128 each, on two streams:    %.3f ms" % t_conc)


if __name__ == "__main__":
    main()
