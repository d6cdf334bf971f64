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
        ns = Namespace(**h)
        m = LetsFaceItGlow(ns).to(dev).train()
        from lets_face_it_amd.trainer import Trainer
        tr = Trainer(ns, device=dev)            # as bench.py: without these hooks the negative-example switch syncs the host every step
        m.seq_glow.allreduce_hook = tr.allreduce_stats
        m.nll_sync_hook = tr.sync_scalar
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
    sys.stdout.flush()
    # the same with the chip PARTITIONED: each model's streams (main + the engine's second stream) may only use one half of the
    # CUs (hipExtStreamCreateWithCUMask; alternate CUs, so both halves span every XCD), so that one model's latency-bound walks
    # (128 workgroups at batch 128) and the other's throughput-bound kernels really run side by side
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def masked(word):
        h = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * 8)(*([word] * 8))
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, arr)
        if rc != 0:
            raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % rc)
        return torch.cuda.ExternalStream(h.value, device=dev)

    for name, wa, wb in (("alternate CUs", 0x55555555, 0xAAAAAAAA), ("low / high 16 of every 32", 0x0000FFFF, 0xFFFF0000)):
        ma, mb = masked(wa), masked(wb)
        a.seq_glow.engine._side_stream, b.seq_glow.engine._side_stream = masked(wa), masked(wb)

        def conc_masked():
            with torch.cuda.stream(ma):
                a.fused_training_step(ba, 1e-5)
            with torch.cuda.stream(mb):
                b.fused_training_step(bb, 1e-5)

        print("two models, batch 128 each, two CU-masked stream pairs (%s): %.3f ms" % (name, timeit(conc_masked)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
