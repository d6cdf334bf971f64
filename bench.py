"""Headline benchmark: FLAME frames/s of one full training step (forward + backward + gradient all-reduce + clip +
Adam) of final_model.yaml at BASELINE.json's synthetic dims (50-d FLAME + 27-d speech), T = 80, batch 256 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `value` = frames processed by all ranks / max-over-ranks wall time of the K timed steps
(inputs resident in HBM before the timed region). `roofline` is for the dominant kernel, the cond_transform GEMM
(F x Ks*D x E on the f32 MFMA), timed with HIP events on its launch stream inside the timed region. `cpu_baseline`
(N = 1 only) times the CPU oracle — a plain-PyTorch port of the reference's per-timestep loop — on the host cores on a
bounded sample of the same workload; it is a reported baseline, not the thing measured above.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (no 2:1 sparsity)


def synthetic_batch(B, T, C, S, seed, device):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, d in (("p1_face", C), ("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        out[name] = torch.randn(B, T, d, generator=g, dtype=torch.float32).to(device).contiguous()
    return out


def cpu_baseline(hp, C, S, T, budget_s):
    """The oracle (kind "port") on the host cores: forward + backward of the reference's op sequence."""
    from oracle import seqglow_oracle as oracle
    from lets_face_it_amd.glow.models import SeqGlow
    from argparse import Namespace
    import copy
    # the per-timestep loop is made of small ops: on a many-core host the default (all cores) is far slower than a
    # modest team, so use at most 16 threads and report that count
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    m = SeqGlow(Namespace(**copy.deepcopy(hp)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")):
            v.requires_grad_(True)
    B = 16
    batch = oracle.synthetic_batch(B, T, C, S, seed=1234)
    start = oracle.longest_history(hp["Conditioning"])
    frames = B * (T - start)

    def step():
        for v in sd.values():
            v.grad = None
        loss = oracle.seqglow_forward(hp, sd, batch)[1]
        loss.sum().backward()

    t0 = time.time()
    step()  # warm-up
    warm = time.time() - t0
    times = []
    while len(times) < 2 or (sum(times) + warm < budget_s and len(times) < 5):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": frames / med, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "oracle/seqglow_oracle.py fwd+bwd (torch CPU fp32, per-timestep loop as the reference), "
                      "final_model C=%d S=%d T=%d at batch %d, median of %d steps after 1 warm-up" % (C, S, T, B, len(times))}


def _timed(fn, steps, world, device):
    """EXACTLY `steps` calls bracketed by barrier + synchronize on both sides; max over ranks (seconds)."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, last


def _roofline(spec, F, timing, precision):
    """Dominant kernel = the cond_transform forward GEMM (F x Ks*D x Ef). SURVEY.md par. 8d counts 2*E*D per flow step and
    frame as written (E = 1530: every GRU output twice); the kernel runs on the folded layout (Ef = 890).
    `achieved` is the product's FLOPs (2 M N K on the folded K) over the HIP-event launch time. In bf16x3 mode every
    product costs three bf16 MFMAs, so the MFMA pipe does 3x that work: both fractions are reported."""
    KD = spec.Ks * spec.D
    flops_alg = 2.0 * F * KD * spec.E
    flops = 2.0 * F * KD * spec.Ef
    n_launch, ms = timing.get("gemm_cond_fwd", (0, float("nan")))
    ach = flops / (ms * 1e-3) / 1e12 if n_launch else float("nan")
    # which kernel lfi_gemm_f32 picks for this product: bf16x3 takes the 256 x 256 tile (1024 threads) unless
    # LFI_GEMM_256=0, f32 the 128 x 128 tile (256 threads); the rocprof kernel names below are what --kernel-trace prints
    if precision == "bf16x3":
        big = os.environ.get("LFI_GEMM_256", "1") != "0"
        peak, mult = BF16_MFMA_PEAK_TFLOPS, 3.0
        k32 = os.environ.get("LFI_GEMM_K32", "0") == "1"   # the opt-in 32-k variant of the 256 x 256 kernel
        kern = ("gemm_bf16x3_256k_kernel<true, true>" if k32 else "gemm_bf16x3_256_kernel<true, true>") if big \
            else "gemm_bf16x3_kernel<true, true>"
        tile, threads = (256, 1024) if big else (128, 256)
    else:
        peak, kern, mult = F32_MFMA_PEAK_TFLOPS, "gemm_f32_kernel<128, 128, 2, 2, true, true", 1.0
        tile, threads = 128, 256
    # HBM bytes per launch of that kernel from the PMC passes of the same command (tools/pmc_bench.sh -> profiles/):
    # bench.py cannot collect counters itself; null when no committed measurement matches this kernel and grid
    traffic, traffic_src = None, None
    try:
        tiles = ((F + tile - 1) // tile) * ((KD + tile - 1) // tile)
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % precision)))
        for name, v in tj["kernels"].items():
            if kern in name and name.endswith("grid=%d" % (tiles * threads)):
                traffic, traffic_src = v["hbm_bytes"], "profiles/pmc_traffic_%s.json (%s)" % (precision, tj["source"])
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "mfma", "kernel": kern + " cond_transform forward (F x Ks*D x Ef)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
            "traffic_unit": "bytes per launch (HBM read + write)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": 4.0 * (F * spec.ldf + KD * spec.ldf + F * KD),
            "mfma_flops_multiplier": mult, "frac_of_mfma_issue": mult * ach / peak,
            "flops_per_launch": flops, "flops_per_launch_algorithmic": flops_alg,
            "achieved_algorithmic": flops_alg / (ms * 1e-3) / 1e12 if n_launch else None,
            "ms_per_launch": ms, "launches_timed": n_launch}


def _roofline_hbm(spec, B, T, timing, precision):
    """Second roofline, for the largest HBM-bound kernel: the fused GRU window encoder of p2_face, forward
    (enc_gru_fwd_fused_kernel). Algorithmic bytes per launch = what must cross HBM once: the BPTT stash it writes (r, z, n,
    W_hn h + b_hn and h: 5 * hid floats per window and history step), the projected inputs it reads (B*T x 3*hid, shared by the
    overlapping windows) and the feature block it writes (F x hid). Launch time: HIP events on the launch stream."""
    e = next((x for x in spec.encoders if x.name == "p2_face" and x.enc == "rnn"), None)
    n_launch, ms = timing.get("enc_fwd.p2_face", (0, float("nan")))
    if e is None or not n_launch:
        return None
    F = B * (T - spec.start)
    alg = 4.0 * (e.hist * F * 5 * e.hid + B * T * 3 * e.hid + F * e.hid)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % precision)))
        for name, v in tj["kernels"].items():
            if "enc_gru_fwd_fused_kernel" in name and name.endswith("grid=%d" % (((F + 31) // 32) * 256)):
                traffic = max(traffic or 0.0, v["hbm_bytes"])   # p2_face (24 steps) is the larger of the two 256-wide launches
    except (OSError, ValueError, KeyError):
        pass
    ach = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "enc_gru_fwd_fused_kernel, p2_face windows (hist %d, hid %d)" % (e.hist, e.hid),
            "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
            "traffic_note": "PMC bytes per launch of this kernel name and grid: the mean over the p2_face (24 steps) and p2_speech "
                            "(16 steps) launches, which share them",
            "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "launches_timed": n_launch}


def bench_train(args, model, trainer, spec, device, world, rank, hp):
    C, S, T, B = spec.C, spec.S, args.seq_len, args.batch
    N = T - spec.start
    batches = [synthetic_batch(B, T, C, S, 1234 + 1000 * rank + i, device) for i in range(2)]
    lr = trainer.lr_at(0)
    allreduce = trainer.allreduce_grads if world > 1 else None

    def step(i):
        return model.fused_training_step(batches[i & 1], lr, world, allreduce)

    for i in range(args.warmup):  # includes the one-off ActNorm data-dependent init
        step(i)
    eng = model.seq_glow.engine
    eng.enable_timing(True)
    elapsed, loss = _timed(step, args.steps, world, device)
    timing = eng.timing_summary()
    eng.enable_timing(False)
    if rank != 0:
        return None
    frames = world * B * N * args.steps
    F = B * N
    out = {
        "metric": "FLAME frames/s, full training step (fwd+bwd+clip+Adam), final_model.yaml batch 256 per GPU",
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f32 (GEMM operands split into bf16 hi+lo, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "final_model.yaml training step, synthetic 50-d FLAME + 27-d speech, T=%d, "
                               "batch %d per GPU (BASELINE.json configs[1]%s)" % (T, B, "" if world == 1 else ", data-parallel"),
                   "K": spec.Ks, "H": spec.H, "cond_dim": spec.D, "feature_dim": spec.E, "frames_per_step_per_gpu": F,
                   "parallelism": "dp%d" % world, "params": eng.n_params, "gemm_precision": args.precision},
        "final_loss": float(loss),
        "roofline": _roofline(spec, F, timing, args.precision),
        "roofline_hbm": _roofline_hbm(spec, B, T, timing, args.precision),
        "kernel_timing": {t: {"launches": n, "ms": round(m_, 4)} for t, (n, m_) in timing.items()},
    }
    if world == 1 and args.cpu_baseline_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(hp, C, S, T, args.cpu_baseline_seconds)
    return out


def bench_sample(args, model, spec, device, world, rank):
    """BASELINE.json configs[3]: SeqGlow.inference, batch 1024, seq_len 300, seed frames zeros, eps 1."""
    B = args.batch if args.batch != 256 else 1024
    T = args.seq_len if args.seq_len != 80 else 300
    C, S = spec.C, spec.S
    model.eval()
    model.seq_glow.glow.set_actnorm_init(True)
    g = torch.Generator().manual_seed(1234 + rank)
    data = {"p1_face": torch.zeros(B, T, C, device=device)}
    for name, d in (("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        data[name] = torch.randn(B, T, d, generator=g).to(device).contiguous()
    nframes = T - spec.start
    noise = torch.randn(nframes, B, C, generator=g).to(device).contiguous()

    def step(i):
        return model.seq_glow.inference(T, data, noise=noise)

    for i in range(args.warmup):
        step(i)
    elapsed, out_faces = _timed(step, args.steps, world, device)
    if rank != 0:
        return None
    frames = world * B * nframes * args.steps
    return {
        "metric": "FLAME frames/s, autoregressive sampling (SeqGlow.inference), final_model.yaml",
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f32 (GEMM operands split into bf16 hi+lo, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "autoregressive sampling, batch %d, seq_len %d (%d generated frames per sequence), "
                               "BASELINE.json configs[3]" % (B, T, nframes), "K": spec.Ks, "H": spec.H,
                   "parallelism": "replicas%d" % world, "gemm_precision": args.precision},
        "ms_per_generated_frame": 1e3 * elapsed / args.steps / nframes,
        "finite": bool(torch.isfinite(out_faces).all()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--seq-len", type=int, default=80)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=25.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--precision", choices=("f32", "bf16x3"), default=os.environ.get("LFI_PRECISION", "bf16x3"),
                    help="GEMM arithmetic: exact f32 MFMA, or bf16 hi/lo split operands (3 bf16 MFMAs per product)")
    ap.add_argument("--workload", choices=("train", "sample"), default="train",
                    help="train: BASELINE.json configs[1] (the headline metric); sample: configs[3], autoregressive "
                         "inference at --batch 1024 --seq-len 300 unless given")
    ap.add_argument("--hparams", default=os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # LFI_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share cards, tensors travel
    # through the host): same code path, barriers and timing; the default is RCCL with one GPU per rank
    backend = os.environ.get("LFI_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    device = torch.device("cuda", local_rank % ndev if (backend != "nccl" and ndev > 0) else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    import random
    import numpy as np
    from argparse import Namespace
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    from lets_face_it_amd.glow.utils import load_hparams_file
    from lets_face_it_amd.trainer import Trainer

    hp = load_hparams_file(args.hparams)
    hp["batch_size"] = args.batch
    hp["engine_precision"] = args.precision
    hp["Train"]["seq_len"] = args.seq_len
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    ns = Namespace(**hp)
    model = LetsFaceItGlow(ns)
    model.to(device)
    model.train()
    trainer = Trainer(ns, device=device)
    model.seq_glow.allreduce_hook = trainer.allreduce_stats
    model.nll_sync_hook = trainer.sync_scalar
    trainer.broadcast_parameters(model)

    spec = model.seq_glow.spec
    if args.workload == "sample":
        out = bench_sample(args, model, spec, device, world, rank)
    else:
        out = bench_train(args, model, trainer, spec, device, world, rank, hp)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
