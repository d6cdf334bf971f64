"""Headline benchmark: FLAME frames/s of one full training step (forward + backward + gradient all-reduce + clip +
Adam) of final_model.yaml at BASELINE.json's synthetic dims (50-d FLAME + 27-d speech), T = 80, batch 256 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `value` = frames processed by all ranks / max-over-ranks wall time of the K timed steps
(inputs resident in HBM before the timed region). `roofline` is for the step's largest kernel by time, the persistent GRU
window-encoder forward recurrence (two launches per step), timed with HIP events on its launch stream inside the timed region;
`roofline_best_gemm` is the cond_transform forward product (F x Ks*D x Ef, bf16 hi/lo operand planes, three bf16 MFMA products per
k-step: the fastest kernel of the step, ~5 % of its time); `roofline_whole_step` prices the whole step. `cpu_baseline`
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


def _oracle_state(hp):
    """Random-init final_model weights as an oracle state_dict with gradients enabled (the reference's `model.parameters()`)."""
    from lets_face_it_amd.glow.models import SeqGlow
    from argparse import Namespace
    import copy
    torch.manual_seed(1234)
    m = SeqGlow(Namespace(**copy.deepcopy(hp)))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def _oracle_train_steps(hp, sd, batch, budget_s, min_steps, max_steps, sync=None, reference_ops=False):
    """1 warm-up + up to max_steps timed forward+backward passes of the oracle (at least min_steps; stops early once the
    budget is spent). reference_ops: the cells as the fused ops nn.GRU / nn.GRUCell dispatch to (oracle.reference_op_set) instead
    of their spelled-out equations. -> (median seconds, number of timed steps)"""
    from oracle import seqglow_oracle as oracle
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")):
            v.requires_grad_(True)

    def step():
        for v in sd.values():
            v.grad = None
        with oracle.reference_op_set(reference_ops):
            loss = oracle.seqglow_forward(hp, sd, batch)[1]
            loss.sum().backward()
        if sync is not None:
            sync()

    t0 = time.time()
    step()  # warm-up
    spent = time.time() - t0
    times = []
    while len(times) < min_steps or (spent < budget_s and len(times) < max_steps):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
        spent += times[-1]
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline(hp, C, S, T, B, budget_s, note=""):
    """The oracle (kind "port") on the host cores: forward + backward of the reference's op sequence (per-timestep loop, the
    cells through the fused ops nn.GRU / nn.GRUCell dispatch to) at the metric's own batch (256): ~7-25 s per step, so 1 warm-up
    + 3 timed steps by default (--cpu-baseline-seconds bounds it)."""
    from oracle import seqglow_oracle as oracle
    # the per-timestep loop is made of small ops: on a many-core host the default (all cores) is far slower than a
    # modest team, so use at most 16 threads and report that count
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = _oracle_state(hp)
    batch = oracle.synthetic_batch(B, T, C, S, seed=1234)
    frames = B * (T - oracle.longest_history(hp["Conditioning"]))
    med, n = _oracle_train_steps(hp, sd, batch, budget_s, 2, 3, reference_ops=True)
    return {"value": frames / med, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "oracle/seqglow_oracle.py fwd+bwd (torch CPU fp32, per-timestep loop and fused GRU ops as the reference), "
                      "%d flow steps, C=%d S=%d T=%d at batch %d%s, median of %d steps after 1 warm-up, "
                      "%.1f s per step" % (oracle.n_flow_steps(hp), C, S, T, B, note or (" (the metric's batch)" if B == 256 else ""), n, med)}


def torch_gpu_baseline(hp, C, S, T, B, device, budget_s):
    """BASELINE.md par. 3: "the same restatement on one MI355X through stock PyTorch-ROCm = the reference single-GPU
    PyTorch frames/s that the >= 10x target is measured against". The oracle's op sequence (per-timestep Python loop,
    per-flow-step small ATen ops, autograd backward; models.py:534-561) with every tensor on the GPU; eager mode, fp32, no
    custom kernels — nothing of lets_face_it_amd's HIP library runs in this leg."""
    from oracle import seqglow_oracle as oracle
    sd = {k: v.to(device) for k, v in _oracle_state(hp).items()}
    batch = {k: v.to(device) for k, v in oracle.synthetic_batch(B, T, C, S, seed=1234).items()}
    frames = B * (T - oracle.longest_history(hp["Conditioning"]))
    # the reference's own op set: nn.GRU -> torch._VF.gru (MIOpen RNN), nn.GRUCell -> torch._VF.gru_cell (two addmm
    # + one fused pointwise kernel); models.py:21-27,60-64,176-179,206-208. This is north_star's denominator.
    med, n = _oracle_train_steps(hp, sd, batch, budget_s / 2, 3, 5, sync=torch.cuda.synchronize, reference_ops=True)
    # for the record, the same loop with every cell spelled out as ~10 ATen ops (round 2's figure)
    med2, n2 = _oracle_train_steps(hp, sd, batch, budget_s / 2, 2, 3, sync=torch.cuda.synchronize, reference_ops=False)
    return {"value": frames / med, "unit": "frames/s", "kind": "port",
            "sample": "oracle/seqglow_oracle.py fwd+bwd on cuda:0 through stock PyTorch-ROCm %s (eager, fp32; per-timestep loop, "
                      "window encoders as ONE torch._VF.gru call each and coupling cells as torch._VF.gru_cell - the ops the "
                      "reference's nn.GRU / nn.GRUCell dispatch to), final_model C=%d S=%d T=%d at batch %d, median of %d steps "
                      "after 1 warm-up, %.2f s per step" % (torch.__version__, C, S, T, B, n, med),
            "spelled_out_cells": {"value": frames / med2, "unit": "frames/s",
                                  "sample": "the same loop with every GRU cell as its ~10 separate ATen ops (round 2's "
                                            "denominator), median of %d steps, %.2f s per step" % (n2, med2)}}


def cpu_baseline_sample(hp, C, S, B, nframes, budget_s):
    """Sampling workload: oracle.seqglow_inference (models.py:567-596) on the host cores on a bounded sample — the full batch,
    the first `nframes` generated frames of the sequence (every generated frame costs the same)."""
    from oracle import seqglow_oracle as oracle
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = _oracle_state(hp)
    start = oracle.longest_history(hp["Conditioning"])
    seq_len = start + nframes
    g = torch.Generator().manual_seed(1234)
    data = {"p1_face": torch.zeros(B, start, C)}
    for name, d in (("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(nframes, B, C, generator=g)
    times = []
    with torch.no_grad():
        oracle.seqglow_inference(hp, sd, start + 2, data, noise)   # warm-up: two frames
        while len(times) < 1 or (sum(times) < budget_s and len(times) < 3):
            t0 = time.time()
            oracle.seqglow_inference(hp, sd, seq_len, data, noise)
            times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * nframes / med, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "oracle/seqglow_oracle.py seqglow_inference (torch CPU fp32, per-frame loop as the reference), final_model "
                      "C=%d S=%d, batch %d, first %d generated frames, median of %d runs, %.1f s per run" % (C, S, B, nframes, len(times), med)}


_SMI_HELPER = r"""
import json, shutil, subprocess, sys
exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
for line in sys.stdin:
    dev = line.strip()
    if not dev:
        break
    try:
        r = subprocess.run([exe, "-d", dev, "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        print(json.dumps({"raw": json.loads(r.stdout)}), flush=True)
    except Exception as e:
        print(json.dumps({"error": "%s: %s" % (type(e).__name__, e)}), flush=True)
"""


def start_smi_helper():
    """A child process that runs rocm-smi on request. Started BEFORE this process touches the GPU: on this pool a process that has
    initialised HIP must not fork + exec another program (the box refuses it; under rocprofv3 --pmc it did), and rocm-smi is a
    Python script. The helper never initialises the GPU itself. None when it cannot be started."""
    import subprocess
    if under_profiler():
        return None
    env = {k: v for k, v in os.environ.items() if not (_is_profiler_var(k) and (k != "LD_PRELOAD" or _profiler_preload(v)))}
    try:
        return subprocess.Popen([sys.executable, "-c", _SMI_HELPER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env)
    except Exception:      # noqa: BLE001 - diagnostics only
        return None


def _is_profiler_var(name):
    return name in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCP_TOOL_LIB") or name.startswith(("ROCPROFILER_", "ROCPROF_", "ROCTRACER_", "ROCP_"))


def _profiler_preload(value):
    return any(t in value.lower() for t in ("rocprof", "roctracer", "roctx"))


def under_profiler():
    """True when a tool library is (or may be) preloaded into this process - rocprofv3 sets LD_PRELOAD / ROCP_TOOL_LIBRARIES /
    ROCPROFILER_* and its library can initialise the GPU before bench.py's first line runs (with --pmc it does): the helper would
    then be a fork + exec from a GPU-initialised process, which the pool forbids, so it is not started at all."""
    for k, v in os.environ.items():
        if not v or not _is_profiler_var(k):
            continue
        if k == "LD_PRELOAD" and not _profiler_preload(v):
            continue      # (the GPU pool preloads a library of its own into every process: not a profiler)
        return True
    return False


def stop_smi_helper(helper):
    if helper is None:
        return
    try:
        helper.stdin.write("\n")
        helper.stdin.flush()
        helper.wait(timeout=5)
    except Exception:      # noqa: BLE001
        helper.kill()


def gpu_state_under_load(fn, device, helper, max_s=6.0):
    """Shader clock / socket power of this GPU while `fn(i)` keeps it busy (outside every timed region): the helper runs rocm-smi
    while this process keeps enqueueing steps, so the reading is of the loaded chip, not of the idle one. A slow box (power-capped,
    hot) can then be told from a regression in the record itself. None when there is no helper."""
    import threading
    if helper is None or helper.poll() is not None:
        return None
    box = {}

    def read():
        try:
            helper.stdin.write("%d\n" % (device.index or 0))
            helper.stdin.flush()
            box.update(json.loads(helper.stdout.readline()))
        except Exception as e:      # noqa: BLE001 - diagnostics only
            box["error"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=read)
    t0 = time.time()
    i = 0
    for _ in range(8):      # the queue holds work before the reader starts
        fn(i)
        i += 1
    th.start()
    while th.is_alive() and time.time() - t0 < max_s:
        fn(i)
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    th.join(timeout=25)
    torch.cuda.synchronize()
    if "raw" not in box:
        return {"error": box.get("error", "no output")}
    out = {"steps_issued_meanwhile": i}
    for card, vals in box["raw"].items():
        if not isinstance(vals, dict):
            continue
        for k, v in vals.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor hotspot)")):
                out["%s %s" % (card, k.strip())] = v
    return out


def _dp_summary(prof, eng, world):
    """Mean milliseconds of the HIP-event intervals fused_training_step recorded around its two gradient buckets."""
    if not prof:
        return None
    torch.cuda.synchronize()
    keys = [k for k in prof[0] if k != "mode"]
    res = {"mode": prof[0]["mode"], "steps": len(prof), "world_size": dist.get_world_size() if dist.is_initialized() else world,
           "backend": dist.get_backend() if dist.is_initialized() else None,
           "flow_bucket_bytes": 4 * (eng.n_params - eng.flow_offset), "encoder_bucket_bytes": 4 * eng.flow_offset}
    for k in keys:
        vals = [a.elapsed_time(b) for a, b in (p[k] for p in prof) if a is not None and b is not None]
        res[k + "_ms"] = sum(vals) / len(vals) if vals else None
    if res["mode"] == "overlap" and res.get("flow_bucket_exposed_ms") is not None:
        res["flow_bucket_finished_under_bptt"] = bool(res["flow_bucket_exposed_ms"] < 0.05)
        res["note"] = ("flow bucket all-reduced asynchronously from the moment its gradients are enqueued; bptt_window = main-stream "
                       "time from that launch to the end of the window encoders' BPTT; flow_bucket_exposed = how long the main stream "
                       "still waited for it after the encoder bucket (0 = fully hidden). LFI_DP_SYNC=1 runs both buckets "
                       "synchronously after backward and reports their own durations")
    return res


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


def _host_issue_ms(fn, reps=3):
    """Host cost of issuing ONE step on an EMPTY queue (nothing awaited), median of `reps`, measured OUTSIDE the timed region:
    well under ms_per_step means the step is GPU-bound (inside the timed loop the host runs ahead until the launch queue
    pushes back, so its loop time says nothing)."""
    ts = []
    for i in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return 1e3 * sorted(ts)[len(ts) // 2]


def _roofline_gemm(spec, F, timing, precision):
    """The step's best GEMM (NOT its largest kernel: ~5 % of kernel time) = the cond_transform forward product (F x Ks*D x Ef). SURVEY.md par. 8d counts 2*E*D per flow step and
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
    if precision == "bf16x3" and os.environ.get("LFI_PGEMM", "1") != "0":
        # the product runs on pre-split operand planes (lfi_gemm_planes, lfi_pgemm.hip): 128 x 256 tiles, 512 threads, two
        # workgroups per CU; both operands as row planes, no column-sum epilogue
        # (v_mfma_f32_16x16x32_bf16 on pairs of k-tiles when K has an even number of them - final_model.yaml: 56 - else the
        # 32 x 32 x 16 kernel; LFI_PGEMM_16=0 pins the latter)
        peak, mult = BF16_MFMA_PEAK_TFLOPS, 3.0
        pairs = ((spec.ldf + 15) // 16) % 2 == 0 and os.environ.get("LFI_PGEMM_16", "1") != "0"
        # (name prefix as rocprofv3 prints it; the 16 x 16 x 32 kernel carries one more template argument: its epilogue form)
        kern = "gemm_planes16_kernel<false, false, false, 2, 4" if pairs else "gemm_planes_kernel<false, false, false, 2, 4>"
        tile, threads = 128, 512
        tile_n = 256
    elif precision == "bf16x3":
        big = os.environ.get("LFI_GEMM_256", "1") != "0"
        peak, mult = BF16_MFMA_PEAK_TFLOPS, 3.0
        kern = "gemm_bf16x3_256_kernel<true, true>" if big else "gemm_bf16x3_kernel<true, true>"
        tile, threads = (256, 1024) if big else (128, 256)
        tile_n = tile
    else:
        peak, kern, mult = F32_MFMA_PEAK_TFLOPS, "gemm_f32_kernel<128, 128, 2, 2, true, true", 1.0
        tile, threads = 128, 256
        tile_n = tile
    # HBM bytes per launch of that kernel from the PMC passes of the same command (tools/pmc_bench.sh -> profiles/):
    # bench.py cannot collect counters itself; null when no committed measurement matches this kernel and grid
    traffic, traffic_src = None, None
    try:
        tiles = ((F + tile - 1) // tile) * ((KD + tile_n - 1) // tile_n)
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % precision)))
        for name, v in tj["kernels"].items():
            if kern in name and name.endswith("grid=%d" % (tiles * threads)):
                traffic, traffic_src = v["hbm_bytes"], "profiles/pmc_traffic_%s.json (%s)" % (precision, tj["source"])
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "mfma", "kernel": kern + ("" if kern.endswith(">") else ", ...>") + " cond_transform forward (F x Ks*D x Ef)",
            "role": "best GEMM of the step (about 5 % of its kernel time), not its dominant kernel",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
            "traffic_unit": "bytes per launch (HBM read + write)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": 4.0 * (F * spec.ldf + KD * spec.ldf + F * KD),
            "algorithmic_bytes_note": "A (F x ldf) and B (Ks D x ldf) once as bf16 hi + lo planes (4 B per element, as fp32); the result c "
                                      "(F x Ks D) leaves ONCE as bf16 hi + lo planes (4 B per element; one block format serves gic, the backward mask "
                                      "and, read transposed, dW_c = dgi^T c), and never as fp32",
            "mfma_flops_multiplier": mult, "frac_of_mfma_issue": mult * ach / peak,
            "flops_per_launch": flops, "flops_per_launch_algorithmic": flops_alg,
            "achieved_algorithmic": flops_alg / (ms * 1e-3) / 1e12 if n_launch else None,
            "ms_per_launch": ms, "launches_timed": n_launch}


def _pmc_traffic(precision):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % precision)))
    except (OSError, ValueError):
        return None


def _fwd_flops_per_frame(spec):
    """SURVEY.md par. 8d: as-written forward GEMM FLOPs per frame (2 m n k, elementwise excluded; every window re-encoded per
    timestep, every GRU output twice in the feature vector) -> (encoders, flow steps)."""
    enc = 0.0
    for e in spec.encoders:
        if e.enc in ("rnn", "lstm"):
            enc += e.hist * 2.0 * (e.in_dim + e.hid) * e.ng * e.hid
        elif e.enc == "mlp":
            enc += 2.0 * e.in_dim * e.hist * e.hid
    flow = spec.Ks * (2.0 * spec.E * spec.D + 2.0 * (spec.Ch + spec.D) * spec.G + 2.0 * spec.H * spec.G + 2.0 * spec.H * spec.Cout
                      + 2.0 * spec.C * spec.C)
    return enc, flow


def _roofline_whole_step(spec, F, ms_per_step, precision):
    """The whole training step against both nominal peaks: SURVEY.md 8d's algorithmic FLOPs (3 x forward, as written) over the
    step time and the bf16 dense peak; PMC bytes per step (every kernel, L2 <-> fabric counters of the committed passes of this
    command) over the step time and 8 TB/s."""
    enc, flow = _fwd_flops_per_frame(spec)
    flops = 3.0 * (enc + flow) * F
    peak = BF16_MFMA_PEAK_TFLOPS if precision == "bf16x3" else F32_MFMA_PEAK_TFLOPS
    ach = flops / (ms_per_step * 1e-3) / 1e12
    out = {"flops_per_step_algorithmic": flops, "mflop_per_frame_forward_as_written": (enc + flow) / 1e6,
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
           "hbm_bytes_per_step": None, "hbm_achieved_GBps": None, "hbm_frac": None}
    # the committed PMC passes are of the headline command (16 flow steps, F = 14 336): for any other shape (configs[4], another
    # batch) the bytes do not apply and the record says null instead of dividing another shape's bytes by this shape's time
    tj = _pmc_traffic(precision) if (F == 14336 and spec.Ks == 16) else None
    if tj:
        per_step = tj.get("per_step_hbm_bytes")
        if per_step is None:     # every kernel's bytes / the number of optimiser steps of the profiled run (one adam_clip_kernel each)
            steps = sum(v["dispatches"] for k, v in tj["kernels"].items() if k.startswith("adam_clip_kernel"))
            per_step = sum(v["hbm_bytes"] * v["dispatches"] for v in tj["kernels"].values()) / steps if steps else None
        if per_step:
            gbps = per_step / (ms_per_step * 1e-3) / 1e9
            out.update(hbm_bytes_per_step=per_step, hbm_achieved_GBps=gbps, hbm_frac=gbps / 8000.0,
                       hbm_source="profiles/pmc_traffic_%s.json (committed PMC passes of this command at F = 14 336; applies to that shape)" % precision)
    return out


def _roofline_encoder(spec, B, T, timing, precision, stash_f16=None):
    """`roofline`: the step's LARGEST kernel by time (profiles/*kernel_stats*: ~12 % forward + ~10 % for its BPTT twin) - the
    persistent window-encoder forward recurrence enc_gru_fwd_*_kernel, two launches per step that share the kernel name (p2_face:
    hist 24, p2_speech: hist 16; hid 256). bound = mfma: `achieved` = the recurrence's as-written FLOPs 2 * hid * 3 hid per window
    step (SURVEY.md 8d's `hist * 2 * (in + hid) * 3 hid` less the x-projection, which is hoisted into ONE GEMM over the B * T distinct
    frames and is not this kernel's work) summed over the launches / their summed HIP-event time; ms_per_launch = the mean over the
    launches, which is what rocprofv3 --stats prints for the kernel name. None when the model has no GRU window encoder of that kind."""
    F = B * (T - spec.start)
    legs = []
    for e in spec.encoders:
        n_launch, ms = timing.get("enc_fwd." + e.name, (0, float("nan")))
        if e.enc == "rnn" and e.hid == 256 and n_launch:
            legs.append((e, n_launch, ms))
    if not legs:
        return None
    peak, mult = (BF16_MFMA_PEAK_TFLOPS, 3.0) if precision == "bf16x3" else (F32_MFMA_PEAK_TFLOPS, 1.0)
    flops = sum(2.0 * e.hid * 3 * e.hid * e.hist * F for e, _, _ in legs)
    flops_x = sum(2.0 * (e.in_dim + e.hid) * 3 * e.hid * e.hist * F for e, _, _ in legs)
    ms_sum = sum(ms for _, _, ms in legs)
    ach = flops / (ms_sum * 1e-3) / 1e12
    alg_bytes = 0.0
    for e, _, _ in legs:
        f16 = bool(stash_f16 and stash_f16.get(e.name))
        # per window step and hidden unit: h (4 B, fp32: a GEMM operand of dW_hh) + r, z, n, W_hn h as four fp16 (8 B) or fp32 (16 B);
        # the projected inputs once (B T x 3 hid, shared by the overlapping windows) and the feature block (F x hid)
        alg_bytes += e.hist * F * e.hid * (4.0 + (8.0 if f16 else 16.0)) + 4.0 * (B * T * 3 * e.hid + F * e.hid)
    traffic = None
    tj = _pmc_traffic(precision)
    if tj:
        for name, v in tj["kernels"].items():
            if "enc_gru_fwd_" in name and any(name.endswith("grid=%d" % g) for g in (((F + 31) // 32) * 256, ((F + 63) // 64) * 256,
                                                                                     ((F + 63) // 64) * 512)):
                traffic = max(traffic or 0.0, v["hbm_bytes"])   # the 256-wide launches (the small encoder shares the grid size, not the bytes)
    sec = ms_sum * 1e-3
    return {"bound": "mfma", "kernel": "enc_gru_fwd_*_kernel: persistent GRU window-encoder forward recurrence, %s" % " + ".join(
                "%s (hist %d, hid %d)" % (e.name, e.hist, e.hid) for e, _, _ in legs),
            "role": "largest kernel of the step by time (with its BPTT twin about 22 % of kernel time and ~40 % of the critical path)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "traffic_unit": "bytes per launch (HBM read + write), mean over the launches of this kernel name and grid",
            "mfma_flops_multiplier": mult, "frac_of_mfma_issue": mult * ach / peak,
            "flops_per_step_this_kernel": flops, "launches_per_step": len(legs),
            "achieved_crediting_hoisted_x_projection": flops_x / sec / 1e12, "frac_crediting_hoisted_x_projection": flops_x / sec / 1e12 / peak,
            "ms_per_launch": ms_sum / len(legs), "ms_by_launch": {e.name: round(ms, 4) for e, _, ms in legs},
            "launches_timed": sum(n for _, n, _ in legs),
            "timing_note": "HIP events on the launch stream inside the timed region, around lfi_encode_windows_fwd (the recurrence + a "
                           "few-us weight-fragment kernel in front of it); the small third encoder runs beside them on the second stream",
            "algorithmic_bytes_per_step_this_kernel": alg_bytes, "hbm_achieved_GBps": alg_bytes / sec / 1e9,
            "hbm_frac": alg_bytes / sec / 1e9 / 8000.0}


def _time_steps_at_batch(model, trainer, spec, B, T, device, warmup, steps):
    """Mean milliseconds of one fused training step at another batch size (single rank, engine workspaces regrown)."""
    batches = [synthetic_batch(B, T, spec.C, spec.S, 4321 + i, device) for i in range(2)]
    lr = trainer.lr_at(0)
    for i in range(warmup):
        model.fused_training_step(batches[i & 1], lr, 1, None)
    elapsed, _ = _timed(lambda i: model.fused_training_step(batches[i & 1], lr, 1, None), steps, 1, device)
    return 1e3 * elapsed / steps


def bench_train(args, model, trainer, spec, device, world, rank, hp):
    C, S, T, B = spec.C, spec.S, args.seq_len, args.batch
    N = T - spec.start
    deep = args.workload == "deep"
    batches = [synthetic_batch(B, T, C, S, 1234 + 1000 * rank + i, device) for i in range(2)]
    lr = trainer.lr_at(0)
    allreduce = trainer.allreduce_grads if world > 1 else None

    def step(i):
        return model.fused_training_step(batches[i & 1], lr, world, allreduce)

    for i in range(args.warmup):  # includes the one-off ActNorm data-dependent init
        step(i)
    eng = model.seq_glow.engine
    # HIP events around the `roofline` kernel inside the timed region; the other tagged kernels (`roofline_hbm`'s among them) are
    # timed in a short region of their own afterwards: every event record is a marker packet in the queue (~6 us of dispatch
    # gap), and with all nine tags on the timed step carried 18 of them - 0.1 ms that no training step pays (same-call A/B:
    # 7.86 ms with all tags, 7.77 with the roofline kernels' only; LFI_BENCH_ALL_TAGS=1: the old behaviour)
    roof_tags = None if os.environ.get("LFI_BENCH_ALL_TAGS") == "1" else (
        ("gemm_cond_fwd",) + tuple("enc_fwd." + e.name for e in spec.encoders if e.enc == "rnn" and e.hid == 256))
    eng.enable_timing(True, only=roof_tags)
    elapsed, loss = _timed(step, args.steps, world, device)
    timing = eng.timing_summary()
    if roof_tags is not None:
        eng.enable_timing(True)
        _timed(step, 5, world, device)
        timing_all = eng.timing_summary()
        timing_all.update(timing)   # the roofline kernels' figures stay those of the timed region
        timing = timing_all
    eng.enable_timing(False)
    dp_line = None
    if world > 1:
        # a short region of its own (event records are marker packets): which bucket held the main stream for how long
        model.dp_profile = []
        _timed(step, 5, world, device)
        dp_line = _dp_summary(model.dp_profile, eng, world)
        model.dp_profile = None
        if args.graph_steps > 0 and os.environ.get("LFI_BENCH_DP_GRAPH") == "1":
            # opt-in (never rehearsed on a real multi-GPU transport: the driver's scaling run stays on the eager step): the same
            # data-parallel step as two replayed hipGraphs with the collectives between them (LetsFaceItGlow._capture_dp_step)
            model.step_graph = True
            for i in range(4):
                step(i)
            el_g, _ = _timed(step, args.graph_steps, world, device)
            dp_line["hipgraph_replay"] = {
                "captured": any(isinstance(v, dict) and v.get("dp") for v in getattr(model, "_step_graphs", {}).values()),
                "ms_per_step": 1e3 * el_g / args.graph_steps, "host_issue_ms_per_step": _host_issue_ms(step), "steps": args.graph_steps,
                "note": "two graphs per step split at the flow bucket's launch point, both all-reduces and clip + Adam eager"}
            model.step_graph = False
    host_issue = _host_issue_ms(step, reps=5)   # (without any per-kernel events)
    gpu_state = None
    if rank == 0 and not args.quick and world == 1 and getattr(args, "smi_helper", None) is not None:
        # the load is the real step at lr = 0 (parameters do not move) with Adam's moments, step count and dropout counter put
        # back afterwards: the legs below run on the state the timed region left
        keep = eng.optimizer_state()
        gpu_state = gpu_state_under_load(lambda i: model.fused_training_step(batches[i & 1], 0.0, world, allreduce), device,
                                         args.smi_helper)
        eng.load_optimizer_state(keep)
        del keep
    graph_line = None
    if world == 1 and args.graph_steps > 0:
        # the same step as ONE replayed hipGraph (opt-in: LetsFaceItGlow.step_graph; bit-identical parameters): two more eager calls
        # of the shape, the capture, then K timed replays. Reported beside the headline, which stays with eager launches.
        model.step_graph = True
        for i in range(4):
            step(i)
        el_g, _ = _timed(step, args.graph_steps, 1, device)
        graphed = any(isinstance(v, dict) for v in getattr(model, "_step_graphs", {}).values())
        graph_line = {"captured": graphed, "ms_per_step": 1e3 * el_g / args.graph_steps, "host_issue_ms_per_step": _host_issue_ms(step),
                      "steps": args.graph_steps, "note": "fused_training_step replayed as one hipGraph per step (forward, backward, "
                      "clip, Adam on both streams; dropout key and Adam step size read from device memory)"}
        model.step_graph = False
    if rank != 0:
        return None
    frames = world * B * N * args.steps
    F = B * N
    if deep:
        metric = "FLAME frames/s, full training step (fwd+bwd+clip+Adam), deep flow K=32 x L=3 (96 flow steps), batch %d per GPU" % B
        workload = ("deep-flow training step: final_model.yaml widths with Glow K=32, L=3 (96 flow steps), synthetic %d-d FLAME + "
                    "%d-d speech, T=%d (%d timesteps), batch %d per GPU (BASELINE.json configs[4]%s)"
                    % (C, S, T, N, B, "" if world == 1 else ", data-parallel"))
    else:
        metric = "FLAME frames/s, full training step (fwd+bwd+clip+Adam), final_model.yaml batch 256 per GPU"
        workload = ("final_model.yaml training step, synthetic %d-d FLAME + %d-d speech, T=%d, batch %d per GPU (%s%s)"
                    % (C, S, T, B, "BASELINE.json configs[1]" if (C, S) == (50, 27) else
                       "the corpus' native dims, hparams/final_model.yaml as shipped", "" if world == 1 else ", data-parallel"))
    ms_step = 1e3 * elapsed / args.steps
    roof_gemm = _roofline_gemm(spec, F, timing, args.precision)
    roof = None if deep else _roofline_encoder(spec, B, T, timing, args.precision, getattr(eng, "_enc_stash_f16", None))
    # the verbose records first, the short ones that summarise the run LAST: the driver stores the tail of this line
    out = {
        "metric": metric,
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "host_issue_ms_per_step": host_issue,
        "step_launch": "eager (~100 launches per step on two streams)",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _dtype_label(args.precision, eng, F),
        "data": "synthetic",
        "config": {"workload": workload,
                   "K": spec.Ks, "H": spec.H, "cond_dim": spec.D, "feature_dim": spec.E, "frames_per_step_per_gpu": F,
                   "parallelism": "dp%d" % world, "params": eng.n_params, "gemm_precision": args.precision,
                   "gemm_products": _products_label(args.precision, eng, F)},
        "final_loss": float(loss),
        "roofline": roof if roof is not None else roof_gemm,
        "roofline_best_gemm": roof_gemm if roof is not None else None,
        "kernel_timing": {t: {"launches": n, "ms": round(m_, 4)} for t, (n, m_) in timing.items()},
        "gpu_state_under_load": gpu_state,
    }
    tail = {"roofline_whole_step": _roofline_whole_step(spec, F, ms_step, args.precision), "hipgraph_replay": graph_line}
    if dp_line is not None:
        out["data_parallel"] = dp_line
    if world == 1 and args.precision == "bf16x3" and eng.backward_product_count(F) == 2 and args.three_products_steps > 0:
        # the same step with three products in the backward GEMMs too (round 2's arithmetic), timed in the same process
        keep = eng.backward_products
        eng.backward_products = 3
        for i in range(3):
            step(i)
        n3 = args.three_products_steps
        el3, _ = _timed(step, n3, 1, device)
        eng.backward_products = keep
        tail["three_products_everywhere"] = {"ms_per_step": 1e3 * el3 / n3, "value": world * B * N * n3 / el3, "unit": "frames/s",
                                            "note": "engine_backward_products=3; same process, same batches, %d steps" % n3}
    if world == 1 and not deep and B == 256 and args.strong_anchor_batch > 0:
        # north_star asks for STRONG scaling at 8 GPUs (configs[2]: global batch 2048 = 8 x 256). The driver's N-GPU runs keep
        # 256 per GPU ("scaling": "weak"); this is the missing denominator: ONE GPU stepping the global batch of the 8-GPU run,
        # so that strong-scaling speed-up at N = 8 is strong_scaling_anchor.ms_per_step / that run's ms_per_step
        gb = args.strong_anchor_batch
        ms = _time_steps_at_batch(model, trainer, spec, gb, T, device, 2, 5)
        tail["strong_scaling_anchor"] = {"batch": gb, "ms_per_step": ms, "frames_per_s": gb * N / (ms * 1e-3), "n_gpus": 1,
                                        "note": "one GPU at the global batch of BASELINE.json configs[2] (8 x 256), mean of 5 steps "
                                                "after 2 warm-ups; strong speed-up at N GPUs = this / the N-GPU run's ms_per_step "
                                                "at batch %d per GPU" % (gb // 8)}
        import gc
        eng._ws.clear()     # the 2048-batch workspaces (~80 GB) go back to the allocator before the torch baseline runs
        eng._last = None
        gc.collect()
        torch.cuda.empty_cache()
    if world == 1 and args.torch_gpu_baseline_seconds > 0 and not deep:
        # the >= 10x denominator of BASELINE.json's north_star. BASELINE.md holds no PUBLISHED number for this metric, so
        # `vs_baseline` stays null (bench contract); the measured ratio is reported under its own name
        tg = torch_gpu_baseline(hp, C, S, T, B, device, args.torch_gpu_baseline_seconds)
        out["torch_gpu_baseline"] = tg
        tail["vs_torch_gpu_baseline"] = out["value"] / tg["value"]
        tail["vs_torch_gpu_baseline_spelled_out_cells"] = out["value"] / tg["spelled_out_cells"]["value"]
    if world == 1 and args.cpu_baseline_seconds > 0:
        if deep:
            # bounded sample of the same workload: the full batch, the first 8 of the 488 timesteps (every timestep costs the
            # same: 96 flow steps + three window encoders), ~5 s per forward + backward on the host
            Ts = spec.start + 8
            out["cpu_baseline"] = cpu_baseline(hp, C, S, Ts, B, min(args.cpu_baseline_seconds, 40.0),
                                               note=" (bounded sample: the full batch, the first 8 of %d timesteps)" % N)
        else:
            out["cpu_baseline"] = cpu_baseline(hp, C, S, T, B, args.cpu_baseline_seconds)
    out.update(tail)
    return out


def _products_label(precision, eng, frames):
    if precision != "bf16x3":
        return "exact fp32 (f32-input MFMA)"
    bwd = eng.backward_product_count(frames)
    txt = "forward classes 3 (a_hi b_hi + a_hi b_lo + a_lo b_hi); backward classes %d%s" % (
        bwd, " (A operand - a gradient - rounded to bf16: a_lo b_hi not issued; engine_backward_products=%s)"
        % eng.backward_products if bwd == 2 else "")
    if eng.pass_skip:
        txt += "; overrides: " + ", ".join("%s: %d" % (c, 3 - bin(b & 3).count("1")) for c, b in sorted(eng.pass_skip.items()))
    return txt


def _dtype_label(precision, eng, frames):
    if precision == "f32":
        return "f32"
    bwd = eng.backward_product_count(frames)
    return "f32 (GEMM operands split into bf16 hi+lo, f32 accumulate; %s)" % (
        "three products in every GEMM" if bwd == 3 and not eng.pass_skip else
        "three products in the forward GEMMs, two in the backward GEMMs" if not eng.pass_skip else _products_label(precision, eng, frames))


def _sample_frame_label(eng):
    fp = eng.sample_frame_precision
    if fp is None:
        fp = 9 if eng.precision == 1 else 0
    return {0: "per-frame GEMMs and reverse cells on the exact f32-input MFMA",
            1: "per-frame GEMMs as three bf16 products (2^-16 relative); the reverse cells' recurrent products as three fp16 "
               "products (2^-22), their LinearZeros / W^-1 products on the f32-input MFMA",
            5: "fp32-grade throughout: per-frame GEMMs as six bf16 products of three-piece operands (dropped terms 2^-24 relative), "
               "the reverse cells' recurrent products as three fp16 products of two-piece operands (2^-22), their LinearZeros / "
               "W^-1 products on the f32-input MFMA",
            9: "fp32-grade throughout: per-frame GEMMs and the reverse cells' recurrent products as three fp16 products of two-piece "
               "operands (11 + 11 mantissa bits, 2^-22 relative; activations and weights sit inside fp16's range), the cells' "
               "LinearZeros / W^-1 products on the f32-input MFMA"}.get(int(fp), str(fp))


def measure_sampling_error(hp, C, S, device, precision):
    """err / fp32 floor of the sampler at the model's full depth, MEASURED in this run (until round 5 the record
    quoted the newest committed parity report): SeqGlow.inference of a perturbed random-init model (LinearZeros / ActNorm moved off
    their zero init, as tests/test_gpu_headline_parity.py::test_k16_sampling_against_oracle does: at init the coupling ignores its
    conditioning), batch 8, 56 generated frames, injected prior noise - the engine on the GPU against the fp64 oracle, and plain fp32
    torch on the CPU against the same oracle (the floor any fp32 implementation of models.py:567-596 sits on). Part of the
    cpu_baseline leg: a few seconds of host time, the only place bench.py may run the oracle."""
    import copy
    from argparse import Namespace
    import numpy as np
    from lets_face_it_amd.glow.models import SeqGlow
    from oracle import seqglow_oracle as oracle
    rng = torch.get_rng_state()
    np_state = np.random.get_state()
    torch.manual_seed(1234)
    np.random.seed(1234)
    hp = copy.deepcopy(hp)
    hp["engine_precision"] = precision
    m = SeqGlow(Namespace(**hp))
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "final_linear" in name:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            elif "actnorm" in name:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(device)
    m.glow.set_actnorm_init(True)
    m.eval()
    start = oracle.longest_history(hp["Conditioning"])
    B, nframes = 8, 56
    seq_len = start + nframes
    g = torch.Generator().manual_seed(3)
    data = {"p1_face": torch.randn(B, start, C, generator=g)}
    for name, d in (("p2_face", C), ("p1_speech", S), ("p2_speech", S)):
        data[name] = torch.randn(B, seq_len, d, generator=g)
    noise = torch.randn(nframes, B, C, generator=g) * 0.8
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))
    t0 = time.time()
    with torch.no_grad():
        ref = oracle.seqglow_inference(hp, {k: v.double() for k, v in sd.items()}, seq_len, {k: v.double() for k, v in data.items()},
                                       noise.double())
        ref32 = oracle.seqglow_inference(hp, sd, seq_len, data, noise)
        host_s = time.time() - t0
        out = m.inference(seq_len, {k: v.to(device).contiguous() for k, v in data.items()}, noise=noise.to(device))
    torch.set_num_threads(threads)
    torch.set_rng_state(rng)
    np.random.set_state(np_state)
    err = float((out.cpu().double() - ref).abs().max())
    floor = float((ref32.double() - ref).abs().max())
    # the maximum over 22 400 values of an error that compounds through 56 autoregressive frames moves by +-15 % between any two fp32
    # roundings of the same computation (profiles/round6_exact_gates.md); the RMS over all values is the stable statistic
    rms = float((out.cpu().double() - ref).pow(2).mean().sqrt())
    floor_rms = float((ref32.double() - ref).pow(2).mean().sqrt())
    switches = sorted(k for k in os.environ if k.startswith("LFI_") and k not in ("LFI_DIST_BACKEND", "LFI_PARITY_REPORT"))
    del m
    return {"max_abs_err_vs_fp64_oracle": err, "fp32_floor": floor, "err_over_fp32_floor": err / max(floor, 1e-30),
            "rms_err_vs_fp64_oracle": rms, "fp32_floor_rms": floor_rms, "rms_over_fp32_floor_rms": rms / max(floor_rms, 1e-30),
            "north_star_abs_tolerance": 1e-5, "measured": "in this run",
            "sample": "%d flow steps, batch %d x %d generated frames, perturbed random-init weights, injected noise; fp64 oracle + fp32 "
                      "CPU pass %.1f s of host time" % (oracle.n_flow_steps(hp), B, nframes, host_s),
            "kernel_switches_in_environment": switches}


def bench_sample(args, model, spec, device, world, rank, hp):
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

    for i in range(max(args.warmup, 2)):   # the second call of a shape captures the per-frame sequence as a hipGraph
        step(i)
    eng = model.seq_glow.engine
    eng.enable_timing(True)
    elapsed, out_faces = _timed(step, args.steps, world, device)
    host_issue = _host_issue_ms(step)
    timing = eng.timing_summary()
    eng.enable_timing(False)
    if rank != 0:
        return None
    frames = world * B * nframes * args.steps
    # roofline of the autoregressive part: per generated frame the two conditioning products (the 5-frame prev_p1_face
    # window through its hist1*C columns of cond_transform, then c W_ih[:, Ch:]^T) and the Ks reverse flow steps
    # (recurrent cell, LinearZeros, W^-1). As-written GEMM FLOPs (2 m n k, SURVEY.md par. 8d's convention), all of them on
    # the critical path of 276 dependent frames; time = HIP events around the replay of the captured per-frame sequence
    e1 = spec.encoders[0]
    per_frame = spec.Ks * (2.0 * e1.fdim * spec.D + 2.0 * spec.D * spec.G + 2.0 * spec.Ch * spec.G + 2.0 * spec.H * spec.G
                           + 2.0 * spec.H * spec.Cout + 2.0 * spec.C * spec.C)
    n_launch, ms = timing.get("sample_graph", (0, float("nan")))
    flops = per_frame * B * nframes
    ach = flops / (ms * 1e-3) / 1e12 if n_launch else float("nan")
    peak, mult = (BF16_MFMA_PEAK_TFLOPS, 3.0) if args.precision == "bf16x3" else (F32_MFMA_PEAK_TFLOPS, 1.0)
    res = {
        "metric": "FLAME frames/s, autoregressive sampling (SeqGlow.inference), final_model.yaml",
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "host_issue_ms_per_step": host_issue,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _dtype_label(args.precision, eng, 0),
        "data": "synthetic",
        "config": {"workload": "autoregressive sampling, batch %d, seq_len %d (%d generated frames per sequence), "
                               "BASELINE.json configs[3]" % (B, T, nframes), "K": spec.Ks, "H": spec.H,
                   "parallelism": "replicas%d" % world, "gemm_precision": args.precision,
                   "autoregressive_part": _sample_frame_label(eng)},
        "ms_per_generated_frame": 1e3 * elapsed / args.steps / nframes,
        "finite": bool(torch.isfinite(out_faces).all()),
        "roofline": {"bound": "mfma", "kernel": "hipGraph of the per-frame sequence x %d frames: 2 conditioning GEMMs + state "
                                                "reset + flow_rev_chain_kernel (%d dependent reverse flow steps)" % (nframes, spec.Ks),
                     "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                     "flops_per_launch": flops, "flops_per_generated_frame_per_sample": per_frame,
                     "mfma_flops_multiplier": mult, "ms_per_launch": ms, "launches_timed": n_launch,
                     "note": "a chain of %d x %d dependent small cells: latency-bound, not throughput-bound; the static part "
                             "(window encoders + the non-autoregressive cond_transform columns) is timed separately"
                             % (nframes, spec.Ks)},
        "kernel_timing": {t: {"launches": n, "ms": round(m_, 4)} for t, (n, m_) in timing.items()},
    }
    if world == 1 and args.cpu_baseline_seconds > 0:
        res["cpu_baseline"] = cpu_baseline_sample(hp, C, S, B, 24, min(args.cpu_baseline_seconds, 30.0))
    if world == 1 and (args.cpu_baseline_seconds > 0 or getattr(args, "measure_sampling_error", False)):
        try:
            res["error_vs_fp32_floor"] = measure_sampling_error(hp, C, S, device, args.precision)
        except Exception as e:      # noqa: BLE001 - the record says so instead of quoting somebody else's number
            res["error_vs_fp32_floor"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return res


def self_launch(n):
    """`bench.py --gpus N` (N > 1) started as a plain process: run the N ranks as CHILD processes under torch.distributed.run (one
    rank per GPU, rendezvous on 127.0.0.1, a free port) and hand back their exit code. The parent imports torch but never
    initialises HIP (no device query, no tensor on a GPU), and it starts a child - it does not exec - so the pool's rule against
    replacing a GPU-initialised process is not in play anywhere. stdout of the ranks: the ONE JSON line of rank 0 goes to this
    process' stdout, anything else a rank or the launcher prints there is passed on to stderr."""
    import socket
    import subprocess
    if under_profiler():
        #  the profiler's preloaded library has initialised the GPU in THIS process before its first line ran (with --pmc it
        # does): starting the launcher from here is the fork + exec the pool refuses - and the ranks would be grandchildren the
        # profiler does not follow anyway
        print("bench.py: --gpus %d under a profiler: this process would have to start the ranks from a GPU-initialised parent, which "
              "this pool refuses, and nothing of the ranks would be profiled. Profile ONE rank directly (program after `--`, RANK / "
              "LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set by hand), or run --gpus 1." % n, file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:
        txt = line.strip()
        is_json = False
        if txt.startswith("{") and txt.endswith("}"):
            try:
                is_json = "metric" in json.loads(txt)
            except ValueError:
                pass
        if is_json:
            lines += 1
            print(txt, flush=True)
        else:
            print(line, end="", file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print("bench.py: the ranks exited 0 but printed %d JSON lines (expected 1)" % lines, file=sys.stderr)
        return 1
    return rc


def launch_check(world, rank):
    """--launch-check: the launch shape only (self_launch / torch.distributed.run, rendezvous, rank 0's one JSON line), on gloo
    with CPU tensors - no GPU, no engine. The -m "not gpu" suite runs this; the real N > 1 bench needs N GPUs."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        seen = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1.0
    if rank == 0:
        print(json.dumps({"metric": "launch check (no measurement)", "value": None, "n_gpus": world, "launch_check": True,
                          "rank_sum": seen, "expected_rank_sum": world * (world + 1) / 2.0}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--seq-len", type=int, default=80)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=70.0,
                    help="budget of the CPU baseline leg (oracle on the host cores at the metric's batch: 1 warm-up + 2-3 "
                         "steps of ~20 s); 0 disables it")
    ap.add_argument("--torch-gpu-baseline-seconds", type=float, default=30.0,
                    help="budget of the stock PyTorch-ROCm leg (the oracle's op sequence on cuda:0, 1 warm-up + 3-5 steps); "
                         "0 disables it")
    ap.add_argument("--precision", choices=("f32", "bf16x3"), default=os.environ.get("LFI_PRECISION", "bf16x3"),
                    help="GEMM arithmetic: exact f32 MFMA, or bf16 hi/lo split operands (3 bf16 MFMAs per product)")
    ap.add_argument("--workload", choices=("train", "sample", "deep"), default="train",
                    help="train: BASELINE.json configs[1] (the headline metric); sample: configs[3], autoregressive "
                         "inference at --batch 1024 --seq-len 300 unless given; deep: configs[4], the training step of a "
                         "K=32 x L=3 flow at --batch 128 --seq-len 512 unless given")
    ap.add_argument("--three-products-steps", type=int, default=20,
                    help="N = 1: also time this many steps with three bf16 products in the backward GEMMs too (reported as "
                         "three_products_everywhere); 0 disables it")
    ap.add_argument("--graph-steps", type=int, default=20,
                    help="N = 1: also time this many steps replayed as a captured hipGraph (reported as hipgraph_replay); 0 disables it")
    ap.add_argument("--strong-anchor-batch", type=int, default=2048,
                    help="N = 1, workload train: also time one GPU at this batch (the global batch of configs[2]) and report it "
                         "as strong_scaling_anchor; 0 disables it")
    ap.add_argument("--hparams", default=os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
    ap.add_argument("--no-gpu-state", action="store_true",
                    help="do not start the rocm-smi helper process (runs under rocprofv3 --pmc, where the profiler has initialised the "
                         "GPU before this program starts and the box refuses every fork + exec)")
    ap.add_argument("--quick", action="store_true",
                    help="the timed region and its roofline only: no baselines, no hipGraph / three-product / anchor legs, no "
                         "further workloads (A/B runs)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous of the ranks on gloo + rank 0's JSON line only (no GPU): checks the way --gpus N starts itself")
    ap.add_argument("--no-more-workloads", action="store_true",
                    help="N = 1, workload train: do not append the `sampling` (configs[3]), `deep_flow` (configs[4]) and "
                         "`native_dims` (C=56 / S=30) sub-records")
    args = ap.parse_args()
    if args.quick:
        args.cpu_baseline_seconds = args.torch_gpu_baseline_seconds = 0.0
        args.three_products_steps = args.graph_steps = args.strong_anchor_batch = 0
        args.no_more_workloads = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus == 1:
        # A process that has trained before it samples (this one: the `sampling` sub-record runs behind the training legs) binds more
        # HIP streams to hardware queues than a process that only samples, and the sampler's ~550 dependent launches per run of frames
        # then cost 15 us more per generated frame: 50.8 ms per 1024 x 300 call against 45.3 in a fresh process on the same box -
        # reproduced by ONE earlier fork / join of a second stream with the default stream, and gone with three hardware queues per
        # process instead of HIP's default four (two runs each, one box: 45.2 / 46.1 against 51.5 / 50.8 ms; the training step, its
        # graph replay, the batch-2048 anchor and the deep flow within run-to-run noise either way:
        # profiles/round6_sampler_stream_pick.md, tools/sample_after_train_probe.py). Read by the HIP runtime when it initialises,
        # i.e. after this line; only for the single-process run (data-parallel ranks keep the default: RCCL's streams want queues).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
    if world != args.gpus:   # checked before anything touches the GPU
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` typed as for N = 1: this process - which has not touched the GPU and never will - starts
            # the N ranks itself and relays rank 0's JSON line and the launcher's exit code
            raise SystemExit(self_launch(args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...` (or without "
                         "WORLD_SIZE in the environment: bench.py then starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if args.launch_check:
        return launch_check(world, rank)
    # LFI_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share cards, tensors travel
    # through the host): same code path, barriers and timing; the default is RCCL with one GPU per rank
    backend = os.environ.get("LFI_DIST_BACKEND", "nccl")
    # (before anything initialises the GPU: see start_smi_helper)
    args.smi_helper = start_smi_helper() if (world == 1 and rank == 0 and not args.quick and not args.no_gpu_state) else None
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

    def build(a, hparams_file=None):
        """Random-init model of workload a.workload at a's batch / length (seeded as train.py does: seed_everything(1234))."""
        hp = load_hparams_file(hparams_file or a.hparams)
        if a.workload == "deep":
            hp["Glow"]["K"], hp["Glow"]["L"] = 32, 3
        hp["batch_size"] = a.batch
        hp["engine_precision"] = a.precision
        hp["Train"]["seq_len"] = a.seq_len
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
        return hp, model, trainer

    def release(model):
        """Give a leg's workspaces (up to ~80 GB) back to the allocator before the next leg builds its own."""
        import gc
        eng = model.seq_glow.engine
        if eng is not None:
            eng._ws.clear()
            eng._last = None
            eng._sample_graphs = {}
        model.__dict__.pop("_step_graphs", None)
        gc.collect()
        torch.cuda.empty_cache()

    def run(a, hparams_file=None):
        hp, model, trainer = build(a, hparams_file)
        spec = model.seq_glow.spec
        try:
            if a.workload == "sample":
                return bench_sample(a, model, spec, device, world, rank, hp)
            return bench_train(a, model, trainer, spec, device, world, rank, hp)
        finally:
            release(model)
            del model, trainer

    if args.workload == "deep":
        if args.batch == 256:
            args.batch = 128
        if args.seq_len == 80:
            args.seq_len = 512
        if args.steps == 20:
            args.steps = 5
    out = run(args)
    if rank == 0 and world == 1 and args.workload == "train" and not args.no_more_workloads:
        # The driver runs `bench.py --gpus 1` only: the other single-GPU workloads of BASELINE.json (configs[3] sampling, configs[4]
        # deep flow) and the training step at the corpus' native dims (final_model.yaml as shipped: C = 56, S = 30) ride on the same
        # line as sub-records, each timed by the same _timed() bracket in this process after the headline's workspaces were freed.
        import copy

        def sub(workload, **kw):
            a = copy.copy(args)
            a.workload = workload
            a.cpu_baseline_seconds = a.torch_gpu_baseline_seconds = 0.0
            a.three_products_steps = a.graph_steps = a.strong_anchor_batch = 0
            a.quick = True
            a.measure_sampling_error = workload == "sample" and args.cpu_baseline_seconds > 0   # (part of the cpu_baseline leg)
            hparams_file = kw.pop("hparams_file", None)
            for k, v in kw.items():
                setattr(a, k, v)
            try:
                r = run(a, hparams_file)
            except Exception as e:      # noqa: BLE001 - a sub-record must not take the headline down with it
                return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            roof = r.get("roofline") or {}
            whole = r.get("roofline_whole_step") or {}
            return {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                    "steps": r["steps"], "warmup": r["warmup"], "workload": r["config"]["workload"], "dtype": r["dtype"],
                    "arithmetic": r["config"].get("autoregressive_part") or r["config"].get("gemm_products"),
                    "roofline_kernel": roof.get("kernel"), "roofline_frac": roof.get("frac"),
                    "roofline_frac_of_mfma_issue": roof.get("frac_of_mfma_issue") or
                    (roof.get("frac") * roof.get("mfma_flops_multiplier", 1.0) if roof.get("frac") is not None else None),
                    "whole_step_frac_of_mfma_peak": whole.get("frac"),
                    "sampling_error_vs_fp32_floor": r.get("error_vs_fp32_floor"),
                    "kernel_timing": r.get("kernel_timing"), "final_loss": r.get("final_loss"), "finite": r.get("finite")}

        out["sampling"] = sub("sample", batch=1024, seq_len=300, steps=5, warmup=2)
        out["deep_flow"] = sub("deep", batch=128, seq_len=512, steps=3, warmup=3)
        out["native_dims"] = sub("train", batch=256, seq_len=80, steps=10, warmup=3,
                                 hparams_file=os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model.yaml"))
    stop_smi_helper(getattr(args, "smi_helper", None))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
