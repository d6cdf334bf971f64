"""GPU, two ranks on ONE card over gloo: the data-parallel path of the real engine (fused_training_step with the two-bucket
asynchronous gradient all-reduce, ActNorm-init statistics all-reduced, parameters broadcast from rank 0), launched exactly as
the driver launches bench.py: python -m torch.distributed.run --nproc-per-node 2 (tools/dp_gloo_check.py). NCCL/RCCL needs
one device per rank, which the test box does not have; gloo moves the same tensors through the host, so everything but the
transport is what runs on the 8-GPU node. Checked: after ActNorm init + 4 optimiser steps both ranks hold identical
parameters, and they equal those of ONE process stepping on the concatenated batch.

This file sorts first on purpose and must not touch the GPU itself: the GPU boxes refuse to start another program from a
process that has initialised the card."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(extra, limit=240, nproc=2):
    """Start tools/dp_gloo_check.py under torch.distributed.run. Nothing here may initialise HIP in the pytest process (the
    boxes refuse to start another program from a process that has): the GPU is detected through /dev/kfd, not through
    torch.cuda.device_count() (which falls through to hipGetDeviceCount on builds without amdsmi)."""
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run tests/test_a_gpu_dp.py on its own or first")
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = ["timeout", "-k", "10", str(limit), sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "dp_gloo_check.py")] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=limit + 60)


def test_two_rank_data_parallel_matches_single_process():
    r = _launch([])
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    m = re.search(r"ranks identical: (\w+); vs one process on the concatenated batch: max rel diff ([0-9.e+-]+)", r.stdout)
    assert m, tail
    print(m.group(0))
    assert m.group(1) == "True" and float(m.group(2)) < 2e-5


def test_four_rank_data_parallel_matches_single_process():
    """World size 4 (four gloo ranks on one card: the bucket offsets, the 1 / world gradient multiplier and the ActNorm-statistics
    all-reduce at a world size that is not 2) against ONE process on the concatenated batch of the four shards."""
    r = _launch([], nproc=4)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    m = re.search(r"ranks identical: (\w+); vs one process on the concatenated batch: max rel diff ([0-9.e+-]+)", r.stdout)
    assert m, tail
    print(m.group(0))
    assert m.group(1) == "True" and float(m.group(2)) < 2e-5


def test_two_rank_data_parallel_at_final_widths():
    """The same check at final_model.yaml widths (17.3 M parameters: a 69 MB flat gradient in two buckets, 16 flow steps of
    ActNorm statistics), batch 256 per rank, T = 80: the bucket offsets, the asynchronous handle and the statistics
    all-reduce at the sizes the 8-GPU node will see (transport aside: gloo here, RCCL there)."""
    r = _launch(["--final"], limit=420)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    m = re.search(r"ranks identical: (\w+); vs one process on the concatenated batch: max rel diff ([0-9.e+-]+)", r.stdout)
    assert m, tail
    print(m.group(0))
    # three Adam steps at lr 1e-3: the two-rank gradient is (g_a + g_b) / 2 with each half summed on its own, the single
    # process sums all 512 samples in one order; Adam's g / (sqrt(v) + eps) turns that rounding difference into a parameter
    # difference of a few 1e-5 of max |p| at 17.3 M parameters (6.5e-5 measured); the ranks themselves must agree bit for bit
    assert m.group(1) == "True" and float(m.group(2)) < 2e-4


@pytest.mark.parametrize("final", [False, True])
def test_two_rank_data_parallel_step_as_two_graphs(final):
    """Round 6: under data parallelism the step replays TWO hipGraphs split at the flow bucket's launch point
    (masks + forward + flow backward | window encoders' backward) with the collectives between / behind them and clip + Adam as three
    launches, instead of ~100 eager launches. Two ranks over gloo, the engine's own dropout masks, 8 optimiser steps: parameters and
    Adam moments bit-identical to the eager data-parallel steps on every rank, the ranks identical, and at least three steps really
    ran on the captured graphs. final=True: final_model.yaml widths at batch 256 per rank (the bucket sizes of the 8-GPU node)."""
    r = _launch(["--graph"] + (["--final"] if final else []), limit=420)
    tail = "\n".join([ln for ln in r.stdout.splitlines() if "differ" in ln or "Error" in ln or "graph data-parallel" in ln] +
                     r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    m = re.search(r"bit-identical to the eager data-parallel step on every rank: (\w+); steps on captured graphs \(min over ranks\): (\d+); "
                  r"ranks identical: (\w+)", r.stdout)
    assert m, tail
    print(m.group(0))
    assert m.group(1) == "True" and int(m.group(2)) >= 3 and m.group(3) == "True"


def test_rccl_launch_path_with_one_rank():
    """RCCL itself, as far as one card allows: ONE rank on torch.distributed backend "nccl" runs fused_training_step with the
    trainer's real all-reduce hooks at final widths (the asynchronous flow bucket travels on RCCL's stream under the encoders'
    BPTT kernels and the side-stream work, the encoder bucket on the main stream, ActNorm statistics, the parameter
    broadcast) and must leave bit-identical parameters to the same steps without collectives (the stand-in doubles the
    gradient as a second identical rank would; the optimiser divides by world = 2). No transport is exercised - that needs the
    8-GPU node - but library load, communicator init, stream ordering and the work handles are."""
    r = _launch(["--nccl1"], limit=300, nproc=1)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert "(nccl)" in r.stdout and "parameters identical to the collective-free steps: True" in r.stdout, tail


def test_bench_gpus_2_starts_itself_and_reports_data_parallel():
    """`python bench.py --gpus 2 --quick ...` typed like the N = 1 line (no launcher, no WORLD_SIZE): bench.py starts its two
    ranks itself (bench.self_launch; here gloo, both ranks on the box's one card) and its stdout is ONE JSON line with n_gpus 2
    and the `data_parallel` record of the two gradient buckets. The record is also kept as
    gpurun_out/bench_n2_gloo_selflaunch.json (the rehearsal the driver's 8-GPU run will repeat on RCCL)."""
    import json
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run tests/test_a_gpu_dp.py on its own or first")
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LFI_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = ["timeout", "-k", "10", "420", sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--quick", "--steps", "3",
           "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=480)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["parallelism"] == "dp2" and rec["config"]["frames_per_step_per_gpu"] == 256 * 56
    dp = rec["data_parallel"]
    assert dp["world_size"] == 2 and dp["backend"] == "gloo" and dp["mode"] == "overlap"
    assert dp["flow_bucket_bytes"] + dp["encoder_bucket_bytes"] == 4 * rec["config"]["params"]
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_n2_gloo_selflaunch.json"), "w") as f:
        f.write(lines[0] + "\n")
