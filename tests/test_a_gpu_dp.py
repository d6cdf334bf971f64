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
