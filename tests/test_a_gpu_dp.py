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


def test_two_rank_data_parallel_matches_single_process():
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run tests/test_a_gpu_dp.py on its own or first")
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = ["timeout", "-k", "10", "240", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tools", "dp_gloo_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    m = re.search(r"ranks identical: (\w+); vs one process on the concatenated batch: max rel diff ([0-9.e+-]+)", r.stdout)
    assert m, tail
    print(m.group(0))
    assert m.group(1) == "True" and float(m.group(2)) < 2e-5
