import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_gpu_memory():
    """Models and their engines form reference cycles (module <-> bound methods), so a finished test's workspaces (tens of GB
    at BASELINE sizes) stay allocated until the cyclic collector happens to run: collect after every test."""
    yield
    import gc
    gc.collect()
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_initialized():     # a Python-side flag: never touches HIP itself (tests/test_a_gpu_dp.py relies on that)
            torch.cuda.empty_cache()


@pytest.fixture(scope="session", autouse=True)
def _modest_cpu_thread_team():
    """The CPU oracle is a Python loop of small ops (batch 4 .. 64, a few hundred floats per op): with every core of a GPU box in
    the team each op costs more in hand-off than in arithmetic (measured on the box: a 96-flow-step oracle pass 54 .. 82 s with
    all cores, 5 s with 8). Tests that walk big batches (the headline configuration) set their own count."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(min(8, n))
    yield
    torch.set_num_threads(n)
