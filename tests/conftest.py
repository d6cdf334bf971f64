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
