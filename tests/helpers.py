"""Shared test helpers: golden-fixture loading and error metrics."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ("tiny", "tiny_lstm", "tiny_additive", "odd", "mid", "mlp", "p1enc", "p1mlp", "framenb", "lstmenc", "p1lstm", "dense")


class Fixture:
    """One ``tests/golden/<name>.npz`` written by ``tests/golden/make_golden.py`` from the reference."""

    def __init__(self, name):
        self.name = name
        self.raw = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.hp = json.loads(str(self.raw["hparams_json"]))
        self.B, self.T, self.C, self.S, self.N, self.start = [int(v) for v in self.raw["dims"]]

    def group(self, prefix, dtype=None):
        out = {}
        for k in self.raw.files:
            if k.startswith(prefix):
                t = torch.from_numpy(self.raw[k])
                if dtype is not None and t.dtype.is_floating_point:
                    t = t.to(dtype)
                out[k[len(prefix):]] = t
        return out

    def get(self, key, dtype=None):
        t = torch.from_numpy(np.asarray(self.raw[key]))
        if dtype is not None and t.dtype.is_floating_point:
            t = t.to(dtype)
        return t

    def has(self, key):
        return key in self.raw.files

    def state_dict(self, dtype=torch.float64):
        return self.group("sd/", dtype)

    def batch(self, dtype=torch.float64):
        return self.group("batch/", dtype)

    def masks(self, dtype=torch.float64, prefix="mask/"):
        m = self.group(prefix, dtype)
        return m if m else None


def rel_err(a, b):
    """max |a-b| / max(1, max|b|) in fp64."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return ((a - b).abs().max() / max(1.0, float(b.abs().max()))).item()


def max_rel(a, b, floor=1e-3):
    """max elementwise |a-b| / max(|b|, floor)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return ((a - b).abs() / b.abs().clamp(min=floor)).max().item()


def report(line):
    """Print a measured-error line and, when LFI_PARITY_REPORT names a file, append it there: the GPU run's numbers end up in
    profiles/parity_report_*.txt instead of being swallowed by `pytest -q`."""
    print(line)
    path = os.environ.get("LFI_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")
