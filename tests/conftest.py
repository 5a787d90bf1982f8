import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'within 1e-3 rel' metric of BASELINE.json north_star."""
    import torch
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
