import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    """Load a committed golden fixture as a dict of torch tensors."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def sub(d, prefix):
    """Entries of `d` whose key starts with `prefix`, prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def gold():
    return golden
