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


def elem_excess(a, b, rtol=1e-4, floor=1e-6):
    """Element-wise relative check (north_star: "within 1e-4 relative error" for numeric columns and losses): the largest
    |a - b| / (rtol * |b| + floor * max|b|) over all elements; <= 1 means every element is within `rtol` of its own
    reference value, with an absolute floor of `floor` x the tensor's largest magnitude for elements near zero (a
    max-norm ratio like `relerr` hides errors of small elements next to large ones)."""
    a, b = a.double(), b.double()
    tol = rtol * b.abs() + floor * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / tol).max())


@pytest.fixture(scope="session")
def gold():
    return golden
