import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    """name -> dict of torch tensors loaded from tests/golden/<name>.npz"""
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
                cache[name] = {k: torch.from_numpy(z[k]) for k in z.files}
        return cache[name]
    return load


def rel_err(a, b):
    """(l2 relative error, max-abs relative to max-abs) of a against reference b."""
    a, b = a.double().cpu(), b.double().cpu()
    l2 = (a - b).norm() / b.norm().clamp_min(1e-30)
    mx = (a - b).abs().max() / b.abs().max().clamp_min(1e-30)
    return float(l2), float(mx)


def assert_close(a, b, tol=1e-4, what=""):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} != {tuple(b.shape)}"
    a, b = a.detach(), b.detach()
    if float(b.abs().max()) == 0.0:            # all-zero reference: absolute check (rounding residue only)
        assert float(a.abs().max()) <= 1e-6, f"{what}: expected zeros, max abs {float(a.abs().max()):.3e}"
        return
    l2, mx = rel_err(a, b)
    assert l2 <= tol and mx <= tol, f"{what}: rel l2 {l2:.3e}, rel max {mx:.3e} > {tol:g}"
