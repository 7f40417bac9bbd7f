import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # torch bundles its own HIP runtime: when libwrp.so has initialised the GPU first, torch finds
    # "no HIP GPUs".  GPU tests that use torch for device memory need torch to come first.
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def stage_close(a, b, rtol=1e-5, l2tol=1e-6, exclude_cols=()):
    """SURVEY.md §8(d2) stage tolerance: |a-b| <= rtol*|b| + rtol*rowmax|b| per element,
    and ||a-b||2/||b||2 <= l2tol per stage.  Returns (ok, worst_ratio, l2rel)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if exclude_cols:
        keep = np.ones(b.shape[-1], bool)
        keep[list(exclude_cols)] = False
        a = a[..., keep]
        b = b[..., keep]
    ab = np.abs(b)
    rowmax = ab.max(axis=-1, keepdims=True)
    bound = rtol * ab + rtol * rowmax
    err = np.abs(a.astype(b.dtype) - b)
    worst = float(np.max(err / np.maximum(bound, np.finfo(np.float64).tiny)))
    l2 = float(np.linalg.norm(err) / np.linalg.norm(ab))
    return (worst <= 1.0 and l2 <= l2tol), worst, l2
