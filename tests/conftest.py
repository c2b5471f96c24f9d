import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


SIZES_1D = [4, 5, 7, 8, 9, 10, 21, 22, 100, 128, 134, 144, 160, 176, 192]
SHAPES_2D = [(4, 6), (4, 7), (5, 6), (5, 7), (5, 8), (6, 7), (6, 8), (7, 9), (8, 10), (35, 60),
             (64, 64), (2, 2), (3, 2), (2, 3), (16, 12), (12, 16), (27, 25), (1, 8), (1, 7)]
KERNEL_NAMES = ["expquad", "matern32", "matern52"]


def variants(g):
    """(kernel, tag) pairs present in a golden file."""
    out = []
    for name in KERNEL_NAMES:
        for tag in "ab":
            if any(k.startswith(f"{name}_{tag}_cov_rfft") for k in g):
                out.append((name, tag))
    return out


@pytest.fixture(scope="session")
def gpu_lib():
    """The shipped HIP library on a real GPU; GPU tests never fall back to anything else."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import gptools_amd

    return gptools_amd
