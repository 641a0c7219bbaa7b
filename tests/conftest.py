import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = {k[5:]: float(z[k]) for k in z.files if k.startswith("meta_")}
    return z, meta


def sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def checksum(arr, salt):
    """Same definition as oracle/gen_golden.py:checksum."""
    a = np.asarray(arr, dtype=np.float64).ravel()
    idx = np.random.default_rng(1000 + salt).integers(0, a.size, size=16)
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[idx]])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
