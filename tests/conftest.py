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


@pytest.fixture(scope="session")
def cartpole_golden():
    return np.load(os.path.join(GOLDEN, "cartpole_golden.npz"))


@pytest.fixture(scope="session")
def mujoco_golden():
    return np.load(os.path.join(GOLDEN, "mujoco_firstparty_golden.npz"))


@pytest.fixture(scope="session")
def hopper_golden():
    return np.load(os.path.join(GOLDEN, "hopper_firstparty_golden.npz"))


@pytest.fixture(scope="session")
def lagrange_golden():
    return np.load(os.path.join(GOLDEN, "lagrange_golden.npz"))


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor) ignoring rows where both are NaN."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), floor)
    d = np.where(both_nan | same_inf, 0.0, d)
    d = np.where(np.isnan(d), np.inf, d)
    return float(d.max()) if d.size else 0.0
