import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU tests that take a minute or more (the sanitizer run); still part of -m 'not gpu'")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the reference itself (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.build()
    return oracle


def geodesic(Ra, Rb):
    """Rotation angle of Ra^T Rb, accurate for tiny angles (atan2 form)."""
    D = np.asarray(Ra).T @ np.asarray(Rb)
    s = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    return float(np.arctan2(np.linalg.norm(s), 0.5 * (np.trace(D) - 1.0)))
