import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope='session')
def golden():
    """Vectors produced by the real reference (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_vectors.npz'))


def rel_l2(a, b):
    a = np.asarray(a, np.complex128)
    b = np.asarray(b, np.complex128)
    return np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())


def max_over_rms(a, b):
    a = np.asarray(a, np.complex128)
    b = np.asarray(b, np.complex128)
    return np.abs(a - b).max() / np.sqrt(np.mean(np.abs(b) ** 2))
