import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # libbbt_hip.so is built in-tree (git-ignored): build it if this checkout
    # does not have it yet and hipcc is around.  Never a substitute path: if the
    # build is impossible the tests that need the library fail.
    lib = os.path.join(ROOT, 'baseband-tasks_amd', 'lib', 'libbbt_hip.so')
    if not os.path.exists(lib) and os.path.exists('/opt/rocm/bin/hipcc'):
        import __graft_entry__
        __graft_entry__.build()


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
