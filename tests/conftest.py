import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_numpy.npz"))


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def ofk(pkg):
    import of_amd.ofk as m
    return m


@pytest.fixture(scope="session")
def gpu_ctx(ofk):
    """One context big enough for every GPU test (1080p, batch 4, 2048 points, 5 levels)."""
    c = ofk.Context(0, 1920, 1080, 4, 2048, 5)
    yield c
    c.close()
