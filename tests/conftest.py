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
def golden():
    def _load(name):
        z = np.load(os.path.join(GOLDEN, name))
        return {k: z[k] for k in z.files}
    return _load


@pytest.fixture(scope="session")
def luts(golden):
    return golden("g1_luts.npz")


@pytest.fixture(scope="session")
def oracle_c():
    from oracle import clib
    clib.build()
    return clib
