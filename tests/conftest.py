import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden2010():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "loopclosure2010.npz"))
