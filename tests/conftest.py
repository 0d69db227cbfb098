import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by executing the reference's own Python on CPU (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "window_attention_1000.npz")
    return dict(np.load(path, allow_pickle=False))
