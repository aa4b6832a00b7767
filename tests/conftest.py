import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    def load(name):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def datasets():
    """Integer-id triples of the reference's data/ splits (sorted-string id maps; tests/golden/make_golden.py)."""
    import numpy as np
    return np.load(os.path.join(GOLDEN, "datasets.npz"))
