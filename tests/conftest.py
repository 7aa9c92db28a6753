import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "numpy-nn-model_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _seed_global_rng(request):
    """The layers draw their initial weights from NumPy's GLOBAL generator (as the reference's do); seed it per test
    (from the test id), so that a test which builds a layer sees the same weights in every run."""
    import zlib

    import numpy as np
    np.random.seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))

    return load
