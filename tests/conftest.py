import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see imgadapointr_amd/__init__.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    """The CPU oracle (test infrastructure)."""
    from oracle import ref as _ref
    _ref.build()
    return _ref


@pytest.fixture(scope="session")
def native():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imgadapointr_amd import _native
    _native.load()
    return _native
