import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/rodio_oracle.py).  Test infrastructure only."""
    from oracle import rodio_oracle

    return rodio_oracle


@pytest.fixture(scope="session")
def rh():
    """The product package, with the HIP library loaded.  Fails loudly if the .so is missing."""
    import rodio_amd

    return rodio_amd
