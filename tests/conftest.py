import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from _pkg import load_pkg

    return load_pkg()


@pytest.fixture(scope="session")
def ctx(pkg):
    """A libmotifs_hip context on device 0; fails loudly when the library or the GPU is missing."""
    c = pkg._lib.Context(0)
    yield c
    c.close()
