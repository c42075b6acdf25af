import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def ctx():
    """One plk_ctx on cuda:0.  Fails (never skips) when the library or the GPU is missing."""
    from plonk_amd.hip import Context
    c = Context(0)
    yield c
    c.close()
