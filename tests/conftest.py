import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ok():
    """The CPU oracle (oracle/), built on demand with gcc."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def klt():
    from gslam_amd import klt as k

    k.lib()  # raises if libgklt.so is missing: there is no CPU fallback
    return k


@pytest.fixture(scope="session")
def gpu_ctx(klt):
    ctx = klt.Context(0)
    yield ctx
    ctx.close()
