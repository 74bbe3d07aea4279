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


@pytest.fixture(autouse=True)
def _pinned_memory_outlives_the_report():
    """Tests compare views of page-locked host buffers (Context.pinned_array) and free the buffers in a
    `finally`.  When such a comparison fails, pytest formats the failing frame's locals -- those views
    -- AFTER the `finally` has run, i.e. it reads freed memory: round 2 saw a real mismatch surface
    as "Fatal Python error: Segmentation fault" with no test name.  Every pinned_free of a test is
    therefore postponed to this fixture's teardown, which pytest runs after the call's report."""
    from gslam_amd import klt as k

    real = k.Context.pinned_free
    pending = []
    k.Context.pinned_free = lambda self, ptr: pending.append((self, ptr))
    try:
        yield
    finally:
        k.Context.pinned_free = real
        for ctx, ptr in pending:
            if ctx.h:               # (a context the test closed has released nothing of these)
                real(ctx, ptr)
