"""A bounded, seeded slice of the randomised parity sweeps (tools/fuzz_track.py, fuzz_pyramid.py,
fuzz_select.py, fuzz_lookahead.py: random window shapes, level counts, ragged images, list sizes,
selector flavours / round kernels, lookahead batch patterns), each in a child process as the tools
are written, so that the sweeps the design document quotes are part of the suite the driver runs.
Every comparison inside them is bit for bit / slot by slot against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_track.py", 14, 303), ("fuzz_pyramid.py", 12, 304),
                                             ("fuzz_select.py", 14, 305), ("fuzz_lookahead.py", 5, 306)])
def test_fuzz_slice(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)],
                       capture_output=True, text=True, timeout=110)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert "MISMATCH" not in r.stdout and ", 0 mismatches" in r.stdout, tail
