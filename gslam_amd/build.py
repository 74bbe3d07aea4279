"""Builds gslam_amd/libgklt.so (HIP, gfx950 only) in-tree with hipcc.  No GPU needed."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgklt.so")
SOURCES = ["gklt_host.cpp", "gklt_api.hip", "gklt_pyramid.hip", "gklt_track.hip", "gklt_pipeline.hip", "gklt_lookahead.hip", "gklt_multistream.hip", "gklt_bearing.hip",
           "gklt_select.hip"]
HEADERS = ["gklt_internal.h", os.path.join("..", "..", "include", "gklt.h")]

# -ffp-contract=off: the reference's arithmetic is unfused AVX (no FMA); fusing mul+add on the
# GPU would change borderline status flags.  Correctly rounded fp32 divide/sqrt is hipcc's
# default and is requested explicitly because parity depends on it.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-result",
]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + ["-x", "hip"] + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True)
    print(LIB)
