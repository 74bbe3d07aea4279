"""Builds gslam_amd/libgklt.so (HIP, gfx950 only) in-tree with hipcc.  No GPU needed.

Each translation unit is compiled to an object under gslam_amd/build/ (in parallel, only when it
or a header changed) and the objects are linked into the shared library."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgklt.so")
SOURCES = ["gklt_host.cpp", "gklt_api.hip", "gklt_pyramid.hip", "gklt_pyramid_stream.hip", "gklt_track.hip", "gklt_pipeline.hip",
           "gklt_lookahead.hip", "gklt_multistream.hip", "gklt_bearing.hip", "gklt_select.hip",
           "gklt_ingest.hip", "gklt_ransac.hip"]
HEADERS = ["gklt_internal.h", os.path.join("..", "..", "include", "gklt.h"),
           os.path.join("..", "..", "include", "gklt_order.h")]

# -ffp-contract=off: the reference's arithmetic is unfused AVX (no FMA); fusing mul+add on the
# GPU would change borderline status flags.  Correctly rounded fp32 divide/sqrt is hipcc's
# default and is requested explicitly because parity depends on it.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-result",
]


FMA_SRC = os.path.join(CSRC, "gklt_pyramid.hip")
FMA_OBJ = os.path.join(OBJ, "gklt_pyramid_fma.hip.o")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src) + ".o")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale():
    return _newer(LIB, sources() + _deps())


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src in sources():
        if force or extra or _newer(_obj(src), [src] + _deps()):
            jobs.append([hipcc] + FLAGS + list(extra) + ["-x", "hip", "-c", src, "-o", _obj(src)])
    # the fused level kernels a second time with mul+add contraction: pyramid impl 4 (tolerance
    # variant, never the default; see gklt_pyramid.hip)
    if force or extra or _newer(FMA_OBJ, [FMA_SRC] + _deps()):
        flags = [f for f in FLAGS if f != "-ffp-contract=off"] + ["-ffp-contract=fast", "-DGKLT_PYR_FMA", "-Wno-unused-function",
                                                                        "-Wno-unneeded-internal-declaration"]
        jobs.append([hipcc] + flags + list(extra) + ["-x", "hip", "-c", FMA_SRC, "-o", FMA_OBJ])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in sources()] + [FMA_OBJ, "-o", LIB])
    return LIB


ROOT = os.path.dirname(HERE)
SHIM_SRC = [os.path.join(ROOT, "shim", "gklt_shim.cpp"), os.path.join(ROOT, "tests", "shim_mock", "shim_driver.cpp")]
SHIM_TEST_LIB = os.path.join(ROOT, "tests", "shim_mock", "libgklt_shimtest.so")


def shim_deps():
    deps = list(SHIM_SRC) + [os.path.join(ROOT, "include", "gklt.h"), LIB]
    for d in (os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests", "shim_mock")):
        for dirpath, _, files in os.walk(d):
            deps += [os.path.join(dirpath, f) for f in files if not f.endswith(".so")]
    return deps


def build_shim_test(force=False, verbose=False):
    """shim/gklt_shim.cpp (GSLAM's C++ host API over the C ABI) + the test driver, compiled with g++
    against the cv::Mat / Eigen stand-ins of tests/shim_mock/ and linked to libgklt.so.  Host code
    only: proves that the shim compiles against include/gklt.h and lets tests/test_shim.py run it."""
    if not force and not _newer(SHIM_TEST_LIB, shim_deps()):
        return SHIM_TEST_LIB
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Werror", "-Wno-write-strings",
           "-I" + os.path.join(ROOT, "tests", "shim_mock"), "-I" + os.path.join(ROOT, "shim"),
           "-I" + os.path.join(ROOT, "include")] + SHIM_SRC + [
           "-o", SHIM_TEST_LIB, "-L" + HERE, "-lgklt", "-Wl,-rpath,$ORIGIN/../../gslam_amd", "-Wl,--no-undefined", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SHIM_TEST_LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True)
    build_shim_test(force="-f" in sys.argv, verbose=True)
    print(LIB)
