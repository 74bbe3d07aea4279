"""SURVEY section 5 ("ASAN host build"): the host-side code -- the CPU oracle, gklt_host.cpp and the
shim's host-only entry points -- built with AddressSanitizer + UndefinedBehaviorSanitizer and run on
a small case, in the build container (CPU only; no GPU entry point is called)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_code_under_asan_and_ubsan(tmp_path):
    from gslam_amd import build as b

    b.build()  # the shim's GPU entry points resolve against libgklt.so (never called here)
    exe = str(tmp_path / "san_driver")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
           "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = ["-I" + os.path.join(ROOT, "tests", "shim_mock"), "-I" + os.path.join(ROOT, "shim"),
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]  # (host pass of hip_runtime.h)
    objs = []
    for src, cc, std in [("oracle/klt_oracle.c", "gcc", "-std=c99"), ("oracle/ransac_oracle.c", "gcc", "-std=c99"),
                         ("gslam_amd/csrc/gklt_host.cpp", "g++", "-std=c++17"),
                         ("shim/gklt_shim.cpp", "g++", "-std=c++17"),
                         ("tests/sanitize/san_driver.cpp", "g++", "-std=c++17")]:
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call([cc, std, "-c", os.path.join(ROOT, src), "-o", obj, "-Wno-write-strings",
                               "-Wno-unknown-pragmas", "-mavx", "-mno-fma", "-ffp-contract=off"] + san + inc)
        objs.append(obj)
    # gklt_host.cpp is linked in instrumented form AND exported by libgklt.so: the executable's own
    # (instrumented) definitions win
    subprocess.check_call(["g++"] + objs + ["-o", exe, "-L" + os.path.join(ROOT, "gslam_amd"), "-lgklt",
                                            "-Wl,-rpath," + os.path.join(ROOT, "gslam_amd"), "-lm", "-lpthread"] + san)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "sanitizer driver: ok" in r.stdout
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
