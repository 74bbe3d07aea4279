#!/usr/bin/env python3
"""PCIe-inclusive rate: frames start in page-locked HOST memory (staged H2D by the library), the
list comes back into pinned host memory.  gklt_pipeline_step, C2.  Never bench.py's `value`."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"])
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 16)
host = []
for f in frames:
    a, p = ctx.pinned_array((H, W), np.uint8)
    a[:] = f
    host.append((a, p))
out, outp = ctx.pinned_array((3, n), np.float32)
sx, sy = synth.lattice_seeds(H, W, n, gp.borderx)
pipe = klt.Pipeline(ctx, W, H, n)
pipe.upload_features(sx, sy, np.zeros(n, np.int32))
L = klt.lib()
u8p = C.POINTER(C.c_uint8)
fp = lambda k: C.cast(C.c_void_p(host[k % 16][1]), u8p)
klt._check(L.gklt_pipeline_start(pipe.h, fp(0), fp(1), W, 0))
t = 1
for K in (64, 512):
    q0 = time.perf_counter()
    for _ in range(K):
        klt._check(L.gklt_pipeline_step(pipe.h, fp(t + 1), W, 0, C.c_void_p(outp)))
        t += 1
    pipe.sync()
    dt = time.perf_counter() - q0
print("PCIe-inclusive C2 (pinned host frames in, pinned list out): %.1f us/frame, %.0f frames/s" % (dt / K * 1e6, K / dt))
lat = []
for _ in range(128):
    q0 = time.perf_counter()
    klt._check(L.gklt_pipeline_step(pipe.h, fp(t + 1), W, 0, C.c_void_p(outp)))
    pipe.sync()
    lat.append(time.perf_counter() - q0)
    t += 1
print("  synchronous latency incl. H2D of the next frame: p50 %.1f us, p99 %.1f us" % (
    np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6))
pipe.close()
ctx.close()
