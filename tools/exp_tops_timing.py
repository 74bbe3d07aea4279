#!/usr/bin/env python3
"""Where a launch of k_tops_round spends its time (experiment build, tools/build_timing.sh):
per workgroup of launch `round` of a steady-state C5 selection, the 100 MHz wall clock at entry,
after staging and at exit, and the iterations it ran."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslam_amd import klt, synth
klt.LIB_PATH = os.environ.get("GKLT_LIB") or os.path.join(ROOT, "tools", "bin", "libgklt_timing.so")
L = klt.lib()
cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=7, levels=4, min_eigenvalue=100, mindist=10)
gp.bordery = max(gp.bordery, 24)
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 4)
ring = [klt.Pyramid(ctx, W, H) for _ in range(3)]
fl = klt.FeatureList(ctx, 100 * n)
fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
for t in range(4):
    ring[t % 3].build(frames[t])
for rnd in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    L.gklt_debug_tops_round(rnd)
    for t in range(6):
        klt.select_from_pyramid(ctx, ring[t % 3], klt.SELECT_ALL, klt.VARIANT_UNIFORM, fl, grid=(32, 32))
    ctx.sync()
    buf = np.zeros(4 * 8192, np.uint64)
    L.gklt_debug_tops_timing(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
    d = buf.reshape(-1, 4)
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    start, stage, total, its = (d[:, 0] - t0) / 100.0, (d[:, 1] - d[:, 0]) / 100.0, (d[:, 2] - d[:, 0]) / 100.0, d[:, 3]
    ran = d[:, 2] > d[:, 1]
    print("launch %d: %d workgroups stamped, %d staged; start last %.1f us; kernel span %.1f us" % (
        rnd, len(d), ran.sum(), start.max(), ((d[:, 2].max() - t0) / 100.0)))
    if ran.any():
        print("   staging us: median %.1f p90 %.1f max %.1f | total us: median %.1f p90 %.1f max %.1f | iterations: median %d max %d" % (
            np.median(stage[ran]), np.percentile(stage[ran], 90), stage[ran].max(),
            np.median(total[ran]), np.percentile(total[ran], 90), total[ran].max(), np.median(its[ran]), its[ran].max()))
        print("   start us of staged workgroups: median %.1f p90 %.1f max %.1f" % (
            np.median(start[ran]), np.percentile(start[ran], 90), start[ran].max()))
