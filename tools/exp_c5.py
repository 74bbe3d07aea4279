#!/usr/bin/env python3
"""BASELINE config C5: C2 frames, re-detection (Uniform selector, 32x32 cells, min_eigenvalue 100,
mindist 10, N = 4000) from the current pyramid every frame, then track to the next frame.
Everything stays on the device (pyramid ring, list).  Prints frames/s and per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

VAR = {"uniform": klt.VARIANT_UNIFORM, "plain": klt.VARIANT_PLAIN, "round2": klt.VARIANT_ROUND2}[
    sys.argv[1] if len(sys.argv) > 1 else "uniform"]
cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=7, levels=4, min_eigenvalue=100, mindist=10)
gp.bordery = max(gp.bordery, 24)
ctx = klt.Context(0, gp)
if len(sys.argv) > 2:
    ctx.set_select_iters(int(sys.argv[2]))
frames = synth.make_stream(H, W, 16)
ptrs = [ctx.device_upload(f) for f in frames]
ring = [klt.Pyramid(ctx, W, H) for _ in range(3)]
fl = klt.FeatureList(ctx, 100 * n)
fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))


def run(K, t0=0):
    for t in range(t0, t0 + K):
        cur, nxt = ring[t % 3], ring[(t + 1) % 3]
        nxt.build_from_device(ptrs[(t + 1) % 16], W)
        klt.select_from_pyramid(ctx, cur, klt.SELECT_ALL, VAR, fl, grid=(32, 32))
        klt.track_flist(ctx, cur, nxt, fl)
    ctx.sync()


ring[0].build_from_device(ptrs[0], W)
run(16)
fl.stats(reset=True)
K = 128
q0 = time.perf_counter()
run(K, 16)
dt = time.perf_counter() - q0
tracked, iters = fl.stats(reset=True)
ctx.profile_enable(True); ctx.profile_reset()
run(32, 16 + K)
prof = ctx.profile_get(); ctx.profile_enable(False)
print("C5: %.1f frames/s, %.2f ms/frame, tracked/frame %.0f, list size %d, %.2fM tracked features*frames/s" % (
    K / dt, dt / K * 1e3, tracked / K, len(fl), tracked / dt / 1e6))
print("    suppression launches of the last selection (phase A, phase B):", ctx.select_rounds())
print("    per frame (us):", {k: round(ms / 32 * 1e3, 1) for k, (ms, c) in prof.items() if c})
for p in ring:
    p.close()
fl.close(); ctx.close()
