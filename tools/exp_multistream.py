#!/usr/bin/env python3
"""BASELINE config C4 on one GPU through gklt_multistream: S C2 streams stepped together, one
frame of each per call (no lookahead).  Frames resident in HBM; stream s plays the synthetic
sequence from phase 5*s (frames are read-only, so the streams share the device copies).
usage: exp_multistream.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

cfgname = os.environ.get("GKLT_CONFIG", "C2")
cfg = dict(synth.CONFIGS[cfgname]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
ctx = klt.Context(0, gp)
ctx.set_track_waves(int(os.environ.get("GKLT_TRACK_WAVES", "0")))
NF = 64
frames = synth.make_stream(H, W, NF, seed=synth.BASE_SEED)
dptr = [ctx.device_upload(f) for f in frames]
sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
sv = np.zeros(n, np.int32)
for S in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64]:
    ms = klt.MultiStream(ctx, S, W, H, n)
    host = [ctx.pinned_array((3, S * n), np.float32) for _ in range(4)]
    fr = lambda t: [(dptr[(t + 5 * s) % NF], W) for s in range(S)]
    def seed():
        for s in range(S):
            ms.upload_features(s, sx, sy, sv)
    seed()
    ms.start(fr(0), fr(1))
    t = 1
    def run(k):
        global t
        for _ in range(k):
            if t % 32 == 0:
                seed()
            ms.step(fr(t + 1), host[t % 4][1])
            t += 1
        ms.sync()
    run(max(4, 64 // S))
    ms.stats(reset=True)
    K = max(16, 2048 // S)
    q0 = time.perf_counter()
    run(K)
    dt = time.perf_counter() - q0
    tracked, iters = ms.stats(reset=True)
    # latency of one step: call -> all lists in pinned host memory
    lat = []
    for _ in range(32):
        q0 = time.perf_counter()
        ms.step(fr(t + 1), host[t % 4][1]); t += 1
        ms.wait_tracked()
        lat.append((time.perf_counter() - q0) * 1e3)
        ms.sync()
    print("S=%d streams: %.1f M tracked features*frames/s, %.0f frames/s aggregate, %.2f us/frame, "
          "step %.3f ms, lists after %.3f ms (p50)" % (S, tracked / dt / 1e6, S * K / dt, dt / (S * K) * 1e6,
                                                         dt / K * 1e3, float(np.median(lat))), flush=True)
    ms.close()
    for _, p in host:
        ctx.pinned_free(p)
ctx.close()
