#!/usr/bin/env python3
"""BASELINE config C4 on one GPU: S independent C2 streams (own frames, lists, lookahead
pipelines and HIP streams) sharing the device.  usage: exp_streams.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gslam_amd import klt, synth

cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"])
ctx = klt.Context(0, gp)
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    sts = []
    for s in range(S):
        seed = synth.BASE_SEED + 1000 * s
        frames = synth.make_stream(H, W, 16, seed=seed)
        sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=seed, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
        st = bench.LookaheadStream(ctx, cfg, frames, (sx, sy, np.zeros(len(sx), np.int32)), 16)
        st.start()
        sts.append(st)
    def run(batches):
        for _ in range(batches):
            for st in sts:
                st.run(16)
        for st in sts:
            st.sync()
    run(4)
    for st in sts:
        st.fl.stats(reset=True)
    K = 48
    q0 = time.perf_counter()
    run(K)
    dt = time.perf_counter() - q0
    tracked = sum(st.fl.stats(reset=True)[0] for st in sts)
    print("S=%d streams: %.1f M tracked features*frames/s aggregate, %.0f frames/s aggregate, %.2f us/frame" % (
        S, tracked / dt / 1e6, S * K * 16 / dt, dt / (S * K * 16) * 1e6))
    for st in sts:
        st.close()
ctx.close()
