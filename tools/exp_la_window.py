#!/usr/bin/env python3
"""Lookahead-pipeline throughput by window size and tracker sampling variant (gathers vs LDS-staged
footprint): the C3 geometry (4K, 16000 features, 5 levels) with the window replaced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gslam_amd import klt, synth

base = os.environ.get("CFG", "C3")
for win in [int(v) for v in (sys.argv[1:] or ["9", "11", "13"])]:
    cfg = dict(synth.CONFIGS[base]); cfg["window"] = win
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    gp = klt.make_params(window=win, levels=cfg["levels"], border=cfg.get("border"))
    ctx = klt.Context(0, gp)
    frames = synth.make_stream(H, W, 32, seed=bench.rank_seed(0))
    sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=bench.rank_seed(0), nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    seeds = (sx, sy, np.zeros(len(sx), np.int32))
    for impl in (2, 3, 2, 3):
        ctx.set_track_impl(impl)
        st = bench.LookaheadStream(ctx, cfg, frames, seeds, 16)
        st.start(); st.run(64); st.sync(); st.fl.stats(reset=True)
        t0 = time.perf_counter()
        st.run(512); st.sync(); ctx.device_synchronize()
        dt = time.perf_counter() - t0
        tracked, _ = st.fl.stats(reset=True)
        print("%s window %d impl %d: %.1f M feature-frames/s" % (base, win, impl, tracked / dt / 1e6), flush=True)
        st.close()
    ctx.close()
