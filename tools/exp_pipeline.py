#!/usr/bin/env python3
"""Experiment: where does a pipelined frame go -- host enqueue or GPU?  (graph vs eager)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gslam_amd import klt, synth

cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 8)
sx, sy = synth.lattice_seeds(H, W, n, gp.borderx)
seeds = (sx, sy, np.zeros(n, np.int32))
st = bench.Stream(ctx, cfg, frames, seeds)
st.start()
for _ in range(64): st.step()
st.sync()
for mode in (False,):
    for trial in range(4):
        K = 512
        t0 = time.perf_counter()
        for _ in range(K): st.step()
        t1 = time.perf_counter()
        st.sync()
        t2 = time.perf_counter()
        print("%s: host enqueue %.1f us/step, total %.1f us/step" % ("graph" if mode else "eager", (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
st.close(); ctx.close()
