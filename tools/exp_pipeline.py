#!/usr/bin/env python3
"""Experiment: where does a pipelined frame go -- host enqueue or GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gslam_amd import klt, synth

cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"])
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 8)
sx, sy = synth.lattice_seeds(H, W, n, gp.borderx)
seeds = (sx, sy, np.zeros(n, np.int32))
dev = torch.device("cuda", 0)
fd = [torch.from_numpy(f).to(dev) for f in frames]
st = bench.Stream(ctx, cfg, fd, [f.data_ptr() for f in fd], seeds)
st.start()
for _ in range(64): st.step()
ctx.sync()
for trial in range(6):
    K = 512
    t0 = time.perf_counter()
    for _ in range(K): st.step()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("async: host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# individual API costs
t0 = time.perf_counter()
for _ in range(K): st.ring[0].build_from_device(st.frame_ptr(0), W)
t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
print("pyramid only: host %.1f us, total %.1f us" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
t0 = time.perf_counter()
for _ in range(K): klt.track_flist(ctx, st.ring[1], st.ring[2], st.fl)
t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
print("track only: host %.1f us, total %.1f us" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
