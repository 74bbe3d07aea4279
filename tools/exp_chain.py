#!/usr/bin/env python3
"""A/B in one process: frame-at-a-time steps with levels 1..L-1 in one launch (gklt_ctx_set_chain 1)
or one launch per level (0); also the per-kernel times of each."""
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
K = 512
for trial in range(3):
    for chain in (1, 0):
        ctx.set_chain(chain)
        for _ in range(32): st.step()
        st.sync()
        t0 = time.perf_counter()
        for _ in range(K): st.step()
        t1 = time.perf_counter()
        st.sync()
        t2 = time.perf_counter()
        print("chain %d: host enqueue %.1f us/step, total %.1f us/step" % (chain, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
for chain in (1, 0):
    ctx.set_chain(chain)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(128): st.step()
    st.sync()
    prof = ctx.profile_get(); ctx.profile_enable(False)
    print("chain %d per launch (us):" % chain, {k: (round(ms / c * 1e3, 1), c) for k, (ms, c) in prof.items() if c})
st.close(); ctx.close()
