#!/usr/bin/env python3
"""Experiment: back-to-back lookahead builds with an empty feature list (level kernels alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth
cfg = dict(synth.CONFIGS[os.environ.get("CFG", "C2")]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
ctx = klt.Context(0, gp)
NFR = int(os.environ.get("NFR", "16"))
frames = synth.make_stream(H, W, NFR)
ptrs = [ctx.device_upload(f) for f in frames]
B = 16
la = klt.Lookahead(ctx, W, H, n, B)
e = np.zeros(0, np.float32)
la.upload_features(e, e, np.zeros(0, np.int32))
la.start((ptrs[0], W))
fr = None
for rep in range(2):
    if rep:
        ctx.profile_enable(True); ctx.profile_reset()
    q0 = time.perf_counter()
    for k in range(32):
        la.submit([(ptrs[(k * B + i) % NFR], W) for i in range(B)])
    la.flush()
    dt = time.perf_counter() - q0
prof = ctx.profile_get()
print("level kernels alone, %d distinct frames: %.2f us/frame wall; " % (NFR, dt / (32 * B) * 1e6) +
      "; ".join("%s %.1f us/launch" % (k, ms / c * 1e3) for k, (ms, c) in prof.items() if c))
