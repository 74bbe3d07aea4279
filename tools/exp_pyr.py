#!/usr/bin/env python3
"""Experiment: isolated pyramid builds (no tracker overlap), per-kernel event timings.
usage: exp_pyr.py [C2|C3|Cship] [impl ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]); H, W = cfg["H"], cfg["W"]
impls = [int(a) for a in sys.argv[2:]] or [0, 1]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 4)
ptrs = [ctx.device_upload(f) for f in frames]
pyr = klt.Pyramid(ctx, W, H)
for impl in impls:
    ctx.set_pyramid_impl(impl)
    for k in range(8):
        pyr.build_from_device(ptrs[k % 4], W)
    ctx.sync()
    ctx.profile_enable(True); ctx.profile_reset()
    for k in range(64):
        pyr.build_from_device(ptrs[k % 4], W)
    ctx.sync()
    prof = ctx.profile_get(); ctx.profile_enable(False)
    tot = 0.0
    parts = []
    for name, (ms, n) in prof.items():
        if n:
            parts.append("%s %.2f us x%d" % (name, ms / n * 1e3, n // 64))
            tot += ms / 64 * 1e3
    print("%s impl %d: %.1f us/build  [%s]" % (sys.argv[1] if len(sys.argv) > 1 else "C2", impl, tot, ", ".join(parts)))
pyr.close(); ctx.close()
