#!/usr/bin/env python3
"""Tracker launch time by window size and sampling variant (track impl 0 = default choice, 2 = gathers,
3 = LDS-staged footprint), 1080p, 8000 features, single-frame launches, event-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

H, W, n = 1080, 1920, 8000
for win in [int(v) for v in (sys.argv[1:] or ["9", "11", "13"])]:
    gp = klt.make_params(window=win, levels=4)
    ctx = klt.Context(0, gp)
    T = synth.base_texture(H, W)
    f0, f1 = synth.render_frame(T, H, W, 0), synth.render_frame(T, H, W, 1)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    p0.build(f0); p1.build(f1)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    v = np.zeros(len(x), np.int32)
    fl = klt.FeatureList(ctx, len(x))
    for impl in (0, 2, 3):
        ctx.set_track_impl(impl)
        for rep in range(3):
            fl.upload(x, y, v); klt.track_flist(ctx, p0, p1, fl)
        ctx.sync(); ctx.profile_enable(True); ctx.profile_reset()
        for rep in range(20):
            fl.upload(x, y, v); klt.track_flist(ctx, p0, p1, fl)
        t = ctx.profile_get()["track"]; ctx.profile_enable(False)
        print("window %d impl %d: %.1f us/launch (tracked %d/%d)" % (win, impl, t[0] / t[1] * 1e3, (fl.download()[2] == 0).sum(), len(x)), flush=True)
    fl.close(); p0.close(); p1.close(); ctx.close()
