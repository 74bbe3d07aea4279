#!/usr/bin/env python3
"""Tracker launch time against max_iterations: is a launch as long as its slowest feature?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

name = sys.argv[1] if len(sys.argv) > 1 else "Cship"
cfg = synth.CONFIGS[name]; H, W, n = cfg["H"], cfg["W"], cfg["n"]
T = synth.base_texture(H, W)
f0, f1 = synth.render_frame(T, H, W, 0), synth.render_frame(T, H, W, 1)
IMPL = int(os.environ.get('IMPL', '0'))
for mi in tuple(int(v) for v in os.environ.get('ITERS', '1,2,3,5,10,20').split(',')):
    gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
    gp.max_iterations = mi
    ctx = klt.Context(0, gp)
    ctx.set_track_impl(IMPL)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    p0.build(f0); p1.build(f1)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    v = np.zeros(n, np.int32)
    fl = klt.FeatureList(ctx, n)
    for nn in tuple(int(v) for v in os.environ.get('NS', '%d,256,1' % n).split(',')):
        xs, ys, vs = x[:nn].copy(), y[:nn].copy(), v[:nn].copy()
        fls = klt.FeatureList(ctx, nn)
        for rep in range(3):
            fls.upload(xs, ys, vs); klt.track_flist(ctx, p0, p1, fls)
        ctx.sync(); ctx.profile_enable(True); ctx.profile_reset(); fls.stats(reset=True)
        for rep in range(20):
            fls.upload(xs, ys, vs); klt.track_flist(ctx, p0, p1, fls)
        t = ctx.profile_get()["track"]; ctx.profile_enable(False)
        trk, it = fls.stats(reset=True)
        print("%s max_iter %2d n %5d: %6.1f us/launch, %.2f iters/feature" % (name, mi, nn, t[0] / t[1] * 1e3, it / 20.0 / nn), flush=True)
        fls.close()
    fl.close(); p0.close(); p1.close(); ctx.close()
