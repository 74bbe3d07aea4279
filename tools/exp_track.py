#!/usr/bin/env python3
"""A/B timing of the tracker kernels alone (event-timed through gklt_profile_*)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

IMPLS = tuple(int(v) for v in os.environ.get('IMPLS', '0,2,3,1').split(','))
for name in sys.argv[1:] or ["C2", "C3", "Cship"]:
    cfg = synth.CONFIGS[name]; H, W, n = cfg["H"], cfg["W"], cfg["n"]
    gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
    ctx = klt.Context(0, gp)
    T = synth.base_texture(H, W)
    f0, f1 = synth.render_frame(T, H, W, 0), synth.render_frame(T, H, W, 1)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    p0.build(f0); p1.build(f1)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    v = np.zeros(n, np.int32)
    fl = klt.FeatureList(ctx, n)
    for impl in IMPLS:
        ctx.set_track_impl(impl)
        for rep in range(3):
            fl.upload(x, y, v)
            klt.track_flist(ctx, p0, p1, fl)
        ctx.sync(); ctx.profile_enable(True); ctx.profile_reset()
        for rep in range(20):
            fl.upload(x, y, v)
            klt.track_flist(ctx, p0, p1, fl)
        t = ctx.profile_get()["track"]; ctx.profile_enable(False)
        trk, it = fl.stats(reset=True)
        print("%s impl %d: %.1f us/launch  (tracked %d/%d)" % (name, impl, t[0] / t[1] * 1e3, (fl.download()[2] == 0).sum(), n), flush=True)
    fl.close(); p0.close(); p1.close(); ctx.close()
