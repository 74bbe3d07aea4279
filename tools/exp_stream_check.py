#!/usr/bin/env python3
"""Streaming level kernel (pyramid impl 3) against the oracle, plane by plane, with the positions
of any mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth
from oracle import oracle as ok
ok.build(); ok.set_threads(8)
for (H, W, win, L) in [(192, 320, 7, 4), (1080, 1920, 7, 4), (1080, 1920, 21, 2), (400, 600, 11, 3)]:
    T = synth.base_texture(H, W, seed=3)
    f = synth.render_frame(T, H, W, 0, seed=3)
    gp = klt.make_params(window=win, levels=L); op = ok.make_params(window=win, levels=L)
    ctx = klt.Context(0, gp); ctx.set_pyramid_impl(3)
    py = klt.Pyramid(ctx, W, H); py.build(f)
    got = py.planes(); ref = ok.compute_pyramid(op, f)
    for i, (a, b) in enumerate(zip(got, ref)):
        bad = a.view(np.uint32) != b.view(np.uint32)
        if bad.any():
            ys, xs = np.nonzero(bad)
            print(H, W, win, "plane", i, "level", i // 3, "bad", bad.sum(), "x range", xs.min(), xs.max(), "uniq x", np.unique(xs)[:20], "y", np.unique(ys)[:10])
        else:
            print(H, W, win, "plane", i, "ok")
    py.close(); ctx.close()
