#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/klt_oracle.c).

The reference has no tests or golden vectors and cannot be built or run in this image
(SURVEY.md 4 / 8c), so these fixtures do not come from the reference: they pin the oracle's
own behaviour (regressions, compiler changes) on small seeded inputs.  Parity with the literal
reference stays "unpinned" (see oracle/klt_oracle.h).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gslam_amd import synth  # noqa: E402
from oracle import oracle as ok  # noqa: E402


def main():
    ok.build()
    out = {}
    # kernels / borders
    for s in (0.7, 1.0, 1.1, 1.8, 2.1):
        g, d = ok.compute_kernels(s)
        out["gauss_%s" % s] = g
        out["deriv_%s" % s] = d
    out["borders"] = np.array([[w, L, ok.make_params(window=w, levels=L).borderx]
                               for (w, L) in ((21, 2), (7, 4), (11, 5), (5, 3), (15, 3))], np.int32)
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)

    # pyramid + track + select on a 96x128 stream
    H, W, n = 96, 128, 96
    T = synth.base_texture(H, W, seed=77)
    f0, f1 = synth.render_frame(T, H, W, 0, seed=77), synth.render_frame(T, H, W, 1, seed=77)
    p = ok.make_params(window=7, levels=3)
    P0, P1 = ok.compute_pyramid(p, f0), ok.compute_pyramid(p, f1)
    x, y = synth.lattice_seeds(H, W, n, p.borderx, seed=77)
    val = np.zeros(n, np.int32)
    K = np.array([[100.0, 0, 63.5], [0, 100.0, 47.5], [0, 0, 1]])
    x2, y2, v2, norm, vec = ok.track_features(p, P0, P1, x, y, val, np.linalg.inv(K))
    sxx, syy, sxy = ok.integral_products(P0[1], P0[2])
    q = ok.make_params(window=7, levels=3, border=8, min_eigenvalue=50, mindist=6)
    sel = {}
    for name, variant in (("plain", ok.VARIANT_PLAIN), ("uniform", ok.VARIANT_UNIFORM)):
        fx, fy, fv = ok.select_features(q, ok.SELECT_ALL, variant, sxx, syy, sxy, 40, grid=(16, 16))
        sel[name] = np.stack([fx, fy, fv.astype(np.float32)])
    np.savez_compressed(
        os.path.join(HERE, "stream96x128.npz"), f0=f0, f1=f1, x=x, y=y,
        I2=P0[6], gx2=P0[7], gy2=P0[8], gx0_row40=P0[1][40], I1_row10=P0[3][10],
        x2=x2, y2=y2, v2=v2, norm=norm, vec=vec, sumxx_row50=sxx[50], sel_plain=sel["plain"],
        sel_uniform=sel["uniform"])
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
