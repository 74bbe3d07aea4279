#!/usr/bin/env python3
"""Randomised parity sweep of the selector (plain / round2 / Uniform, SELECT_ALL / REPLACE_SOME)
through the C ABI against the oracle (run on a GPU box, not part of the suite): random image
sizes, mindist, thresholds, borders, skipped pixels, grids, list sizes, both round-kernel
flavours and LDS iteration counts.  Lists compared slot by slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth
from oracle import oracle as ok

ok.build(); ok.set_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = klt.Context(0, klt.make_params(window=7, levels=2))
bad = 0
for it in range(N):
    H = int(rng.integers(70, 300)); W = int(rng.integers(90, 500))
    seed = int(rng.integers(0, 1 << 30))
    T = synth.base_texture(H, W, seed=seed)
    img = synth.render_frame(T, H, W, 0, seed=seed)
    if rng.random() < 0.25:   # flat regions and a periodic pattern: few candidates, ties
        img = img.copy(); img[: H // 2, : W // 2] = 80; img[H // 2:, ::16] = 255
    window = int(rng.choice([3, 5, 7, 9, 11]))
    mindist = int(rng.choice([1, 2, 3, 5, 8, 10, 13, 20]))
    kw = dict(window=window, levels=2, border=int(rng.integers(2, 30)), mindist=mindist,
              min_eigenvalue=int(rng.choice([1, 20, 100, 400])), nSkippedPixels=int(rng.choice([0, 0, 0, 1, 3])))
    gp, op = klt.make_params(**kw), ok.make_params(**kw)
    ctx.set_params(gp)
    pyr = klt.Pyramid(ctx, W, H)
    try:
        pyr.build(img)
        gx, gy = pyr.plane(0, 1), pyr.plane(0, 2)
    finally:
        pyr.close()
    sums = ok.integral_products(gx, gy)
    variant = str(rng.choice(["PLAIN", "ROUND2", "UNIFORM"]))
    gv, ov = getattr(klt, "VARIANT_" + variant), getattr(ok, "VARIANT_" + variant)
    n = int(rng.choice([1, 10, 60, 300, 2000]))
    gw = int(rng.integers(max(8, mindist), 70)); gh = int(rng.integers(max(8, mindist), 70))
    grid = (min(gw, W), min(gh, H))
    cap = max(100 * n, 8192)
    impl, iters = int(rng.choice([0, 0, 1, 2])), int(rng.choice([1, 2, 3, 5, 12]))
    ctx.set_select_impl(impl); ctx.set_select_iters(iters)
    replace = rng.random() < 0.35
    try:
        if replace:
            x, y, v = ok.select_features(op, ok.SELECT_ALL, ok.VARIANT_PLAIN, *sums, n)
            v = np.where(v > 0, 0, v).astype(np.int32)
            lost = rng.random(n) < 0.4
            v[lost] = -4; x[lost] = -1; y[lost] = -1
            got = klt.select_features(ctx, klt.REPLACE_SOME, gv, *sums, n, x, y, v, grid=grid, capacity=cap)
            ref = ok.select_features(op, ok.REPLACE_SOME, ov, *sums, n, x, y, v, grid=grid, capacity=cap)
        else:
            got = klt.select_features(ctx, klt.SELECT_ALL, gv, *sums, n, grid=grid, capacity=cap)
            ref = ok.select_features(op, ok.SELECT_ALL, ov, *sums, n, grid=grid, capacity=cap)
    except klt.GkltError as e:
        print("skip it %d (%s): %s" % (it, variant, str(e)[:80]), flush=True)
        continue
    same = all(len(a) == len(b) and (a == b).all() for a, b in zip(got, ref))
    if not same:
        bad += 1
        print("MISMATCH it %d: %dx%d %s replace %d n %d mindist %d grid %s impl %d iters %d seed %d kw %s" % (
            it, W, H, variant, replace, n, mindist, grid, impl, iters, seed, kw), flush=True)
    elif it % 20 == 0:
        print("ok %d: %dx%d %s replace %d n %d -> %d found" % (it, W, H, variant, replace, n, int((got[2] >= 0).sum())), flush=True)
ctx.set_select_impl(0)
ctx.close()
print("done: %d cases, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
