#!/usr/bin/env python3
"""Randomised parity sweep of computePyramid through the C ABI against the oracle (run on a GPU
box, not part of the suite): random image sizes (ragged, smaller than a tile, odd halvings),
windows 3..29 (level-0 radii 1..6 and beyond), 1..5 levels, subsampling 2/4, strided host frames,
every pyramid implementation (tile, separate kernels, streaming).  Planes compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth
from oracle import oracle as ok

ok.build(); ok.set_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = klt.Context(0, klt.make_params(window=7, levels=2))
bad = 0
for it in range(N):
    window = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 33]))
    ss = 4 if rng.random() < 0.1 else 2
    levels = int(rng.integers(1, 6 if ss == 2 else 3))
    small = rng.random() < 0.3
    H = int(rng.integers(9, 70)) if small else int(rng.integers(70, 420))
    W = int(rng.integers(11, 90)) if small else int(rng.integers(90, 640))
    if rng.random() < 0.3:   # exact halvings
        H = (H >> levels << levels) or (1 << levels); W = (W >> levels << levels) or (1 << levels)
    while levels > 1 and min(H, W) // (ss ** (levels - 1)) < 2:
        levels -= 1
    impl = int(rng.choice([0, 0, 1, 3]))
    seed = int(rng.integers(0, 1 << 30))
    img = rng.integers(0, 256, (H, W), dtype=np.uint8) if rng.random() < 0.3 else None
    if img is None:
        T = synth.base_texture(H, W, seed=seed, margin=64)
        T = np.pad(T, ((0, 200), (0, 200)), mode="wrap")
        img = synth.render_frame(T, H, W, 0, seed=seed)
    if rng.random() < 0.3:   # a strided view
        big = np.zeros((H, W + int(rng.integers(1, 9))), np.uint8); big[:, :W] = img; img = big[:, :W]
    kw = dict(window=window, levels=levels, subsampling=ss) if ss != 2 else dict(window=window, levels=levels)
    gp = klt.make_params(**kw); op = ok.make_params(**kw)
    ctx.set_params(gp)
    ctx.set_pyramid_impl(impl)
    pyr = klt.Pyramid(ctx, W, H)
    try:
        pyr.build(img)
        got = pyr.planes()
    finally:
        pyr.close()
    ref = ok.compute_pyramid(op, np.ascontiguousarray(img))
    same = len(got) == len(ref) and all(
        a.shape == b.shape and (np.ascontiguousarray(a, np.float32).view(np.uint32) == np.ascontiguousarray(b, np.float32).view(np.uint32)).all()
        for a, b in zip(got, ref))
    if not same:
        bad += 1
        print("MISMATCH it %d: %dx%d window %d levels %d ss %d impl %d seed %d" % (it, W, H, window, levels, ss, impl, seed), flush=True)
    elif it % 25 == 0:
        print("ok %d: %dx%d window %d levels %d ss %d impl %d" % (it, W, H, window, levels, ss, impl), flush=True)
ctx.set_pyramid_impl(0)
ctx.close()
print("done: %d cases, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
