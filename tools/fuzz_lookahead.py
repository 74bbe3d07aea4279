#!/usr/bin/env python3
"""Randomised parity sweep of gklt_lookahead (batched builds + multi-frame tracker launches)
against the oracle run frame by frame (GPU box, not part of the suite): random windows incl. the
row-run and staged mappings, levels, batch patterns, wavefront caps, host / device frames, pinned /
pageable outputs, re-seeds between batches, flushes in the middle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth
from oracle import oracle as ok

ok.build(); ok.set_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = klt.Context(0, klt.make_params(window=7, levels=2))
bad = 0
for it in range(N):
    ww, wh = [(7, 7), (7, 7), (11, 11), (21, 21), (15, 21), (19, 17), (5, 9), (23, 23), (13, 13)][int(rng.integers(0, 9))]
    levels = int(rng.integers(1, 4))
    H = int(rng.integers(150, 260)); W = int(rng.integers(200, 360))
    n = int(rng.integers(1, 300))
    B = int(rng.integers(1, 9))
    nb = int(rng.integers(1, 5))
    batches = [int(rng.integers(1, B + 1)) for _ in range(nb)]
    nfr = sum(batches) + 1
    seed = int(rng.integers(0, 1 << 30))
    T = synth.base_texture(H, W, seed=seed)
    f = [synth.render_frame(T, H, W, k, seed=seed, noise_sigma=1.5) for k in range(nfr)]
    kw = dict(window=max(ww, wh), levels=levels)
    gp, op = klt.make_params(**kw), ok.make_params(**kw)
    for q in (gp, op):
        q.window_width, q.window_height = ww, wh
    ctx.set_params(gp)
    waves = int(rng.choice([0, 0, 1, 13, 64]))
    ctx.set_track_waves(waves)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    n = len(x)
    val = np.where(rng.random(n) < 0.05, -2, 0).astype(np.int32)
    device_frames = rng.random() < 0.5
    pinned = rng.random() < 0.7
    flush_after = int(rng.integers(0, nb)) if rng.random() < 0.4 else -1
    reseed_batch = int(rng.integers(1, nb)) if nb > 1 and rng.random() < 0.4 else -1
    la = klt.Lookahead(ctx, W, H, n, B)
    dptr = [ctx.device_upload(a) for a in f] if device_frames else []
    outs = []
    for _ in range(nfr - 1):
        if pinned:
            outs.append(ctx.pinned_array((3, n), np.float32))
        else:
            a = np.zeros((3, n), np.float32); outs.append((a, a.ctypes.data))
    reseed_frames = set()
    try:
        la.upload_features(x, y, val)
        la.start((dptr[0], W) if device_frames else f[0])
        t = 1
        for i, b in enumerate(batches):
            if i == reseed_batch:
                # applies from the first frame not yet handed to a tracker launch
                reseed_frames.add(la.frames_tracked() + 1)
                la.upload_features(x, y, val)
            fr = [(dptr[k], W) for k in range(t, t + b)] if device_frames else f[t:t + b]
            la.submit(fr, [outs[k - 1][1] for k in range(t, t + b)])
            t += b
            if i == flush_after:
                la.flush()
        la.flush()
    finally:
        la.close()
        for p in dptr:
            ctx.device_free(p)
    rx, ry, rv = x, y, val
    prev = ok.compute_pyramid(op, f[0])
    same = True
    for t in range(1, nfr):
        if t in reseed_frames:
            rx, ry, rv = x, y, val
        cur = ok.compute_pyramid(op, f[t])
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        o = outs[t - 1][0]
        if not ((o[2].view(np.int32) == rv).all() and (o[0].view(np.uint32) == rx.view(np.uint32)).all() and (o[1].view(np.uint32) == ry.view(np.uint32)).all()):
            same = False
            break
        prev = cur
    if pinned:
        for _, p in outs:
            ctx.pinned_free(p)
    if not same:
        bad += 1
        print("MISMATCH it %d frame %d: window %dx%d levels %d %dx%d n %d batches %s B %d waves %d dev %d pinned %d flush %d reseed %d seed %d" % (
            it, t, ww, wh, levels, W, H, n, batches, B, waves, device_frames, pinned, flush_after, reseed_batch, seed), flush=True)
    elif it % 10 == 0:
        print("ok %d: window %dx%d levels %d batches %s waves %d" % (it, ww, wh, levels, batches, waves), flush=True)
ctx.set_track_waves(0)
ctx.close()
print("done: %d cases, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
