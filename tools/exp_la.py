#!/usr/bin/env python3
"""Experiment: lookahead batches with build and track NOT overlapped (submit; sync; flush), so the
event timings are those of the isolated launches.  usage: exp_la.py [C2|C3|Cship] [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gslam_amd import klt, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(synth.CONFIGS[name]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
Bs = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16]
gp = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 33)
ptrs = [ctx.device_upload(f) for f in frames]
sx, sy = synth.lattice_seeds(H, W, n, gp.borderx)
seeds = (sx, sy, np.zeros(n, np.int32))
for B in Bs:
    la = klt.Lookahead(ctx, W, H, n, B)
    la.upload_features(*seeds)
    la.start((ptrs[0], W))
    t = 1
    for rep in range(3):
        if rep == 1:
            ctx.profile_enable(True); ctx.profile_reset()
            la.flist.stats(reset=True)
        la.upload_features(*seeds)
        for k in range(32 // B):
            la.submit([(ptrs[(t + i) % 33], W) for i in range(B)])
            la.sync()
            la.flush()
            t += B
    prof = ctx.profile_get(); ctx.profile_enable(False)
    tr, it = la.flist.stats()
    parts = []
    for nm, (ms, cnt) in prof.items():
        if cnt:
            parts.append("%s %.1f us/launch (%.2f us/frame)" % (nm, ms / cnt * 1e3, ms / 64 * 1e3))
    print("%s B=%d: %s  tracked %.3f iters %.2f" % (name, B, "; ".join(parts), tr / (64.0 * n), it / (64.0 * n)))
    la.close()
ctx.close()
