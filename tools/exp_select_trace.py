import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from gslam_amd import klt, synth
cfg = dict(synth.CONFIGS["C2"]); H, W, n = cfg["H"], cfg["W"], cfg["n"]
gp = klt.make_params(window=7, levels=4, min_eigenvalue=100, mindist=10)
gp.bordery = max(gp.bordery, 24)
ctx = klt.Context(0, gp)
frames = synth.make_stream(H, W, 2)
pyr = klt.Pyramid(ctx, W, H); pyr.build(frames[0])
fl = klt.FeatureList(ctx, 100 * n)
fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
klt.select_from_pyramid(ctx, pyr, klt.SELECT_ALL, klt.VARIANT_UNIFORM, fl, grid=(32, 32))
ctx.sync()
