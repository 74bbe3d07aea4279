import sys; sys.path.insert(0, '.')
import numpy as np
from gslam_amd import klt, synth
from oracle import oracle as ok
ok.build(); ok.set_threads(8)
H, W = 1080, 1920
T = synth.base_texture(H, W); img = synth.render_frame(T, H, W, 0)
gp = klt.make_params(window=7, levels=4); op = ok.make_params(window=7, levels=4)
ctx = klt.Context(0, gp); ctx.set_pyramid_impl(4)
p = klt.Pyramid(ctx, W, H); p.build(img); got = p.planes(); p.close(); ctx.close()
ref = ok.compute_pyramid(op, img)
for i, (a, b) in enumerate(zip(got, ref)):
    print(i // 3, i % 3, "max abs err %.3g  magnitude %.3g  rel %.3g  differing %.1f%%" % (np.abs(a - b).max(), np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max(), 100 * (a.view(np.uint32) != b.view(np.uint32)).mean()))
