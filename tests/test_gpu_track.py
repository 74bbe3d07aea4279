"""GPU parity: KLTTrackFeatures through the C ABI vs the CPU oracle on identical pyramids.

The kernel keeps the reference's AVX lane assignment and summation order (one feature per 8-lane
slice) and is built -ffp-contract=off, so positions and status flags are compared EXACTLY
(north_star tolerance: 0.01 px, >= 99.9 % status agreement).
"""
import numpy as np
import pytest

from gslam_amd import synth

pytestmark = pytest.mark.gpu


def _stream(H, W, n, seed=21, noise=1.5):
    T = synth.base_texture(H, W, seed=seed)
    return [synth.render_frame(T, H, W, k, seed=seed, noise_sigma=noise) for k in range(n)]


def _run_pair(klt, ok, ctx, f0, f1, window, levels, x, y, val, border=None, invK=None, **kw):
    H, W = f0.shape
    gp = klt.make_params(window=window, levels=levels, border=border, **kw)
    op = ok.make_params(window=window, levels=levels, border=border, **kw)
    ctx.set_params(gp)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    try:
        p0.build(f0)
        p1.build(f1)
        got = klt.KLTTrackFeatures(ctx, p0, p1, x, y, val, invK)
        # the oracle tracks on the planes the GPU produced (pyramid parity is tested separately)
        ref = ok.track_features(op, p0.planes(), p1.planes(), x, y, val, invK)
    finally:
        p0.close()
        p1.close()
    return got, ref


def _assert_same(got, ref):
    gx, gy, gv = got[:3]
    rx, ry, rv = ref[:3]
    assert (gv == rv).all(), "status differs on %d features: gpu %s ref %s" % (
        (gv != rv).sum(), gv[gv != rv][:8], rv[gv != rv][:8])
    assert (gx.view(np.uint32) == rx.view(np.uint32)).all(), np.abs(gx - rx).max()
    assert (gy.view(np.uint32) == ry.view(np.uint32)).all(), np.abs(gy - ry).max()


@pytest.mark.parametrize("window,levels,H,W,n", [
    (7, 4, 270, 480, 1000),     # C2 shape, reduced frame
    (11, 3, 270, 480, 700),     # 11x11: LDS-staged footprint since round 3 (gathers: impl 2 below)
    (21, 2, 200, 320, 300),     # as-shipped window: LDS-staged footprint, a lane per window-row run
    (5, 3, 135, 240, 333),      # footprint in registers, n not a multiple of 8
    (9, 2, 135, 240, 64),       # smallest staged window (10 x 10 footprint texels)
    (13, 3, 200, 320, 260),     # staged, four window elements per lane
    (3, 2, 135, 240, 100),      # smallest window: footprint-in-registers sampling, 7 tail entries
    (31, 2, 200, 320, 40),      # 961 px: 16 window elements per lane
    (33, 1, 200, 320, 24),      # beyond the wave-per-feature kernel: slice kernel with LDS template
])
def test_track_exact(klt, ok, gpu_ctx, window, levels, H, W, n):
    f = _stream(H, W, 2)
    p = ok.make_params(window=window, levels=levels)
    x, y = synth.lattice_seeds(H, W, n, p.borderx)
    val = np.zeros(len(x), np.int32)
    got, ref = _run_pair(klt, ok, gpu_ctx, f[0], f[1], window, levels, x, y, val)
    _assert_same(got, ref)
    assert (got[2] == 0).mean() > 0.9  # the synthetic pan is well inside the capture range


@pytest.mark.parametrize("impl", [1, 2, 3])
@pytest.mark.parametrize("window,levels", [(7, 4), (11, 3), (21, 2), (5, 3)])
def test_track_alternative_kernels_agree(klt, ok, gpu_ctx, window, levels, impl):
    """The alternative mappings (1: one feature per 8-lane slice; 2: one feature per wavefront
    without the LDS footprint) give the same bits as the default kernel and the oracle."""
    H, W, n = 270, 480, 500
    f = _stream(H, W, 2)
    p = ok.make_params(window=window, levels=levels)
    x, y = synth.lattice_seeds(H, W, n, p.borderx)
    val = np.zeros(len(x), np.int32)
    gpu_ctx.set_track_impl(impl)
    try:
        got, ref = _run_pair(klt, ok, gpu_ctx, f[0], f[1], window, levels, x, y, val)
    finally:
        gpu_ctx.set_track_impl(0)
    _assert_same(got, ref)


_HALO_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, %(root)r)
from gslam_amd import klt, synth
from oracle import oracle as ok
ok.build()
H, W, n = 200, 320, 200
T = synth.base_texture(H, W, seed=5)
f = [synth.render_frame(T, H, W, k, seed=5) for k in range(2)]
ctx = klt.Context(0)
bad = 0
for ww in (15, 19, 21):
    gp, op = klt.make_params(window=ww, levels=2), ok.make_params(window=ww, levels=2)
    ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, op.borderx)
    val = np.zeros(n, np.int32)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    p0.build(f[0]); p1.build(f[1])
    got = klt.KLTTrackFeatures(ctx, p0, p1, x, y, val)
    ref = ok.track_features(op, p0.planes(), p1.planes(), x, y, val)
    bad += int((got[2] != ref[2]).sum() + (got[0] != ref[0]).sum() + (got[1] != ref[1]).sum())
    assert (ref[2] == 0).sum() > n // 2
    p0.close(); p1.close()
ctx.close()
print("MISMATCHES", bad)
"""


@pytest.mark.parametrize("halo", [2, 4])
def test_track_footprint_halo_knob(klt, halo):
    """GKLT_TRACK_HALO (read once per process, hence a child process) widens the LDS footprint of the
    staged windows.  With halo 4 a 19..21-px window needs 15 texels per lane, more than the row-run
    kernels are instantiated for: it must fall back to the element-per-lane mapping (ADVICE round 2:
    it read uninitialised LDS), and every setting must give the oracle's bits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GKLT_TRACK_HALO=str(halo))
    r = subprocess.run([sys.executable, "-c", _HALO_SCRIPT % {"root": root}], env=env, capture_output=True,
                       text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "MISMATCHES 0" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("ww,wh", [(15, 15), (17, 17), (19, 19), (21, 15), (15, 21), (19, 13), (17, 21), (23, 21), (21, 23)])
def test_track_row_run_mapping_windows(klt, ok, gpu_ctx, ww, wh):
    """Staged windows 15..21 px wide take the row-run mapping (a lane owns 7 consecutive elements of
    one window row): widths that are not 3 x 7 leave partial runs, heights below 21 leave idle lanes,
    23 px falls back to the element-per-lane mapping.  Features up to the border exercise the
    footprint fetch at the image edge.  Exact against the oracle, 2 levels."""
    H, W = 200, 320
    f = _stream(H, W, 2, seed=5)
    gp = klt.make_params(window=21, levels=2)
    op = ok.make_params(window=21, levels=2)
    for q in (gp, op):
        q.window_width, q.window_height = ww, wh
    gpu_ctx.set_params(gp)
    rng = np.random.default_rng(ww * 100 + wh)
    n = 257
    x = rng.uniform(0, W - 1, n).astype(np.float32)     # anywhere, incl. inside the border strip
    y = rng.uniform(0, H - 1, n).astype(np.float32)
    x[:8] = [ww // 2, ww // 2 + 0.5, W - 1 - ww // 2, W - 1.5 - ww // 2, 100, 100, 2 * (ww // 2), W - 2 * (ww // 2) - 2]
    y[:8] = [100, 100, 100, 100, wh // 2, H - 1 - wh // 2, 2 * (wh // 2), H - 2 * (wh // 2) - 2]
    val = np.zeros(n, np.int32)
    p0, p1 = klt.Pyramid(gpu_ctx, W, H), klt.Pyramid(gpu_ctx, W, H)
    try:
        p0.build(f[0]); p1.build(f[1])
        got = klt.KLTTrackFeatures(gpu_ctx, p0, p1, x, y, val)
        ref = ok.track_features(op, p0.planes(), p1.planes(), x, y, val)
    finally:
        p0.close(); p1.close()
    _assert_same(got, ref)
    assert (got[2] == 0).sum() > n // 3


def test_track_rectangular_window_and_even_fix(klt, ok, gpu_ctx):
    """Even / too-small windows are corrected (LK.hpp:982-1013): 6x2 becomes 7x3."""
    H, W = 135, 240
    f = _stream(H, W, 2)
    x, y = synth.lattice_seeds(H, W, 100, 20)
    val = np.zeros(len(x), np.int32)
    gp = klt.make_params(window=7, levels=2, border=20)
    gp.window_width, gp.window_height = 6, 2
    op = ok.make_params(window=7, levels=2, border=20)
    op.window_width, op.window_height = 7, 3
    gpu_ctx.set_params(gp)
    q = gpu_ctx.get_params()
    assert (q.window_width, q.window_height) == (7, 3)
    p0, p1 = klt.Pyramid(gpu_ctx, W, H), klt.Pyramid(gpu_ctx, W, H)
    try:
        p0.build(f[0]); p1.build(f[1])
        got = klt.KLTTrackFeatures(gpu_ctx, p0, p1, x, y, val)
        ref = ok.track_features(op, p0.planes(), p1.planes(), x, y, val)
    finally:
        p0.close(); p1.close()
    _assert_same(got, ref)


def test_track_status_codes(klt, ok, gpu_ctx):
    """Every status of KLT.h:69-74 the tracker can produce, on crafted inputs."""
    H, W = 200, 320
    T = synth.base_texture(H, W, seed=3)
    f0 = synth.render_frame(T, H, W, 0, seed=3, noise_sigma=0)
    f1 = synth.render_frame(T, H, W, 1, seed=3, noise_sigma=0)
    f0 = f0.copy(); f1 = f1.copy()
    f0[60:120, 40:110] = 90; f1[60:120, 40:110] = 90       # flat patch -> SMALL_DET
    rng = np.random.default_rng(9)
    f1[130:190, 200:300] = rng.integers(0, 255, (60, 100))  # unrelated content -> LARGE_RESIDUE / MAX_ITER
    x = np.array([80, 75.5, 250, 260.25, 3.0, 316.0, 160, 150.5, 100.0, 30.0], np.float32)
    y = np.array([90, 95.25, 160, 150.5, 100.0, 100.0, 2.0, 197.5, 50.0, 30.0], np.float32)
    val = np.array([0, 500, 0, 0, 0, 0, 0, 0, -3, 0], np.int32)   # one already-lost feature
    got, ref = _run_pair(klt, ok, gpu_ctx, f0, f1, 7, 2, x, y, val, border=8)
    _assert_same(got, ref)
    gv = got[2]
    assert gv[0] == klt.KLT_SMALL_DET and gv[1] == klt.KLT_SMALL_DET
    assert gv[4] == klt.KLT_OOB and gv[5] == klt.KLT_OOB and gv[6] == klt.KLT_OOB and gv[7] == klt.KLT_OOB
    assert gv[8] == -3 and got[0][8] == x[8]              # val < 0 is left untouched (LK.hpp:830)
    assert set(gv[2:4]) <= {klt.KLT_LARGE_RESIDUE, klt.KLT_MAX_ITERATIONS, klt.KLT_SMALL_DET, klt.KLT_OOB}
    assert gv[9] == klt.KLT_TRACKED
    lost = (gv < 0) & (np.arange(len(gv)) != 8)
    assert (got[0][lost] == -1.0).all() and (got[1][lost] == -1.0).all()  # LK.hpp:888-910


@pytest.mark.parametrize("impl", [0, 1, 2, 3])
def test_track_descent_break_keeps_coarse_scale(klt, ok, gpu_ctx, impl):
    """SMALL_DET at a coarse level r > 0 ends the descent (LK.hpp:879-880) and _outOfBounds
    (LK.hpp:890-899) then sees xlocout still at level-r scale: a flat patch near the left / top
    border comes back KLT_OOB, not KLT_SMALL_DET.  Every tracker mapping, 3 levels."""
    H, W = 200, 320
    T = synth.base_texture(H, W, seed=3)
    f0 = synth.render_frame(T, H, W, 0, seed=3, noise_sigma=0).copy()
    f1 = synth.render_frame(T, H, W, 1, seed=3, noise_sigma=0).copy()
    f0[0:180, 0:110] = 90; f1[0:180, 0:110] = 90       # flat down to level 2
    #            left border  top border  inside     textured
    x = np.array([20.0,        60.0,       60.0,      200.5], np.float32)
    y = np.array([100.0,       28.0,       100.0,     120.25], np.float32)
    val = np.zeros(4, np.int32)
    gpu_ctx.set_track_impl(impl)
    try:
        got, ref = _run_pair(klt, ok, gpu_ctx, f0, f1, 7, 3, x, y, val, border=8)
    finally:
        gpu_ctx.set_track_impl(0)
    _assert_same(got, ref)
    assert list(got[2]) == [klt.KLT_OOB, klt.KLT_OOB, klt.KLT_SMALL_DET, klt.KLT_TRACKED]


def test_track_integer_positions_take_copy_path(klt, ok, gpu_ctx):
    """weights[0]==1.0 -> memcpy rows (LK.hpp:80-85)."""
    H, W = 135, 240
    f = _stream(H, W, 2, noise=0)
    xs, ys = np.meshgrid(np.arange(40, 200, 9), np.arange(40, 100, 7))
    x = xs.ravel().astype(np.float32); y = ys.ravel().astype(np.float32)
    val = np.zeros(len(x), np.int32)
    got, ref = _run_pair(klt, ok, gpu_ctx, f[0], f[0], 7, 1, x, y, val, border=10)
    _assert_same(got, ref)
    assert (got[2] == 0).all() and (got[0] == x).all() and (got[1] == y).all()


def test_track_bearing_vectors(klt, ok, gpu_ctx):
    H, W = 135, 240
    f = _stream(H, W, 2)
    x, y = synth.lattice_seeds(H, W, 200, 20)
    val = np.zeros(len(x), np.int32)
    K = np.array([[210.0, 0, 119.5], [0, 205.0, 67.0], [0, 0, 1]])
    got, ref = _run_pair(klt, ok, gpu_ctx, f[0], f[1], 7, 3, x, y, val, border=20, invK=np.linalg.inv(K))
    _assert_same(got, ref)
    m = got[2] == 0
    assert m.sum() > 100
    np.testing.assert_allclose(got[3][m], ref[3][m], rtol=0, atol=1e-15)
    np.testing.assert_allclose(got[4][m], ref[4][m], rtol=1e-15, atol=1e-15)
    assert (got[3][~m] == 0).all()


def test_track_empty_and_all_lost(klt, gpu_ctx):
    H, W = 64, 96
    f = _stream(H, W, 1)
    gpu_ctx.set_params(klt.make_params(window=7, levels=2))
    p0 = klt.Pyramid(gpu_ctx, W, H)
    try:
        p0.build(f[0])
        x, y, v = klt.KLTTrackFeatures(gpu_ctx, p0, p0, [], [], [])
        assert len(x) == 0
        x, y, v = klt.KLTTrackFeatures(gpu_ctx, p0, p0, [5.0, 6.0], [7.0, 8.0], [-1, -4])
        assert list(v) == [-1, -4] and list(x) == [5.0, 6.0]
    finally:
        p0.close()


def test_track_sequence_device_list(klt, ok, gpu_ctx):
    """Chained frames on the device-resident list (pyramid t+1 built while t-1 -> t is tracked,
    System.cpp:501-518) equal the oracle run frame by frame; stats count TRACKED features."""
    H, W, nfr, n = 180, 320, 6, 400
    f = _stream(H, W, nfr)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    ring = [klt.Pyramid(gpu_ctx, W, H) for _ in range(3)]
    fl = klt.FeatureList(gpu_ctx, n)
    try:
        fl.upload(x, y, val)
        ring[0].build(f[0])
        ring[1].build(f[1])
        for t in range(1, nfr):
            if t + 1 < nfr:
                ring[(t + 1) % 3].build(f[t + 1])
            klt.track_flist(gpu_ctx, ring[(t - 1) % 3], ring[t % 3], fl)
        gx, gy, gv = fl.download()
        tracked_total, iters_total = fl.stats()
    finally:
        fl.close()
        for r in ring:
            r.close()
    rx, ry, rv = x, y, val
    total = 0
    prev = ok.compute_pyramid(op, f[0])
    for t in range(1, nfr):
        cur = ok.compute_pyramid(op, f[t])
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        total += int((rv == 0).sum())
        prev = cur
    _assert_same((gx, gy, gv), (rx, ry, rv))
    assert tracked_total == total and iters_total > 0


def test_full_size_c2_frame_pair(klt, ok, gpu_ctx):
    """BASELINE C2 at full size (1080p, 4000 features, 4 levels, 7x7), one frame pair."""
    H, W, n = 1080, 1920, 4000
    T = synth.base_texture(H, W)
    f0, f1 = synth.render_frame(T, H, W, 0), synth.render_frame(T, H, W, 1)
    p = ok.make_params(window=7, levels=4)
    assert p.borderx == 64
    x, y = synth.lattice_seeds(H, W, n, p.borderx)
    val = np.zeros(n, np.int32)
    ok.set_threads(8)
    try:
        got, ref = _run_pair(klt, ok, gpu_ctx, f0, f1, 7, 4, x, y, val)
    finally:
        ok.set_threads(1)
    _assert_same(got, ref)
    assert (got[2] == 0).mean() > 0.95


def test_full_size_c3_frame_pair(klt, ok, gpu_ctx):
    """BASELINE C3 at full size: 3840x2160, 16000 features, 5 levels, 11x11, border 160."""
    H, W, n = 2160, 3840, 16000
    T = synth.base_texture(H, W)
    f0, f1 = synth.render_frame(T, H, W, 0), synth.render_frame(T, H, W, 1)
    p = ok.make_params(window=11, levels=5)
    assert p.borderx == 160
    x, y = synth.lattice_seeds(H, W, n, p.borderx, nx=160, ny=100)
    val = np.zeros(n, np.int32)
    ok.set_threads(8)
    try:
        got, ref = _run_pair(klt, ok, gpu_ctx, f0, f1, 11, 5, x, y, val)
    finally:
        ok.set_threads(1)
    _assert_same(got, ref)
    assert (got[2] == 0).mean() > 0.95


def test_full_size_c3_pyramid(klt, ok, gpu_ctx):
    """The 5-level pyramid of a 4K frame (sigma0 = 1.1 for the 11x11 window), every plane exact."""
    H, W = 2160, 3840
    T = synth.base_texture(H, W, seed=77)
    f0 = synth.render_frame(T, H, W, 3, seed=77)
    gpu_ctx.set_params(klt.make_params(window=11, levels=5))
    op = ok.make_params(window=11, levels=5)
    ok.set_threads(8)
    p0 = klt.Pyramid(gpu_ctx, W, H)
    try:
        p0.build(f0)
        ref = ok.compute_pyramid(op, f0)
        for i, r in enumerate(ref):
            g = p0.plane(i // 3, i % 3)
            assert g.shape == r.shape
            assert (g.view(np.uint32) == np.asarray(r, np.float32).view(np.uint32)).all(), "plane %d" % i
    finally:
        ok.set_threads(1)
        p0.close()


def test_full_size_cship_frame_pair(klt, ok, gpu_ctx):
    """The as-shipped configuration (Settings.h:65 DEBUG, System.cpp:326-344: 21x21, 2 levels,
    border 50; list of 2000, System.cpp:117) at 1080p -- the substitute for BASELINE config 1,
    whose 'robot' sequence is not in the reference repository.  LDS-staged footprint kernel."""
    H, W, n = 1080, 1920, 2000
    T = synth.base_texture(H, W, seed=11)
    f0, f1 = synth.render_frame(T, H, W, 0, seed=11), synth.render_frame(T, H, W, 1, seed=11)
    x, y = synth.lattice_seeds(H, W, n, 50, nx=60, ny=34)
    val = np.zeros(n, np.int32)
    ok.set_threads(8)
    try:
        got, ref = _run_pair(klt, ok, gpu_ctx, f0, f1, 21, 2, x, y, val, border=50)
    finally:
        ok.set_threads(1)
    _assert_same(got, ref)
    assert (got[2] == 0).mean() > 0.95


@pytest.mark.parametrize("pinned_out", [True, False])
def test_pipeline_matches_oracle(klt, ok, gpu_ctx, pinned_out):
    """gklt_pipeline (ring rotation, build(t+1) || track(t-1->t) on alternating streams, list
    delivered into host memory) equals the oracle run frame by frame, including a mid-sequence
    re-seed and the final track-only step."""
    H, W, nfr, n = 180, 320, 9, 320
    f = _stream(H, W, nfr)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    pipe = klt.Pipeline(gpu_ctx, W, H, n)
    per_frame = []
    if pinned_out:
        out, out_ptr = gpu_ctx.pinned_array((3, n), np.float32)
    else:
        out = np.zeros((3, n), np.float32)
        out_ptr = out.ctypes.data
    try:
        pipe.upload_features(x, y, val)
        pipe.start(f[0], f[1])
        for t in range(1, nfr):
            if t == 5:
                pipe.upload_features(x, y, val)
            pipe.step(f[t + 1] if t + 1 < nfr else None, out_ptr)
            per_frame.append(pipe.download())
            # the host copy written by the step is the same list
            assert (out[0] == per_frame[-1][0]).all() and (out[1] == per_frame[-1][1]).all()
            assert (out[2].view(np.int32) == per_frame[-1][2]).all()
        cur = pipe.pyramid(1).planes()
        tracked_total, _ = pipe.flist.stats()
    finally:
        pipe.close()
        if pinned_out:
            gpu_ctx.pinned_free(out_ptr)
    rx, ry, rv = x, y, val
    total = 0
    prev = ok.compute_pyramid(op, f[0])
    for t in range(1, nfr):
        if t == 5:
            rx, ry, rv = x, y, val
        curp = ok.compute_pyramid(op, f[t])
        rx, ry, rv = ok.track_features(op, prev, curp, rx, ry, rv)[:3]
        total += int((rv == 0).sum())
        _assert_same(per_frame[t - 1], (rx, ry, rv))
        prev = curp
    assert tracked_total == total
    for a, b in zip(cur, prev):   # the "current" pyramid after the last step is frame nfr-1
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("device_frames", [True, False])
def test_pipeline_run_is_a_loop_of_steps(klt, ok, gpu_ctx, device_frames):
    """gklt_pipeline_run(frames[0..n)) = n gklt_pipeline_step calls issued from C: the list after
    every frame (one pinned host buffer per frame) equals the oracle's, for device-resident and for
    page-locked host frames."""
    H, W, nfr, n = 180, 320, 10, 300
    f = _stream(H, W, nfr, seed=31)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    if device_frames:
        ptrs = [gpu_ctx.device_upload(a) for a in f]
    else:
        host = [gpu_ctx.pinned_array((H, W), np.uint8) for _ in f]
        for (a, _), src in zip(host, f):
            a[:] = src
        ptrs = [p for _, p in host]
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(nfr - 2)]
    pipe = klt.Pipeline(gpu_ctx, W, H, n)
    try:
        pipe.upload_features(x, y, val)
        dev = 1 if device_frames else 0
        pipe.start((ptrs[0], W, dev), (ptrs[1], W, dev))
        # steps t = 1 .. nfr-2: step t preloads frame t+1 and tracks t-1 -> t
        pipe.run([(ptrs[t + 1], W, dev) for t in range(1, nfr - 1)], [o[1] for o in outs])
        pipe.sync()
        rx, ry, rv = x, y, val
        prev = ok.compute_pyramid(op, f[0])
        for t in range(1, nfr - 1):
            cur = ok.compute_pyramid(op, f[t])
            rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
            o = outs[t - 1][0]
            _assert_same((o[0], o[1], o[2].view(np.int32)), (rx, ry, rv))
            prev = cur
    finally:
        pipe.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)
        for p in ptrs:
            (gpu_ctx.device_free if device_frames else gpu_ctx.pinned_free)(p)


def test_build_and_track_from_two_host_threads(klt, ok, gpu_ctx):
    """System.cpp:508-518: computePyramid(t+1) runs on a std::thread while KLTTrackFeatures works on
    the other two ring buffers.  Same pattern through the C ABI (ctypes releases the GIL): one
    thread builds, the caller tracks, 3-deep ring; results equal the sequential oracle."""
    import threading
    H, W, nfr, n = 180, 320, 8, 300
    f = _stream(H, W, nfr, seed=33)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    ring = [klt.Pyramid(gpu_ctx, W, H) for _ in range(3)]
    got, errors = [], []

    def preload(t):
        try:
            ring[t % 3].build(f[t])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    try:
        ring[0].build(f[0])
        ring[1].build(f[1])
        cx, cy, cv = x, y, val
        for t in range(1, nfr):
            th = None
            if t + 1 < nfr:
                th = threading.Thread(target=preload, args=(t + 1,))
                th.start()
            cx, cy, cv = klt.KLTTrackFeatures(gpu_ctx, ring[(t - 1) % 3], ring[t % 3], cx, cy, cv)
            got.append((cx, cy, cv))
            if th:
                th.join()
        assert not errors
    finally:
        for p in ring:
            p.close()
    rx, ry, rv = x, y, val
    prev = ok.compute_pyramid(op, f[0])
    for t in range(1, nfr):
        cur = ok.compute_pyramid(op, f[t])
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        _assert_same(got[t - 1], (rx, ry, rv))
        prev = cur


def test_pipeline_wait_tracked_delivers_the_list(klt, ok, gpu_ctx):
    """gklt_pipeline_wait_tracked: after it returns the frame's list is complete in pinned host
    memory although the next frame's build may still be running."""
    H, W, nfr, n = 180, 320, 6, 256
    f = _stream(H, W, nfr, seed=41)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    pipe = klt.Pipeline(gpu_ctx, W, H, n)
    out, out_ptr = gpu_ctx.pinned_array((3, n), np.float32)
    got = []
    try:
        pipe.upload_features(x, y, val)
        pipe.start(f[0], f[1])
        for t in range(1, nfr):
            pipe.step(f[t + 1] if t + 1 < nfr else None, out_ptr)
            pipe.wait_tracked()
            got.append((out[0].copy(), out[1].copy(), out[2].view(np.int32).copy()))
        pipe.sync()
    finally:
        pipe.close()
        gpu_ctx.pinned_free(out_ptr)
    rx, ry, rv = x, y, val
    prev = ok.compute_pyramid(op, f[0])
    for t in range(1, nfr):
        cur = ok.compute_pyramid(op, f[t])
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        _assert_same(got[t - 1], (rx, ry, rv))
        prev = cur
