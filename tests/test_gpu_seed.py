"""keyframe seeding on the device (gklt_seed_from_level) against the oracle's
restatement (oracle.seed_from_level: the reference's Shi-Tomasi selector on the integral images of
a pyramid level, seeds x scale, val = KLT_TRACKED -- System.cpp:407-420).  Exact lists.

Parity note: the reference seeds from ORB's FAST keypoints (cv::FAST inside ORBextractor.cc:807-900,
OpenCV, not in /root/reference): that detector is not restated, so parity of the SEED POSITIONS with
the shipped system is unpinned; what is pinned is the selector (S1-S4) the seeds come from here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(H, W, seed):
    from gslam_amd import synth

    T = synth.base_texture(H, W, seed=seed)
    return synth.render_frame(T, H, W, 0, seed=seed), synth.render_frame(T, H, W, 1, seed=seed)


@pytest.mark.parametrize("H,W,win,L,level,n,kw", [
    (270, 480, 7, 3, 2, 300, dict(min_eigenvalue=20, mindist=4)),     # coarsest level, as System does
    (270, 480, 7, 3, 1, 400, dict(min_eigenvalue=50, mindist=6)),
    (200, 320, 11, 2, 1, 5000, dict(min_eigenvalue=10, mindist=3)),   # more slots than corners: NOT_FOUND tail
    (200, 320, 7, 2, 0, 250, dict(min_eigenvalue=100, mindist=10)),   # level 0 = KLTSelectGoodFeatures, val -> 0
    (1080, 1920, 7, 4, 3, 2000, dict(min_eigenvalue=30, mindist=3)),  # C2 frame, 240 x 135 coarsest level
])
def test_seed_from_level_matches_oracle(klt, ok, H, W, win, L, level, n, kw):
    f0, _ = _frames(H, W, 21)
    gp = klt.make_params(window=win, levels=L, **kw)
    op = ok.make_params(window=win, levels=L, **kw)
    ctx = klt.Context(0, gp)
    py = klt.Pyramid(ctx, W, H)
    py.build(f0)
    fl = klt.FeatureList(ctx, n)
    klt.seed_from_level(ctx, py, level, fl, n)
    gx, gy, gv = fl.download()
    ok.set_threads(8)
    rx, ry, rv = ok.seed_from_level(op, ok.compute_pyramid(op, f0), level, n)
    ok.set_threads(1)
    assert len(gx) == n
    assert (gv == rv).all() and (gx == rx).all() and (gy == ry).all()
    found = gv == 0
    assert found.sum() > 30 and set(np.unique(gv)) <= {0, -1}
    # every seed lies inside the level-0 border the tracker tests, on the level's pixel lattice
    s = 2 ** level
    assert (gx[found] >= gp.borderx).all() and (gx[found] <= W - 1 - gp.borderx).all()
    assert (gy[found] >= gp.bordery).all() and (gy[found] <= H - 1 - gp.bordery).all()
    assert (gx[found] % s == 0).all() and (gy[found] % s == 0).all()
    fl.close(); py.close(); ctx.close()


def test_seed_then_track_without_host_round_trip(klt, ok):
    """Keyframe at frame t: seed from the coarsest level of pyramid(t), track t -> t+1, all on the
    device (the list never visits the host in between); equals oracle seeding + oracle tracking."""
    H, W, win, L, n = 270, 480, 7, 3, 500
    f0, f1 = _frames(H, W, 4)
    kw = dict(min_eigenvalue=20, mindist=4)
    gp = klt.make_params(window=win, levels=L, **kw)
    op = ok.make_params(window=win, levels=L, **kw)
    ctx = klt.Context(0, gp)
    p0, p1 = klt.Pyramid(ctx, W, H), klt.Pyramid(ctx, W, H)
    p0.build(f0)
    p1.build(f1)
    fl = klt.FeatureList(ctx, n)
    klt.seed_from_level(ctx, p0, L - 1, fl, n)
    klt.track_flist(ctx, p0, p1, fl)
    gx, gy, gv = fl.download()
    r0, r1 = ok.compute_pyramid(op, f0), ok.compute_pyramid(op, f1)
    sx, sy, sv = ok.seed_from_level(op, r0, L - 1, n)
    rx, ry, rv = ok.track_features(op, r0, r1, sx, sy, sv)[:3]
    assert (gv == rv).all() and (gx == rx).all() and (gy == ry).all()
    assert (gv == 0).sum() > 0.7 * (sv == 0).sum()
    # the selector scratch sized for the coarse level is reused by a level-0 selection and back
    klt.select_from_pyramid(ctx, p0, klt.SELECT_ALL, klt.VARIANT_PLAIN, fl)
    klt.seed_from_level(ctx, p0, L - 1, fl, n)
    gx2, gy2, gv2 = fl.download()
    assert (gx2 == sx).all() and (gy2 == sy).all() and (gv2 == sv).all()
    fl.close(); p0.close(); p1.close(); ctx.close()


def test_seed_rejects_bad_arguments(klt):
    gp = klt.make_params(window=7, levels=2)
    ctx = klt.Context(0, gp)
    py = klt.Pyramid(ctx, 160, 120)
    fl = klt.FeatureList(ctx, 10)
    with pytest.raises(klt.GkltError):
        klt.seed_from_level(ctx, py, 0, fl, 5)       # never built
    py.build(np.zeros((120, 160), np.uint8))
    with pytest.raises(klt.GkltError):
        klt.seed_from_level(ctx, py, 2, fl, 5)       # no such level
    with pytest.raises(klt.GkltError):
        klt.seed_from_level(ctx, py, 1, fl, 11)      # beyond the capacity
    klt.seed_from_level(ctx, py, 1, fl, 10)          # flat image: no corner anywhere
    assert (fl.download()[2] == -1).all()
    fl.close(); py.close(); ctx.close()
