"""frame ingest on the device (gklt_frame_*) against oracle.ingest_nv12 /
oracle.feature_colors (numpy restatements of main.cpp:77-107 and System.cpp:310-315).  Integer
work: bit-exact.  Parity with OpenCV's own resize / cvtColor is unpinned (not vendored); the
restatement follows its documented 8-bit fixed-point path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nv12(H, W, seed):
    from gslam_amd import synth

    T = synth.base_texture(H, W, seed=seed)
    y = synth.render_frame(T, H, W, 0, seed=seed)
    rng = np.random.default_rng(seed)
    uv = rng.integers(0, 256, (H // 2, W), dtype=np.uint8)   # full chroma range: saturation both ways
    return y, uv


@pytest.mark.parametrize("H,W", [(1080, 1920), (270, 480), (54, 70), (2, 2)])
def test_nv12_colour_image_matches_oracle(klt, ok, H, W):
    y, uv = _nv12(H, W, 31)
    ctx = klt.Context(0, klt.make_params(window=7, levels=1))
    f = klt.Frame(ctx, W, H)
    f.ingest_nv12(y, uv)
    _, want = ok.ingest_nv12(y, uv)
    assert (f.color() == want).all()
    # extremes of the fixed-point formula
    y2 = np.where(np.arange(W)[None, :] % 3 == 0, 255, 0).astype(np.uint8) * np.ones((H, 1), np.uint8)
    uv2 = np.where(np.arange(W)[None, :] % 4 < 2, 255, 0).astype(np.uint8) * np.ones((H // 2, 1), np.uint8)
    f.ingest_nv12(y2, uv2)
    assert (f.color() == ok.ingest_nv12(y2, uv2)[1]).all()
    f.close(); ctx.close()


def test_unaligned_device_planes_take_the_general_kernel(klt, ok):
    H, W = 120, 168   # W/2 = 84 is a multiple of 4, but the planes start at odd addresses
    y, uv = _nv12(H, W, 5)
    ctx = klt.Context(0, klt.make_params(window=7, levels=1))
    pad = np.zeros(3, np.uint8)
    dy = ctx.device_upload(np.concatenate([pad, y.ravel()]))
    duv = ctx.device_upload(np.concatenate([pad, uv.ravel()]))
    f = klt.Frame(ctx, W, H)
    f.ingest_nv12((dy + 3, W), (duv + 3, W))
    assert (f.color() == ok.ingest_nv12(y, uv)[1]).all()
    assert f.luma() == (dy + 3, W)          # device planes are borrowed, not copied
    f.close(); ctx.device_free(dy); ctx.device_free(duv); ctx.close()


def test_ingested_luma_feeds_the_pyramid_and_the_pipeline(klt, ok):
    """Two alternating gklt_frames through gklt_pipeline_step_frame == host frames through
    gklt_pipeline_step == oracle, list by list."""
    from gslam_amd import synth

    H, W, n, win, L = 270, 480, 400, 7, 3
    frames = synth.make_stream(H, W, 6, seed=8)
    rng = np.random.default_rng(1)
    uvs = [rng.integers(0, 256, (H // 2, W), dtype=np.uint8) for _ in frames]
    gp = klt.make_params(window=win, levels=L)
    op = ok.make_params(window=win, levels=L)
    ctx = klt.Context(0, gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)

    # standalone: gklt_pyramid_build_frame
    fr = [klt.Frame(ctx, W, H) for _ in range(2)]
    fr[0].ingest_nv12(frames[0], uvs[0])
    py = klt.Pyramid(ctx, W, H)
    klt.pyramid_build_frame(ctx, py, fr[0])
    ref0 = ok.compute_pyramid(op, frames[0])
    for a, b in zip(py.planes(), ref0):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    py.close()

    pipe = klt.Pipeline(ctx, W, H, n)
    pipe.upload_features(x, y, val)
    pipe.start(frames[0], frames[1])
    prev, cur = ref0, ok.compute_pyramid(op, frames[1])
    rx, ry, rv = x, y, val
    for t in range(1, 5):
        k = t % 2
        fr[k].ingest_nv12(frames[t + 1], uvs[t + 1])
        klt.pipeline_step_frame(pipe, fr[k])
        gx, gy, gv = pipe.download()
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        assert (gv == rv).all() and (gx == rx).all() and (gy == ry).all(), t
        prev, cur = cur, ok.compute_pyramid(op, frames[t + 1])
        # colours of the live features from the frame just ingested (System.cpp:310-315)
        got = fr[k].feature_colors(pipe.flist)
        want = ok.feature_colors(ok.ingest_nv12(frames[t + 1], uvs[t + 1])[1], gx, gy, gv)
        assert (got == want).all()
    assert (rv == 0).sum() > 0.8 * n
    pipe.close()
    for f in fr:
        f.close()
    ctx.close()


def test_gray_ingest_and_errors(klt, ok):
    H, W = 96, 128
    y, uv = _nv12(H, W, 2)
    gp = klt.make_params(window=7, levels=2)
    ctx = klt.Context(0, gp)
    with pytest.raises(klt.GkltError):
        klt.Frame(ctx, 127, 96)                     # odd width
    f = klt.Frame(ctx, W, H)
    py = klt.Pyramid(ctx, W, H)
    with pytest.raises(klt.GkltError):
        klt.pyramid_build_frame(ctx, py, f)         # nothing ingested
    f.ingest_gray(y)
    with pytest.raises(klt.GkltError):
        f.color()                                    # no colour image without chroma
    klt.pyramid_build_frame(ctx, py, f)
    for a, b in zip(py.planes(), ok.compute_pyramid(ok.make_params(window=7, levels=2), y)):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    py.close(); f.close(); ctx.close()
