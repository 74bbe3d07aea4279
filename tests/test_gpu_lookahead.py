"""GPU parity: gklt_lookahead (batched pyramid builds + one multi-frame tracker launch per batch)
vs the CPU oracle run frame by frame.  Lists are compared EXACTLY after every frame."""
import numpy as np
import pytest

from gslam_amd import synth

pytestmark = pytest.mark.gpu


def _stream(H, W, n, seed=21, noise=1.5):
    T = synth.base_texture(H, W, seed=seed)
    return [synth.render_frame(T, H, W, k, seed=seed, noise_sigma=noise) for k in range(n)]


def _oracle_lists(ok, op, frames, x, y, val, reseed_before=()):
    """list after tracking (t-1 -> t) for t = 1..len-1; reseed_before: frames t whose step starts
    from the seeds again."""
    out = []
    rx, ry, rv = x, y, val
    prev = ok.compute_pyramid(op, frames[0])
    for t in range(1, len(frames)):
        if t in reseed_before:
            rx, ry, rv = x, y, val
        cur = ok.compute_pyramid(op, frames[t])
        rx, ry, rv = ok.track_features(op, prev, cur, rx, ry, rv)[:3]
        out.append((rx.copy(), ry.copy(), rv.copy()))
        prev = cur
    return out


def _same(out, ref, what):
    rx, ry, rv = ref
    assert (out[2].view(np.int32) == rv).all(), what
    assert (out[0].view(np.uint32) == rx.view(np.uint32)).all(), what
    assert (out[1].view(np.uint32) == ry.view(np.uint32)).all(), what


@pytest.mark.parametrize("batches,device_frames,pinned,split", [
    ([1, 1, 1, 1, 1, 1, 1, 1], False, True, 0),   # degenerate: frame at a time
    ([3, 3, 2], False, True, 0),                  # ragged last batch
    ([8], True, True, 0),                         # one full batch, frames resident in HBM
    ([4, 4], True, False, 0),                     # pageable host_out: copy path
    ([2, 5, 1], False, True, 0),
    # coarse levels on the second build stream (gklt_ctx_set_coarse_split): more batches than the
    # ring holds, one-frame batches (built on one stream) between split ones, host and device frames
    ([3, 3, 3, 3, 3, 3, 2], True, True, 1),
    ([2, 1, 4, 1, 1, 4, 3], False, True, 1),
    ([4, 4, 4, 4], True, False, 1),
])
def test_lookahead_matches_oracle(klt, ok, gpu_ctx, batches, device_frames, pinned, split):
    H, W, n = 180, 320, 320
    nfr = sum(batches) + 1
    f = _stream(H, W, nfr)
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    x[5] = 3.0          # a feature that is out of bounds from the start
    val = np.zeros(n, np.int32)
    val[7] = -1         # lost features stay untouched
    ref = _oracle_lists(ok, op, f, x, y, val)

    gpu_ctx.set_coarse_split(split)
    la = klt.Lookahead(gpu_ctx, W, H, n, max(batches))
    dptr = [gpu_ctx.device_upload(a) for a in f] if device_frames else []
    outs = []
    for _ in range(nfr - 1):
        if pinned:
            outs.append(gpu_ctx.pinned_array((3, n), np.float32))
        else:
            a = np.zeros((3, n), np.float32)
            outs.append((a, a.ctypes.data))
    try:
        la.upload_features(x, y, val)
        la.start((dptr[0], W) if device_frames else f[0])
        t = 1
        for b in batches:
            fr = [(dptr[k], W) for k in range(t, t + b)] if device_frames else f[t:t + b]
            la.submit(fr, [outs[k - 1][1] for k in range(t, t + b)])
            assert la.frames_tracked() == t - 1 or b == 0
            t += b
        la.flush()
        assert la.frames_tracked() == nfr - 1
        final = la.flist.download()
        tracked_total, _ = la.flist.stats()
    finally:
        la.close()
        gpu_ctx.set_coarse_split(0)
        for p in dptr:
            gpu_ctx.device_free(p)
    try:
        for k in range(nfr - 1):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
        _same(np.stack([final[0], final[1], final[2].view(np.float32)]), ref[-1], "final list")
        assert tracked_total == sum(int((r[2] == 0).sum()) for r in ref)
        assert ref[-1][2][7] == -1 and ref[0][2][5] < 0
    finally:
        if pinned:
            for _, p in outs:
                gpu_ctx.pinned_free(p)


@pytest.mark.parametrize("pyr_impl", [1, 2, 3])
def test_lookahead_split_build_other_pyramid_kernels(klt, ok, gpu_ctx, pyr_impl):
    """The second build stream with the unfused kernels (the whole build stays on the first stream
    and the second one waits for it) and with the other fused variants: same lists."""
    gpu_ctx.set_pyramid_impl(pyr_impl)
    try:
        test_lookahead_matches_oracle(klt, ok, gpu_ctx, [3, 4, 1, 3], True, True, 1)
    finally:
        gpu_ctx.set_pyramid_impl(0)


def test_lookahead_reseed_between_batches(klt, ok, gpu_ctx):
    """upload_features() applies from the first frame that is submitted but not yet tracked."""
    H, W, n, B = 135, 240, 200, 3
    f = _stream(H, W, 3 * B + 1, seed=5)
    gp = klt.make_params(window=11, levels=2)
    op = ok.make_params(window=11, levels=2)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    # batches: frames 1-3, 4-6, 7-9; the re-seed is uploaded after submit(batch 1) has been
    # called, i.e. it applies to the tracking of batch 1?  No: batch 1 is still pending then, so
    # upload before submit(2) re-seeds at frame 1; upload after submit(2) re-seeds at frame 4.
    ref = _oracle_lists(ok, op, f, x, y, val, reseed_before={4})
    la = klt.Lookahead(gpu_ctx, W, H, n, B)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(3 * B)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:4], [o[1] for o in outs[0:3]])
        la.submit(f[4:7], [o[1] for o in outs[3:6]])   # tracks frames 1-3
        la.upload_features(x, y, val)                  # next tracker launch: frames 4-6
        la.submit(f[7:10], [o[1] for o in outs[6:9]])
        la.flush()
        for k in range(3 * B):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
    finally:
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)


@pytest.mark.parametrize("impl", [1, 3])
def test_lookahead_other_tracker_kernels(klt, ok, gpu_ctx, impl):
    """Slice kernel (one launch per frame pair) and the staged wave kernel behind the same API."""
    H, W, n = 135, 240, 150
    f = _stream(H, W, 6, seed=9)
    gp = klt.make_params(window=7, levels=2)
    op = ok.make_params(window=7, levels=2)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    ref = _oracle_lists(ok, op, f, x, y, val)
    gpu_ctx.set_track_impl(impl)
    la = klt.Lookahead(gpu_ctx, W, H, n, 5)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(5)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:6], [o[1] for o in outs])
        la.flush()
        for k in range(5):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
    finally:
        gpu_ctx.set_track_impl(0)
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)


def test_lookahead_argument_errors(klt, gpu_ctx):
    gpu_ctx.set_params(klt.make_params(window=7, levels=2))
    with pytest.raises(klt.GkltError):
        klt.Lookahead(gpu_ctx, 64, 64, 10, 17)      # > GKLT_MAX_LOOKAHEAD
    la = klt.Lookahead(gpu_ctx, 64, 64, 10, 2)
    try:
        img = np.zeros((64, 64), np.uint8)
        with pytest.raises(klt.GkltError):
            la.submit([img])                        # not started
        la.start(img)
        with pytest.raises(klt.GkltError):
            la.submit([img, img, img])              # more than max_batch
        la.submit([img, img])
        la.flush()
    finally:
        la.close()


@pytest.mark.parametrize("waves", [1, 7, 64])
def test_lookahead_wavefronts_take_several_features(klt, ok, gpu_ctx, waves):
    """A multi-frame launch is capped at 4096 wavefronts; each takes features w, w + waves, ...
    Forced here to a few wavefronts so that every one follows several features through the batch."""
    H, W, n = 135, 240, 150
    f = _stream(H, W, 5, seed=13)
    gp = klt.make_params(window=7, levels=2)
    op = ok.make_params(window=7, levels=2)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    x[3] = 2.0                      # out of bounds from the start
    val = np.zeros(n, np.int32)
    val[11] = -4                    # lost before the batch
    ref = _oracle_lists(ok, op, f, x, y, val)
    gpu_ctx.set_track_waves(waves)
    la = klt.Lookahead(gpu_ctx, W, H, n, 4)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(4)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:5], [o[1] for o in outs])
        la.flush()
        for k in range(4):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
        tracked_total, _ = la.flist.stats()
        assert tracked_total == sum(int((r[2] == 0).sum()) for r in ref)
    finally:
        gpu_ctx.set_track_waves(0)
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)


@pytest.mark.parametrize("waves", [0, 37])
def test_lookahead_11x11_capped_launch(klt, ok, gpu_ctx, waves):
    """The multi-frame 11x11 launch of BASELINE C3 (two window elements per lane, 16-element
    chains) with fewer wavefronts than features, so each follows several features through the
    batch; 5 levels as C3."""
    H, W, n, B = 480, 768, 900, 6     # coarsest level 30 x 48: border 96 keeps windows inside it
    f = _stream(H, W, B + 1, seed=17)
    gp = klt.make_params(window=11, levels=5, border=96)
    op = ok.make_params(window=11, levels=5, border=96)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    val[13] = -2
    ref = _oracle_lists(ok, op, f, x, y, val)
    gpu_ctx.set_track_waves(waves)
    la = klt.Lookahead(gpu_ctx, W, H, n, B)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(B)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:B + 1], [o[1] for o in outs])
        la.flush()
        for k in range(B):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
        assert (ref[-1][2] == 0).mean() > 0.8
    finally:
        gpu_ctx.set_track_waves(0)
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)


@pytest.mark.parametrize("ww,wh,waves", [(21, 21, 0), (21, 21, 29), (17, 19, 0)])
def test_lookahead_large_window_row_run_launch(klt, ok, gpu_ctx, ww, wh, waves):
    """The multi-frame launch of the as-shipped configuration (21x21, 2 levels; the row-run mapping
    with the footprint staged in LDS, incl. the reuse of a staged footprint across passes and
    frames), also with fewer wavefronts than features and with partial runs (17 wide)."""
    H, W, n, B = 200, 320, 260, 5
    f = _stream(H, W, B + 1, seed=23)
    gp = klt.make_params(window=21, levels=2)
    op = ok.make_params(window=21, levels=2)
    for q in (gp, op):
        q.window_width, q.window_height = ww, wh
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    val[5] = -3
    ref = _oracle_lists(ok, op, f, x, y, val)
    gpu_ctx.set_track_waves(waves)
    la = klt.Lookahead(gpu_ctx, W, H, n, B)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(B)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:B + 1], [o[1] for o in outs])
        la.flush()
        for k in range(B):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
        assert (ref[-1][2] == 0).mean() > 0.7
    finally:
        gpu_ctx.set_track_waves(0)
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)


def test_lookahead_full_size_c3(klt, ok, gpu_ctx):
    """BASELINE C3 as bench.py runs it: 4K frames, 16000 features, 5 levels, 11x11, batches through
    gklt_lookahead with the default cap of 4096 wavefronts (each follows ~4 features)."""
    H, W, n, B = 2160, 3840, 16000, 3
    T = synth.base_texture(H, W, seed=23)
    f = [synth.render_frame(T, H, W, k, seed=23) for k in range(B + 1)]
    gp = klt.make_params(window=11, levels=5)
    op = ok.make_params(window=11, levels=5)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx, nx=160, ny=100)
    val = np.zeros(n, np.int32)
    ok.set_threads(8)
    try:
        ref = _oracle_lists(ok, op, f, x, y, val)
    finally:
        ok.set_threads(1)
    la = klt.Lookahead(gpu_ctx, W, H, n, B)
    outs = [gpu_ctx.pinned_array((3, n), np.float32) for _ in range(B)]
    try:
        la.upload_features(x, y, val)
        la.start(f[0])
        la.submit(f[1:B + 1], [o[1] for o in outs])
        la.flush()
        for k in range(B):
            _same(outs[k][0], ref[k], "frame %d" % (k + 1))
        assert (ref[-1][2] == 0).mean() > 0.9
    finally:
        la.close()
        for _, p in outs:
            gpu_ctx.pinned_free(p)
