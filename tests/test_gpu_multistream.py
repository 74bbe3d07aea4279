"""GPU parity: gklt_multistream (several independent streams stepped together: batched pyramid
builds + one tracker launch for all lists) vs the CPU oracle run stream by stream, frame by frame.
Lists are compared EXACTLY after every frame."""
import numpy as np
import pytest

from gslam_amd import synth

pytestmark = pytest.mark.gpu


def _stream(H, W, n, seed, noise=1.5):
    T = synth.base_texture(H, W, seed=seed)
    return [synth.render_frame(T, H, W, k, seed=seed, noise_sigma=noise) for k in range(n)]


def _oracle_lists(ok, op, frames, x, y, val, reseed_before=()):
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


def _same(x, y, v, ref, what):
    rx, ry, rv = ref
    assert (np.asarray(v).view(np.int32) == rv).all(), what
    assert (np.asarray(x).view(np.uint32) == rx.view(np.uint32)).all(), what
    assert (np.asarray(y).view(np.uint32) == ry.view(np.uint32)).all(), what


@pytest.mark.parametrize("S,device_frames,pinned,window,levels,waves", [
    (1, False, True, 7, 3, 0),     # degenerate: a gklt_pipeline
    (3, False, True, 7, 3, 0),
    (5, True, True, 7, 3, 64),     # frames resident in HBM; wavefronts take many (stream, feature) slots
    (3, True, False, 11, 2, 0),    # pageable host_out: copy path
    (18, True, True, 7, 2, 0),     # more streams than one batched build launch takes (16)
    (2, False, True, 21, 2, 0),    # LDS-staged footprint variant of the tracker
    (64, True, True, 7, 4, 0),     # BASELINE C4's stream count: four 16-stream build launches, 64-stream slot space
])
def test_multistream_matches_oracle(klt, ok, gpu_ctx, S, device_frames, pinned, window, levels, waves):
    H, W, cap, nfr = (120, 200, 150, 5) if S <= 5 else ((96, 160, 60, 4) if S < 64 else (192, 320, 90, 4))
    gp = klt.make_params(window=window, levels=levels)
    op = ok.make_params(window=window, levels=levels)
    gpu_ctx.set_params(gp)
    gpu_ctx.set_track_waves(waves)
    streams, seeds, refs = [], [], []
    for s in range(S):
        f = _stream(H, W, nfr, seed=31 + 7 * s)
        n = cap - (s % 3) * 11          # ragged list lengths
        x, y = synth.lattice_seeds(H, W, n, gp.borderx, seed=s)
        val = np.zeros(n, np.int32)
        if n > 7:
            x[5] = 3.0                  # out of bounds from the start
            val[7] = -1                 # lost features stay untouched
        streams.append(f)
        seeds.append((x, y, val))
        refs.append(_oracle_lists(ok, op, f, x, y, val))

    ms = klt.MultiStream(gpu_ctx, S, W, H, cap)
    dptr = [[gpu_ctx.device_upload(a) for a in f] for f in streams] if device_frames else []

    def frames(t):
        if device_frames:
            return [(dptr[s][t], W) for s in range(S)]
        return [streams[s][t] for s in range(S)]

    outs = []
    for _ in range(nfr - 1):
        if pinned:
            outs.append(gpu_ctx.pinned_array((3, S * cap), np.float32))
        else:
            a = np.zeros((3, S * cap), np.float32)
            outs.append((a, a.ctypes.data))
    try:
        for s in range(S):
            ms.upload_features(s, *seeds[s])
        ms.start(frames(0), frames(1))
        for t in range(1, nfr):
            ms.step(frames(t + 1) if t + 1 < nfr else None, outs[t - 1][1])
            if t == 1:
                ms.wait_tracked()
        ms.sync()
        finals = [ms.download(s) for s in range(S)]
        tracked_total, _ = ms.stats()
        # the current pyramid of a stream is that stream's last frame
        py = ms.pyramid(S - 1, 1)
        lvl0 = py.plane(0, 0)
        ref0 = ok.compute_pyramid(op, streams[S - 1][nfr - 1])
    finally:
        ms.close()
        gpu_ctx.set_track_waves(0)
        for f in dptr:
            for p in f:
                gpu_ctx.device_free(p)
    try:
        for s in range(S):
            n = len(seeds[s][0])
            for k in range(nfr - 1):
                o = outs[k][0]
                seg = slice(s * cap, s * cap + n)
                _same(o[0, seg], o[1, seg], o[2, seg], refs[s][k], "stream %d frame %d" % (s, k + 1))
            _same(finals[s][0], finals[s][1], finals[s][2], refs[s][-1], "stream %d final list" % s)
        assert tracked_total == sum(int((r[2] == 0).sum()) for ref in refs for r in ref)
        assert (lvl0.view(np.uint32) == np.asarray(ref0[0], np.float32).view(np.uint32)).all()
    finally:
        if pinned:
            for _, p in outs:
                gpu_ctx.pinned_free(p)


def test_multistream_reseed_one_stream(klt, ok, gpu_ctx):
    """upload_features() of one stream between steps re-seeds that stream only."""
    H, W, cap, nfr, S = 120, 200, 100, 6, 3
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    streams = [_stream(H, W, nfr, seed=3 + s) for s in range(S)]
    x, y = synth.lattice_seeds(H, W, cap, gp.borderx)
    val = np.zeros(cap, np.int32)
    refs = [_oracle_lists(ok, op, streams[s], x, y, val, reseed_before={3} if s == 1 else ()) for s in range(S)]
    ms = klt.MultiStream(gpu_ctx, S, W, H, cap)
    try:
        for s in range(S):
            ms.upload_features(s, x, y, val)
        ms.start([f[0] for f in streams], [f[1] for f in streams])
        for t in range(1, nfr):
            if t == 3:
                ms.upload_features(1, x, y, val)
            ms.step([f[t + 1] for f in streams] if t + 1 < nfr else None)
            ms.sync()
            for s in range(S):
                _same(*ms.download(s), refs[s][t - 1], "stream %d frame %d" % (s, t))
    finally:
        ms.close()


def test_multistream_argument_errors(klt, gpu_ctx):
    gp = klt.make_params(window=7, levels=2)
    gpu_ctx.set_params(gp)
    with pytest.raises(klt.GkltError):
        klt.MultiStream(gpu_ctx, 0, 64, 64, 10)
    with pytest.raises(klt.GkltError):
        klt.MultiStream(gpu_ctx, 65, 64, 64, 10)
    ms = klt.MultiStream(gpu_ctx, 2, 96, 96, 10)
    try:
        with pytest.raises(klt.GkltError):  # not started
            ms.step(None)
        with pytest.raises(klt.GkltError):  # list longer than the capacity
            ms.upload_features(0, np.zeros(11, np.float32), np.zeros(11, np.float32), np.zeros(11, np.int32))
        with pytest.raises(klt.GkltError):
            ms.upload_features(2, np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.int32))
    finally:
        ms.close()
