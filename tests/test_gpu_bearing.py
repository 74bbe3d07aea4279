"""GPU parity: gklt_rotate_normalize / gklt_flist_rotate_normalize (ImageGrids.hpp:73-106 over the
tracked features, System.cpp:246-267) vs the CPU oracle.  fp64, unfused, same order: compared
BIT FOR BIT."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rotations(nrot, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(nrot):
        w = rng.normal(0, 2e-3, 3) * k
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        out.append(R)
    return np.array(out)


def _case(n, H=1080, W=1920, gh=20, seed=1):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, W - 1, n).astype(np.float32)
    y = rng.uniform(0, H - 1, n).astype(np.float32)
    k = min(n, 4)
    y[:k] = [0.0, 20.0, 39.999, H - 1.0][:k]
    val = np.zeros(n, np.int32)
    if n >= 10:
        val[rng.integers(0, n, n // 10)] = rng.integers(-5, 0, n // 10)
    invK = np.linalg.inv(np.array([[1500.0, 0, W / 2.0], [0, 1490.0, H / 2.0], [0, 0, 1]]))
    return x, y, val, _rotations(H // gh + 1, seed), invK


def _bits(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    bad = a.view(np.uint64) != b.view(np.uint64)
    assert not bad.any(), "%d/%d doubles differ, max abs %g" % (bad.sum(), bad.size, np.abs(a - b).max())


@pytest.mark.parametrize("n", [1, 257, 4000])
def test_rotate_normalize_host_list(klt, ok, gpu_ctx, n):
    x, y, val, rot, invK = _case(n)
    norm0 = np.full((n, 3), 7.0)
    vec0 = np.full((n, 3), 9.0)
    gn, gv = klt.rotate_normalize(gpu_ctx, x, y, val, 20, rot, invK, norm0, vec0)
    rn, rv = ok.rotate_normalize(x, y, val, 20, rot, invK, norm0, vec0)
    _bits(gn, rn)
    _bits(gv, rv)
    lost = val != 0
    assert (gn[lost] == 7.0).all() and (gv[lost] == 9.0).all()


def test_rotate_normalize_device_list_after_tracking(klt, ok, gpu_ctx):
    """The call pattern of System::Track: track on the device list, then normalise it in place
    (no list transfer in between)."""
    from gslam_amd import synth
    H, W, n = 180, 320, 300
    T = synth.base_texture(H, W, seed=4)
    f0, f1 = (synth.render_frame(T, H, W, k, seed=4, noise_sigma=1.0) for k in range(2))
    gp = klt.make_params(window=7, levels=3)
    op = ok.make_params(window=7, levels=3)
    gpu_ctx.set_params(gp)
    x, y = synth.lattice_seeds(H, W, n, gp.borderx)
    val = np.zeros(n, np.int32)
    rot = _rotations(H // 20 + 1, 8)
    invK = np.linalg.inv(np.array([[300.0, 0, 160.0], [0, 300.0, 90.0], [0, 0, 1]]))
    p0, p1 = klt.Pyramid(gpu_ctx, W, H), klt.Pyramid(gpu_ctx, W, H)
    fl = klt.FeatureList(gpu_ctx, n)
    try:
        p0.build(f0)
        p1.build(f1)
        fl.upload(x, y, val)
        klt.track_flist(gpu_ctx, p0, p1, fl)
        gn, gv = klt.rotate_normalize(gpu_ctx, None, None, None, 20, rot, invK, flist=fl)
        tx, ty, tv = fl.download()
        planes0, planes1 = p0.planes(), p1.planes()
    finally:
        fl.close()
        p0.close()
        p1.close()
    rx, ry, rv = ok.track_features(op, planes0, planes1, x, y, val)[:3]
    assert (tv == rv).all() and (tv == 0).sum() > n // 2
    rn, rv3 = ok.rotate_normalize(rx, ry, rv, 20, rot, invK)
    _bits(gn, rn)
    _bits(gv, rv3)


def test_rotate_normalize_pinned_outputs(klt, ok, gpu_ctx):
    """Page-locked norm/vec are written by the kernel itself; untracked entries stay as they are."""
    import ctypes as C
    n = 1000
    x, y, val, rot, invK = _case(n, seed=5)
    norm, pn = gpu_ctx.pinned_array((n, 3), np.float64)
    vec, pv = gpu_ctx.pinned_array((n, 3), np.float64)
    try:
        norm[:] = 7.0
        vec[:] = 9.0
        rot9 = np.ascontiguousarray(rot, np.float64).reshape(-1, 9)
        K9 = np.ascontiguousarray(invK, np.float64).reshape(9)
        dp = C.POINTER(C.c_double)
        rc = klt.lib().gklt_rotate_normalize(
            gpu_ctx.h, n, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
            val.ctypes.data_as(C.POINTER(C.c_int)), 20, len(rot9), rot9.ctypes.data_as(dp),
            K9.ctypes.data_as(dp), C.cast(C.c_void_p(pn), dp), C.cast(C.c_void_p(pv), dp))
        assert rc == 0
        rn, rv = ok.rotate_normalize(x, y, val, 20, rot, invK, np.full((n, 3), 7.0), np.full((n, 3), 9.0))
        _bits(norm, rn)
        _bits(vec, rv)
    finally:
        gpu_ctx.pinned_free(pn)
        gpu_ctx.pinned_free(pv)


def test_rotate_normalize_argument_errors(klt, gpu_ctx):
    x = np.zeros(2, np.float32)
    v = np.zeros(2, np.int32)
    with pytest.raises(klt.GkltError):
        klt.rotate_normalize(gpu_ctx, x, x, v, 0, np.tile(np.eye(3), (3, 1, 1)), np.eye(3))   # grid height
    with pytest.raises(klt.GkltError):
        klt.rotate_normalize(gpu_ctx, x, x, v, 20, np.eye(3)[None], np.eye(3))               # one rotation
