"""ctypes wrapper around oracle/libklt_oracle.so -- TEST INFRASTRUCTURE ONLY.

The CPU restatement of GSLAM's KLT front end (see klt_oracle.h; parity unpinned).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libklt_oracle.so")

KLT_TRACKED, KLT_NOT_FOUND, KLT_SMALL_DET = 0, -1, -2
KLT_MAX_ITERATIONS, KLT_OOB, KLT_LARGE_RESIDUE = -3, -4, -5
SELECT_ALL, REPLACE_SOME = 0, 1
VARIANT_PLAIN, VARIANT_ROUND2, VARIANT_UNIFORM = 0, 1, 2


class Params(C.Structure):
    """Mirror of ok_params (klt_oracle.h) == the KLT_TrackingContextRec fields the path reads."""

    _fields_ = [
        ("mindist", C.c_int),
        ("window_width", C.c_int),
        ("window_height", C.c_int),
        ("min_eigenvalue", C.c_int),
        ("min_determinant", C.c_float),
        ("min_displacement", C.c_float),
        ("max_iterations", C.c_int),
        ("max_residue", C.c_float),
        ("grad_sigma", C.c_float),
        ("smooth_sigma_fact", C.c_float),
        ("pyramid_sigma_fact", C.c_float),
        ("step_factor", C.c_float),
        ("nSkippedPixels", C.c_int),
        ("borderx", C.c_int),
        ("bordery", C.c_int),
        ("nPyramidLevels", C.c_int),
        ("subsampling", C.c_int),
    ]

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("klt_oracle.c", "ransac_oracle.c", "klt_oracle.h", os.path.join("..", "include", "gklt_order.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libklt_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ok_smooth_sigma.restype = C.c_float
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_params():
    p = Params()
    lib().ok_default_params(C.byref(p))
    return p


def make_params(window=7, levels=4, subsampling=2, border=None, **kw):
    """Reference defaults, then the overrides System.cpp:101-109 applies, then the default border."""
    p = default_params()
    p.window_width = p.window_height = window
    p.nPyramidLevels = levels
    p.subsampling = subsampling
    for k, v in kw.items():
        setattr(p, k, v)
    if border is None:
        if lib().ok_update_tc_border(C.byref(p)):
            raise RuntimeError("kernel too wide")
    else:
        p.borderx = p.bordery = border
    return p


def set_threads(n):
    lib().ok_set_threads(int(n))


def compute_kernels(sigma):
    g = np.zeros(71, np.float32)
    d = np.zeros(71, np.float32)
    gw, dw = C.c_int(), C.c_int()
    rc = lib().ok_compute_kernels(C.c_float(sigma), _fp(g), C.byref(gw), _fp(d), C.byref(dw))
    if rc:
        raise ValueError("MAX_KERNEL_WIDTH too small / degenerate sigma %r" % sigma)
    return g[: gw.value].copy(), d[: dw.value].copy()


def update_border(p):
    if lib().ok_update_tc_border(C.byref(p)):
        raise RuntimeError("kernel too wide")
    return p.borderx


def change_tc_pyramid(p, search_range):
    return lib().ok_change_tc_pyramid(C.byref(p), int(search_range))


def level_size(p, W, H, level):
    w, h = C.c_int(), C.c_int()
    lib().ok_level_size(C.byref(p), W, H, level, C.byref(w), C.byref(h))
    return w.value, h.value


def smooth(img, sigma):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(img)
    if lib().ok_smooth(_fp(img), img.shape[1], img.shape[0], C.c_float(sigma), _fp(out)):
        raise ValueError("bad sigma")
    return out


def gradients(img, sigma):
    img = np.ascontiguousarray(img, np.float32)
    gx, gy = np.empty_like(img), np.empty_like(img)
    if lib().ok_gradients(_fp(img), img.shape[1], img.shape[0], C.c_float(sigma), _fp(gx), _fp(gy)):
        raise ValueError("bad sigma")
    return gx, gy


def resize_linear(img, W2, H2):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty((H2, W2), np.float32)
    lib().ok_resize_linear(_fp(img), img.shape[1], img.shape[0], _fp(out), W2, H2)
    return out


def compute_pyramid(p, img):
    """-> list of 3*L float32 planes [I0,gx0,gy0,I1,...] (pyramid.cpp:45-84)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    planes = []
    for l in range(p.nPyramidLevels):
        w, h = level_size(p, W, H, l)
        planes += [np.empty((h, w), np.float32) for _ in range(3)]
    arr = (C.POINTER(C.c_float) * len(planes))(*[_fp(a) for a in planes])
    rc = lib().ok_compute_pyramid(C.byref(p), img.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, W, arr)
    if rc:
        raise RuntimeError("ok_compute_pyramid failed")
    return planes


def track_features(p, pyr1, pyr2, x, y, val, invK=None, want_iters=False):
    """In-place on copies; returns (x, y, val, norm, vec[, iters]) (LK.hpp:972-1046)."""
    H, W = pyr1[0].shape
    x = np.array(x, np.float32)
    y = np.array(y, np.float32)
    val = np.array(val, np.int32)
    n = len(x)
    a1 = (C.POINTER(C.c_float) * len(pyr1))(*[_fp(a) for a in pyr1])
    a2 = (C.POINTER(C.c_float) * len(pyr2))(*[_fp(a) for a in pyr2])
    norm = np.zeros((n, 3), np.float64)
    vec = np.zeros((n, 3), np.float64)
    iters = np.zeros(n, np.int32)
    if invK is None:
        kp = None
        np_, vp = None, None
    else:
        invK = np.ascontiguousarray(invK, np.float64)
        kp = invK.ctypes.data_as(C.POINTER(C.c_double))
        np_ = norm.ctypes.data_as(C.POINTER(C.c_double))
        vp = vec.ctypes.data_as(C.POINTER(C.c_double))
    q = p.copy()
    rc = lib().ok_track_features(C.byref(q), W, H, a1, a2, n, _fp(x), _fp(y), _ip(val), kp, np_, vp, _ip(iters))
    if rc:
        raise RuntimeError("ok_track_features failed")
    if want_iters:
        return x, y, val, norm, vec, iters
    return x, y, val, norm, vec


def count_remaining(val):
    val = np.ascontiguousarray(val, np.int32)
    return lib().ok_count_remaining(len(val), _ip(val))


def rotate_normalize(x, y, val, grid_height, rotations, invK, norm=None, vec=None):
    """ImageGrids.hpp:73-106 on the tracked features; returns (norm, vec), n x 3 doubles."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    val = np.ascontiguousarray(val, np.int32)
    rot = np.ascontiguousarray(rotations, np.float64).reshape(-1, 9)
    K = np.ascontiguousarray(invK, np.float64).reshape(9)
    n = len(x)
    norm = np.zeros((n, 3)) if norm is None else np.array(norm, np.float64)
    vec = np.zeros((n, 3)) if vec is None else np.array(vec, np.float64)
    dp = C.POINTER(C.c_double)
    rc = lib().ok_rotate_normalize(n, _fp(x), _fp(y), _ip(val), int(grid_height), len(rot),
                                   rot.ctypes.data_as(dp), K.ctypes.data_as(dp),
                                   norm.ctypes.data_as(dp), vec.ctypes.data_as(dp))
    if rc:
        raise RuntimeError("ok_rotate_normalize failed")
    return norm, vec


def integral_products(gx, gy):
    gx = np.ascontiguousarray(gx, np.float32)
    gy = np.ascontiguousarray(gy, np.float32)
    H, W = gx.shape
    s = [np.empty((H + 1, W + 1), np.float32) for _ in range(3)]
    lib().ok_integral_products(_fp(gx), _fp(gy), W, H, _fp(s[0]), _fp(s[1]), _fp(s[2]))
    return s  # sumxx, sumyy, sumxy


def select_features(p, mode, variant, sumxx, sumyy, sumxy, nFeatures, x=None, y=None, val=None,
                    capacity=None, grid=(20, 20)):
    H, W = sumxx.shape[0] - 1, sumxx.shape[1] - 1
    capacity = capacity or 100 * max(nFeatures, 1)
    fx = np.full(capacity, -1, np.float32)
    fy = np.full(capacity, -1, np.float32)
    fv = np.full(capacity, -1, np.int32)
    if x is not None:
        fx[: len(x)] = x
        fy[: len(y)] = y
        fv[: len(val)] = val
    n = C.c_int(nFeatures)
    q = p.copy()
    rc = lib().ok_select_features(
        C.byref(q), mode, variant,
        _fp(np.ascontiguousarray(sumxx, np.float32)), _fp(np.ascontiguousarray(sumyy, np.float32)),
        _fp(np.ascontiguousarray(sumxy, np.float32)), W, H, C.byref(n), capacity,
        _fp(fx), _fp(fy), _ip(fv), grid[0], grid[1])
    if rc:
        raise RuntimeError("ok_select_features failed rc=%d" % rc)
    return fx[: n.value].copy(), fy[: n.value].copy(), fv[: n.value].copy()


def box_products(gx, gy, hw, hh):
    """H x W planes of window sums of gx*gx, gy*gy, gx*gy (SURVEY section 7 fast path)."""
    gx = np.ascontiguousarray(gx, np.float32)
    gy = np.ascontiguousarray(gy, np.float32)
    H, W = gx.shape
    b = [np.zeros((H, W), np.float32) for _ in range(3)]
    lib().ok_box_products(_fp(gx), _fp(gy), W, H, int(hw), int(hh), _fp(b[0]), _fp(b[1]), _fp(b[2]))
    return b


def select_features_direct(p, mode, variant, gx, gy, nFeatures, **kw):
    """select_features with the candidate scores formed from direct window sums of the level's
    gradient planes instead of integral images."""
    H, W = gx.shape
    b = box_products(gx, gy, p.window_width // 2, p.window_height // 2)
    dummy = np.zeros((H + 1, W + 1), np.float32)
    lib().ok_set_box_sums(_fp(b[0]), _fp(b[1]), _fp(b[2]))
    try:
        return select_features(p, mode, variant, dummy, dummy, dummy, nFeatures, **kw)
    finally:
        lib().ok_set_box_sums(None, None, None)


def seed_from_level(p, pyr, level, nFeatures):
    """Keyframe seeding (System.cpp:407-420,672-689: keypoints of the coarsest level x mScaleFactor,
    val = KLT_TRACKED) with the Shi-Tomasi selector in place of ORB's FAST detector, as
    gklt_seed_from_level documents it: plain KLTSelectGoodFeatures (selectGoodFeatures.hpp:424-433)
    on the integral images (System.cpp:385-392) of level `level`, border = ceil(border / scale)."""
    scale = p.subsampling ** level
    q = p.copy()
    q.borderx = (p.borderx + scale - 1) // scale
    q.bordery = (p.bordery + scale - 1) // scale
    sums = integral_products(pyr[3 * level + 1], pyr[3 * level + 2])
    x, y, v = select_features(q, 0, 0, *sums, nFeatures)
    found = v >= 0
    x = np.where(found, x * np.float32(scale), x).astype(np.float32)
    y = np.where(found, y * np.float32(scale), y).astype(np.float32)
    v = np.where(found, 0, v).astype(np.int32)
    return x, y, v


def ingest_nv12(y, uv):
    """main.cpp:77-93 in integers: -> (luma, rgb_half).  luma is the Y plane itself (image.copyTo);
    rgb_half = cvtColor(merge(resize(Y, 1/2), split(uv)[0], split(uv)[1]), CV_YCrCb2RGB): 2x2 area
    mean (a+b+c+d+2)>>2 (cv::resize INTER_LINEAR at exactly 1/2, 8 bit), channel 1 read as Cr and
    channel 2 as Cb, OpenCV's 8-bit fixed-point YCrCb->RGB (coefficients 22987, -11698, -5636, 29049,
    shift 14, round half up, saturate).  OpenCV is not vendored: parity with it is unpinned."""
    y = np.ascontiguousarray(y, np.uint8)
    uv = np.ascontiguousarray(uv, np.uint8)
    H, W = y.shape
    assert uv.shape == (H // 2, W)
    yi = y.astype(np.int32)
    Y = (yi[0::2, 0::2] + yi[0::2, 1::2] + yi[1::2, 0::2] + yi[1::2, 1::2] + 2) >> 2
    cr = uv[:, 0::2].astype(np.int32) - 128
    cb = uv[:, 1::2].astype(np.int32) - 128
    r = Y + ((cr * 22987 + 8192) >> 14)
    g = Y + ((cb * -5636 + cr * -11698 + 8192) >> 14)
    b = Y + ((cb * 29049 + 8192) >> 14)
    rgb = np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)
    return y, rgb


def feature_colors(rgb, x, y, val):
    """NormalizeKeyFrameInvoker's colour lookup (System.cpp:310-315): rgb[round(y/2), round(x/2)] for
    live features (C round(): half away from zero; y/2 in fp32); indices clamped; dead -> 0."""
    h2, w2 = rgb.shape[:2]
    def rnd(a):
        a = (np.asarray(a, np.float32) / np.float32(2)).astype(np.float64)
        return (np.sign(a) * np.floor(np.abs(a) + 0.5)).astype(np.int64)
    r = np.clip(rnd(y), 0, h2 - 1)
    c = np.clip(rnd(x), 0, w2 - 1)
    out = rgb[r, c].copy()
    out[np.asarray(val) < 0] = 0
    return out


def relative_ransac(key_norm, cur_norm, val, is_valid, R, K, min_view_angle, med_view_angle, threshold=10.0,
                    prob=0.98, max_iters=1000, seed=0):
    """RelativeOutlierRejection::relativeRansac (Estimation.cpp:621-683; System.cpp:550-564 sets
    threshold 10, prob 0.98).  -> (inlier count or -1, is_valid after, F in pixels, hypotheses examined)."""
    kn = np.ascontiguousarray(key_norm, np.float64).reshape(-1, 3)
    cn = np.ascontiguousarray(cur_norm, np.float64).reshape(-1, 3)
    val = np.ascontiguousarray(val, np.int32)
    valid = np.array(is_valid, np.uint8)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    F = np.zeros(9)
    it = C.c_int()
    dp = C.POINTER(C.c_double)
    f = lib().ok_relative_ransac
    f.restype = C.c_int
    rc = f(len(val), kn.ctypes.data_as(dp), cn.ctypes.data_as(dp), _ip(val), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
           R.ctypes.data_as(dp), K.ctypes.data_as(dp), C.c_double(min_view_angle), C.c_double(med_view_angle),
           C.c_double(threshold), C.c_double(prob), int(max_iters), C.c_uint64(seed), F.ctypes.data_as(dp), C.byref(it))
    return rc, valid, F.reshape(3, 3), it.value
