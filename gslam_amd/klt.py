"""Thin ctypes binding of libgklt.so (include/gklt.h), named after the reference's entry points.

This is harness plumbing for tests and bench.py; the product is the C ABI.  The reference's host
API is C++ (KLT.h / KLTUtil.h / pyramid.h / convolve.h / LK.hpp); INTEGRATION.md shows the C++
shim.  There is no CPU fallback here: if libgklt.so is missing or no gfx950 GPU is present the
calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GKLT_LIB") or os.path.join(_HERE, "libgklt.so")  # GKLT_LIB: A/B builds

KLT_TRACKED, KLT_NOT_FOUND, KLT_SMALL_DET = 0, -1, -2
KLT_MAX_ITERATIONS, KLT_OOB, KLT_LARGE_RESIDUE = -3, -4, -5
K_SMOOTH0, K_DOWN, K_GRAD, K_TRACK, K_MISC, K_INTEGRAL, K_SELECT, K_LEVEL0, K_LEVELN, K_RANSAC = range(10)
KERNEL_NAMES = {K_SMOOTH0: "smooth0", K_DOWN: "down", K_GRAD: "grad", K_TRACK: "track", K_MISC: "misc",
                K_INTEGRAL: "integral", K_SELECT: "select", K_LEVEL0: "level0", K_LEVELN: "leveln", K_RANSAC: "ransac"}


class GkltError(RuntimeError):
    pass


class Params(C.Structure):
    """gklt_params == the KLT_TrackingContextRec fields the path reads (KLT.h:18-67)."""

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


_lib = None


def lib():
    """Loads libgklt.so; raises if it has not been built (python -m gslam_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GkltError("%s not built: run `python gslam_amd/build.py`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gklt_last_error.restype = C.c_char_p
        L.gklt_compute_smooth_sigma.restype = C.c_float
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise GkltError("gklt error %d: %s" % (rc, lib().gklt_last_error().decode()))
    return rc


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


# ---- host-only context arithmetic (KLTUtil.h:12-16) -----------------------------------------

def KLTCreateTrackingContext():
    p = Params()
    _check(lib().gklt_default_params(C.byref(p)))
    return p


def KLTChangeTCPyramid(p, search_range):
    n = C.c_int()
    _check(lib().gklt_change_tc_pyramid(C.byref(p), int(search_range), C.byref(n)))
    return n.value


def KLTUpdateTCBorder(p):
    _check(lib().gklt_update_tc_border(C.byref(p)))
    return p.borderx


def computeKernels(sigma):
    g = np.zeros(71, np.float32)
    d = np.zeros(71, np.float32)
    gw, dw = C.c_int(), C.c_int()
    _check(lib().gklt_compute_kernels(C.c_float(sigma), _fp(g), C.byref(gw), _fp(d), C.byref(dw)))
    return g[: gw.value].copy(), d[: dw.value].copy()


def _KLTComputeSmoothSigma(p):
    return lib().gklt_compute_smooth_sigma(C.byref(p))


def KLTCountRemainingFeatures(val):
    val = np.ascontiguousarray(val, np.int32)
    return lib().gklt_count_remaining(len(val), _ip(val))


def make_params(window=7, levels=4, subsampling=2, border=None, **kw):
    """Defaults, then the overrides System.cpp:101-109 applies, then the default border."""
    p = KLTCreateTrackingContext()
    p.window_width = p.window_height = window
    p.nPyramidLevels = levels
    p.subsampling = subsampling
    for k, v in kw.items():
        setattr(p, k, v)
    if border is None:
        KLTUpdateTCBorder(p)
    else:
        p.borderx = p.bordery = border
    return p


# ---- device objects --------------------------------------------------------------------------

class Context:
    def __init__(self, device=0, params=None):
        self.h = C.c_void_p()
        _check(lib().gklt_ctx_create(int(device), C.byref(self.h)))
        self.device = device
        if params is not None:
            self.set_params(params)

    def set_params(self, p):
        _check(lib().gklt_ctx_set_params(self.h, C.byref(p)))

    def get_params(self):
        p = Params()
        _check(lib().gklt_ctx_get_params(self.h, C.byref(p)))
        return p

    def sync(self):
        _check(lib().gklt_sync(self.h))

    def set_track_impl(self, impl):
        """0 = one feature per wavefront (default), 1 = one feature per 8-lane slice."""
        _check(lib().gklt_ctx_set_track_impl(self.h, int(impl)))

    def set_select_impl(self, impl):
        """Plain selector rounds: 0 = 32-bit keys (default), 1 = 64-bit keys."""
        _check(lib().gklt_ctx_set_select_impl(self.h, int(impl)))

    def set_select_scores(self, mode):
        _check(lib().gklt_ctx_set_select_scores(self.h, int(mode)))

    def set_select_iters(self, iters):
        """Fixpoint iterations per tile and suppression launch of the selector (default 3)."""
        _check(lib().gklt_ctx_set_select_iters(self.h, int(iters)))

    def select_rounds(self):
        """Suppression launches of the last selection: (plain or Uniform phase A, Uniform phase B)."""
        r = (C.c_int * 2)()
        _check(lib().gklt_ctx_select_rounds(self.h, r))
        return int(r[0]), int(r[1])

    def set_track_waves(self, waves):
        """wavefronts per tracker launch (0 = default cap of 4096)"""
        _check(lib().gklt_ctx_set_track_waves(self.h, int(waves)))

    def set_chain(self, on):
        _check(lib().gklt_ctx_set_chain(self.h, int(on)))

    def set_coarse_split(self, on):
        _check(lib().gklt_ctx_set_coarse_split(self.h, int(on)))

    def set_pyramid_impl(self, impl):
        """0 = fused level kernels when the radii allow (default), 1 = separate kernels."""
        _check(lib().gklt_ctx_set_pyramid_impl(self.h, int(impl)))

    def close(self):
        if self.h:
            lib().gklt_ctx_destroy(self.h)
            self.h = C.c_void_p()

    # measurement
    def profile_enable(self, on=True):
        _check(lib().gklt_profile_enable(self.h, int(on)))

    def profile_reset(self):
        _check(lib().gklt_profile_reset(self.h))

    def profile_get(self):
        out = {}
        for k, name in KERNEL_NAMES.items():
            ms, n = C.c_double(), C.c_int64()
            _check(lib().gklt_profile_get(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    # memory helpers (harness plumbing)
    def device_upload(self, arr):
        """Copies a numpy array into a fresh device buffer; returns its address (int)."""
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        _check(lib().gklt_device_malloc(self.h, C.c_size_t(arr.nbytes), C.byref(p)))
        _check(lib().gklt_copy_to_device(self.h, p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        return p.value

    def device_free(self, ptr):
        _check(lib().gklt_device_free(self.h, C.c_void_p(ptr)))

    def pinned_array(self, shape, dtype):
        """numpy view of a page-locked host buffer (freed with pinned_free(ptr))."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(lib().gklt_host_malloc_pinned(self.h, C.c_size_t(n), C.byref(p)))
        buf = (C.c_char * n).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape), p.value

    def pinned_free(self, ptr):
        _check(lib().gklt_host_free_pinned(self.h, C.c_void_p(ptr)))

    def device_synchronize(self):
        _check(lib().gklt_device_synchronize(self.h))

    def device_mem_info(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_size_t(), C.c_size_t()
        _check(lib().gklt_device_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    # standalone P1 / P2 (convolve.h:12-13)
    def computeSmoothedImage(self, img, sigma):
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(img)
        _check(lib().gklt_smooth(self.h, _fp(img), img.shape[1], img.shape[0], C.c_float(sigma), _fp(out)))
        return out

    def computeGradients(self, img, sigma):
        img = np.ascontiguousarray(img, np.float32)
        gx, gy = np.empty_like(img), np.empty_like(img)
        _check(lib().gklt_gradients(self.h, _fp(img), img.shape[1], img.shape[0], C.c_float(sigma), _fp(gx), _fp(gy)))
        return gx, gy


class Pyramid:
    """Device-resident counterpart of the reference's std::vector<cv::Mat> pyramid."""

    def __init__(self, ctx, W, H):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.W, self.H = W, H
        _check(lib().gklt_pyramid_create(ctx.h, W, H, C.byref(self.h)))

    @property
    def nlevels(self):
        return lib().gklt_pyramid_levels(self.h)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(lib().gklt_pyramid_level_size(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def build(self, img):
        """computePyramid (pyramid.h:11) from a host u8 frame; asynchronous."""
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.H, self.W)
        self._keep = img
        _check(lib().gklt_pyramid_build(self.ctx.h, self.h, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.strides[0], 0))

    def build_from_device(self, dptr, stride):
        _check(lib().gklt_pyramid_build(self.ctx.h, self.h, C.cast(C.c_void_p(dptr), C.POINTER(C.c_uint8)), stride, 1))

    def plane(self, level, plane):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.float32)
        _check(lib().gklt_pyramid_download(self.ctx.h, self.h, level, plane, _fp(out), w))
        return out

    def plane_u8(self, level):
        """pyramid[3*level].convertTo(CV_8UC1) (System.cpp:403)."""
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        _check(lib().gklt_pyramid_download_u8(self.ctx.h, self.h, level,
                                              out.ctypes.data_as(C.POINTER(C.c_uint8)), w))
        return out

    def planes(self):
        return [self.plane(l, c) for l in range(self.nlevels) for c in range(3)]

    def upload(self, level, plane, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        _check(lib().gklt_pyramid_upload(self.ctx.h, self.h, level, plane, _fp(arr), arr.shape[1]))

    def close(self):
        if self.h:
            lib().gklt_pyramid_destroy(self.h)
            self.h = C.c_void_p()


class FeatureList:
    """Device-resident SoA feature list (x, y, val of KLT_FeatureRec, KLT.h:76-80)."""

    def __init__(self, ctx, capacity):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.capacity = capacity
        _check(lib().gklt_flist_create(ctx.h, capacity, C.byref(self.h)))

    def upload(self, x, y, val):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        val = np.ascontiguousarray(val, np.int32)
        _check(lib().gklt_flist_upload(self.ctx.h, self.h, len(x), _fp(x), _fp(y), _ip(val)))
        self.ctx.sync()

    def __len__(self):
        return lib().gklt_flist_size(self.h)

    def download(self):
        n = len(self)
        x, y, v = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.int32)
        _check(lib().gklt_flist_download(self.ctx.h, self.h, _fp(x), _fp(y), _ip(v)))
        return x, y, v

    def download_async(self, x, y, v):
        _check(lib().gklt_flist_download_async(self.ctx.h, self.h, _fp(x), _fp(y), _ip(v)))

    def download_packed_async(self, dst_ptr):
        _check(lib().gklt_flist_download_packed_async(self.ctx.h, self.h, C.c_void_p(dst_ptr)))

    def stats(self, reset=False):
        t, i = C.c_uint64(), C.c_uint64()
        _check(lib().gklt_flist_stats(self.ctx.h, self.h, C.byref(t), C.byref(i), int(reset)))
        return t.value, i.value

    def close(self):
        if self.h:
            lib().gklt_flist_destroy(self.h)
            self.h = C.c_void_p()


class _Borrowed:
    """A handle owned by another object (no destroy)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, C.c_void_p(h)


class Pipeline:
    """gklt_pipeline: PyramidBuffer (System.h:30-56) + System::Track's per-frame call pattern."""

    def __init__(self, ctx, W, H, capacity):
        L = lib()
        L.gklt_pipeline_flist.restype = C.c_void_p
        L.gklt_pipeline_pyramid.restype = C.c_void_p
        self.ctx, self.W, self.H, self.capacity = ctx, W, H, capacity
        self.h = C.c_void_p()
        _check(L.gklt_pipeline_create(ctx.h, W, H, capacity, C.byref(self.h)))
        self.flist = FeatureList.__new__(FeatureList)
        self.flist.ctx, self.flist.capacity = ctx, capacity
        self.flist.h = C.c_void_p(L.gklt_pipeline_flist(self.h))

    @staticmethod
    def _img(img):
        if isinstance(img, np.ndarray):
            img = np.ascontiguousarray(img, np.uint8)
            return img, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.strides[0], 0
        # (pointer, stride[, on_device = 1]): a device frame, or with on_device = 0 a page-locked
        # host frame that the caller keeps alive until the step's staging copy has run
        ptr, stride = img[0], img[1]
        dev = int(img[2]) if len(img) > 2 else 1
        return None, C.cast(C.c_void_p(ptr), C.POINTER(C.c_uint8)), stride, dev

    def start(self, img0, img1):
        k0, p0, s0, d0 = self._img(img0)
        k1, p1, s1, d1 = self._img(img1)
        assert s0 == s1 and d0 == d1
        _check(lib().gklt_pipeline_start(self.h, p0, p1, s0, d0))
        self.sync()

    def step(self, img_next=None, host_out=None):
        if img_next is None:
            _check(lib().gklt_pipeline_step(self.h, None, 0, 0, C.c_void_p(host_out)))
            return
        keep, p, s, d = self._img(img_next)
        _check(lib().gklt_pipeline_step(self.h, p, s, d, C.c_void_p(host_out)))
        if keep is not None:
            self.sync()  # pageable host frame must stay alive until staged

    def run(self, frames, host_outs=None):
        """gklt_pipeline_run: one gklt_pipeline_step per frame, the loop itself in C.  frames:
        (pointer, stride, on_device) tuples of device or page-locked host frames, same stride and
        residency; host_outs: pointers (or None)."""
        n = len(frames)
        if n == 0:
            return
        stride, dev = frames[0][1], frames[0][2] if len(frames[0]) > 2 else 1
        fp = (C.c_void_p * n)(*[f[0] for f in frames])
        ho = (C.c_void_p * n)(*host_outs) if host_outs is not None else None
        _check(lib().gklt_pipeline_run(self.h, fp, n, int(stride), int(dev), ho))

    def upload_features(self, x, y, val):
        self._seed = (np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32),
                      np.ascontiguousarray(val, np.int32))
        x, y, val = self._seed
        _check(lib().gklt_pipeline_upload_features(self.h, len(x), _fp(x), _fp(y), _ip(val)))

    def sync(self):
        _check(lib().gklt_pipeline_sync(self.h))

    def wait_tracked(self):
        """waits for the list of the last step only (the next frame's build keeps running)"""
        _check(lib().gklt_pipeline_wait_tracked(self.h))

    def pyramid(self, which):
        """which = 0 previous, 1 current (after sync)."""
        py = Pyramid.__new__(Pyramid)
        py.ctx, py.W, py.H = self.ctx, self.W, self.H
        py.h = C.c_void_p(lib().gklt_pipeline_pyramid(self.h, which))
        return py

    def download(self):
        self.sync()
        return self.flist.download()

    def close(self):
        if self.h:
            lib().gklt_pipeline_destroy(self.h)
            self.h = C.c_void_p()


def rotate_normalize(ctx, x, y, val, grid_height, rotations, invK, norm=None, vec=None, flist=None):
    """ImageGrids::rotateAndNormalizePoint over the tracked features (gklt_rotate_normalize, or
    gklt_flist_rotate_normalize when `flist` is given).  Returns (norm, vec)."""
    rot = np.ascontiguousarray(rotations, np.float64).reshape(-1, 9)
    K = np.ascontiguousarray(invK, np.float64).reshape(9)
    dp = C.POINTER(C.c_double)
    if flist is not None:
        n = len(flist)
    else:
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        val = np.ascontiguousarray(val, np.int32)
        n = len(x)
    norm = np.zeros((n, 3)) if norm is None else np.array(norm, np.float64)
    vec = np.zeros((n, 3)) if vec is None else np.array(vec, np.float64)
    if flist is not None:
        _check(lib().gklt_flist_rotate_normalize(ctx.h, flist.h, int(grid_height), len(rot),
                                                 rot.ctypes.data_as(dp), K.ctypes.data_as(dp),
                                                 norm.ctypes.data_as(dp), vec.ctypes.data_as(dp)))
    else:
        _check(lib().gklt_rotate_normalize(ctx.h, n, _fp(x), _fp(y), _ip(val), int(grid_height), len(rot),
                                           rot.ctypes.data_as(dp), K.ctypes.data_as(dp),
                                           norm.ctypes.data_as(dp), vec.ctypes.data_as(dp)))
    return norm, vec


class Lookahead:
    """gklt_lookahead: the frame loop with batches of frames handed over ahead of time."""

    def __init__(self, ctx, W, H, capacity, max_batch):
        L = lib()
        L.gklt_lookahead_flist.restype = C.c_void_p
        L.gklt_lookahead_frames_tracked.restype = C.c_long
        self.ctx, self.W, self.H, self.capacity, self.max_batch = ctx, W, H, capacity, max_batch
        self.h = C.c_void_p()
        _check(L.gklt_lookahead_create(ctx.h, W, H, capacity, max_batch, C.byref(self.h)))
        self.flist = FeatureList.__new__(FeatureList)
        self.flist.ctx, self.flist.capacity = ctx, capacity
        self.flist.h = C.c_void_p(L.gklt_lookahead_flist(self.h))
        self._keep = []

    def start(self, img0):
        keep, p, s, d = Pipeline._img(img0)
        _check(lib().gklt_lookahead_start(self.h, p, s, d))

    def submit(self, frames, host_out=None):
        """frames: list of numpy u8 images or of (device pointer, stride); host_out: list of
        host addresses (pinned or not) or None."""
        n = len(frames)
        parts = [Pipeline._img(f) for f in frames]
        strides = {q[2] for q in parts}
        dev = {q[3] for q in parts}
        assert len(strides) == 1 and len(dev) == 1
        arr = (C.POINTER(C.c_uint8) * n)(*[q[1] for q in parts])
        outs = None
        if host_out is not None:
            outs = (C.c_void_p * n)(*[C.c_void_p(h) for h in host_out])
        _check(lib().gklt_lookahead_submit(self.h, arr, n, strides.pop(), dev.pop(), outs))
        if parts[0][0] is not None:
            _check(lib().gklt_lookahead_sync(self.h))  # pageable host frames must stay alive until staged

    def flush(self):
        _check(lib().gklt_lookahead_flush(self.h))

    def sync(self):
        _check(lib().gklt_lookahead_sync(self.h))

    def upload_features(self, x, y, val):
        self._seed = (np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32),
                      np.ascontiguousarray(val, np.int32))
        x, y, val = self._seed
        _check(lib().gklt_lookahead_upload_features(self.h, len(x), _fp(x), _fp(y), _ip(val)))

    def frames_tracked(self):
        return lib().gklt_lookahead_frames_tracked(self.h)

    def close(self):
        if self.h:
            lib().gklt_lookahead_destroy(self.h)
            self.h = C.c_void_p()


class MultiStream:
    """gklt_multistream: S independent streams on one GPU, one frame of each per step."""

    def __init__(self, ctx, nstreams, W, H, capacity):
        L = lib()
        L.gklt_multistream_pyramid.restype = C.c_void_p
        self.ctx, self.S, self.W, self.H, self.capacity = ctx, nstreams, W, H, capacity
        self.h = C.c_void_p()
        _check(L.gklt_multistream_create(ctx.h, nstreams, W, H, capacity, C.byref(self.h)))
        self._seed = {}

    def _imgs(self, imgs):
        assert len(imgs) == self.S
        parts = [Pipeline._img(f) for f in imgs]
        strides = {q[2] for q in parts}
        dev = {q[3] for q in parts}
        assert len(strides) == 1 and len(dev) == 1
        arr = (C.POINTER(C.c_uint8) * self.S)(*[q[1] for q in parts])
        return parts, arr, strides.pop(), dev.pop()

    def start(self, imgs0, imgs1):
        k0, a0, s0, d0 = self._imgs(imgs0)
        k1, a1, s1, d1 = self._imgs(imgs1)
        assert s0 == s1 and d0 == d1
        _check(lib().gklt_multistream_start(self.h, a0, a1, s0, d0))

    def step(self, imgs_next=None, host_out=None):
        """imgs_next: S numpy u8 images or S (device pointer, stride) pairs; host_out: address of
        3 * S * capacity floats (pinned or not) or None."""
        if imgs_next is None:
            _check(lib().gklt_multistream_step(self.h, None, 0, 0, C.c_void_p(host_out)))
            return
        keep, arr, stride, dev = self._imgs(imgs_next)
        _check(lib().gklt_multistream_step(self.h, arr, stride, dev, C.c_void_p(host_out)))
        if keep[0][0] is not None:
            self.sync()  # pageable host frames must stay alive until staged

    def upload_features(self, stream, x, y, val):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        val = np.ascontiguousarray(val, np.int32)
        self._seed[stream] = (x, y, val)  # the copies are asynchronous: keep the arrays alive
        _check(lib().gklt_multistream_upload_features(self.h, stream, len(x), _fp(x), _fp(y), _ip(val)))

    def download(self, stream):
        n = lib().gklt_multistream_size(self.h, stream)
        _check(min(n, 0))
        x, y, v = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.int32)
        _check(lib().gklt_multistream_download(self.h, stream, _fp(x), _fp(y), _ip(v)))
        return x, y, v

    def stats(self, reset=False):
        t, i = C.c_uint64(), C.c_uint64()
        _check(lib().gklt_multistream_stats(self.h, C.byref(t), C.byref(i), int(reset)))
        return t.value, i.value

    def sync(self):
        _check(lib().gklt_multistream_sync(self.h))

    def wait_tracked(self):
        _check(lib().gklt_multistream_wait_tracked(self.h))

    def pyramid(self, stream, which):
        py = Pyramid.__new__(Pyramid)
        py.ctx, py.W, py.H = self.ctx, self.W, self.H
        py.h = C.c_void_p(lib().gklt_multistream_pyramid(self.h, stream, which))
        return py

    def close(self):
        if self.h:
            lib().gklt_multistream_destroy(self.h)
            self.h = C.c_void_p()


def KLTTrackFeatures(ctx, pyr1, pyr2, x, y, val, invK=None):
    """LK.hpp:972 -- returns updated copies (x, y, val[, norm, vec])."""
    x = np.array(x, np.float32)
    y = np.array(y, np.float32)
    val = np.array(val, np.int32)
    n = len(x)
    if invK is None:
        _check(lib().gklt_track(ctx.h, pyr1.h, pyr2.h, n, _fp(x), _fp(y), _ip(val), None, None, None))
        return x, y, val
    invK = np.ascontiguousarray(invK, np.float64)
    norm = np.zeros((n, 3), np.float64)
    vec = np.zeros((n, 3), np.float64)
    dp = C.POINTER(C.c_double)
    _check(lib().gklt_track(ctx.h, pyr1.h, pyr2.h, n, _fp(x), _fp(y), _ip(val),
                            invK.ctypes.data_as(dp), norm.ctypes.data_as(dp), vec.ctypes.data_as(dp)))
    return x, y, val, norm, vec


NOALLOC_MAX_FEATURES = 32768


def track_noalloc(ctx, pyr1, pyr2, x, y, val):
    """gklt_track_noalloc: KLTTrackFeatures without device allocations -> (x, y, val, tracked)."""
    x = np.array(x, np.float32)
    y = np.array(y, np.float32)
    val = np.array(val, np.int32)
    t = C.c_uint64()
    _check(lib().gklt_track_noalloc(ctx.h, pyr1.h, pyr2.h, len(x), _fp(x), _fp(y), _ip(val), C.byref(t)))
    return x, y, val, t.value


def track_flist(ctx, pyr1, pyr2, fl):
    _check(lib().gklt_track_flist(ctx.h, pyr1.h, pyr2.h, fl.h))


# ---- selector (selectGoodFeatures.hpp) ------------------------------------------------------

SELECT_ALL, REPLACE_SOME = 0, 1
VARIANT_PLAIN, VARIANT_ROUND2, VARIANT_UNIFORM = 0, 1, 2


def integral_products(ctx, pyr):
    """System.cpp:385-392: integral images of gx*gx, gy*gy, gx*gy of level 0 -> 3 x (H+1, W+1)."""
    w, h = pyr.level_size(0)
    s = [np.empty((h + 1, w + 1), np.float32) for _ in range(3)]
    _check(lib().gklt_integral_products(ctx.h, pyr.h, _fp(s[0]), _fp(s[1]), _fp(s[2])))
    return s


def select_features(ctx, mode, variant, sumxx, sumyy, sumxy, nFeatures, x=None, y=None, val=None,
                    capacity=None, grid=(20, 20)):
    """KLTSelectGoodFeatures / KLTReplaceLostFeatures[2|Uniform] on host sums and a host list."""
    H, W = sumxx.shape[0] - 1, sumxx.shape[1] - 1
    capacity = capacity or 100 * max(nFeatures, 1)   # KLTCreateFeatureList, KLTUtil.cpp:53-55
    fx = np.full(capacity, -1, np.float32)
    fy = np.full(capacity, -1, np.float32)
    fv = np.full(capacity, -1, np.int32)
    if x is not None:
        fx[: len(x)] = x
        fy[: len(y)] = y
        fv[: len(val)] = val
    n = C.c_int(nFeatures)
    sums = [np.ascontiguousarray(a, np.float32) for a in (sumxx, sumyy, sumxy)]
    _check(lib().gklt_select(ctx.h, mode, variant, _fp(sums[0]), _fp(sums[1]), _fp(sums[2]), W, H,
                             C.byref(n), capacity, _fp(fx), _fp(fy), _ip(fv), grid[0], grid[1]))
    return fx[: n.value].copy(), fy[: n.value].copy(), fv[: n.value].copy()


def select_from_pyramid(ctx, pyr, mode, variant, fl, grid=(20, 20)):
    _check(lib().gklt_select_from_pyramid(ctx.h, pyr.h, mode, variant, fl.h, grid[0], grid[1]))


def select_prepare(ctx, pyr):
    _check(lib().gklt_select_prepare(ctx.h, pyr.h))


def seed_from_level(ctx, pyr, level, fl, nFeatures):
    """gklt_seed_from_level: keyframe seeds (System.cpp:407-420) selected on pyramid level `level`."""
    _check(lib().gklt_seed_from_level(ctx.h, pyr.h, int(level), fl.h, int(nFeatures)))


class Frame:
    """gklt_frame: a decoded frame on the GPU (Y plane + half-resolution RGB), main.cpp:77-107."""

    def __init__(self, ctx, W, H):
        self.ctx, self.W, self.H = ctx, W, H
        self.h = C.c_void_p()
        _check(lib().gklt_frame_create(ctx.h, W, H, C.byref(self.h)))
        self._keep = None

    def ingest_nv12(self, y, uv):
        """y: H x W u8, uv: H/2 x W u8 (interleaved chroma) numpy arrays (host), or
        ((ptr, stride), (ptr, stride)) device planes."""
        u8 = C.POINTER(C.c_uint8)
        if isinstance(y, np.ndarray):
            y, uv = np.ascontiguousarray(y, np.uint8), np.ascontiguousarray(uv, np.uint8)
            self._keep = (y, uv)
            _check(lib().gklt_frame_ingest_nv12(self.ctx.h, self.h, y.ctypes.data_as(u8), y.strides[0],
                                                uv.ctypes.data_as(u8), uv.strides[0], 0))
            _check(lib().gklt_sync(self.ctx.h))  # pageable sources: stay alive until staged
        else:
            _check(lib().gklt_frame_ingest_nv12(self.ctx.h, self.h, C.cast(C.c_void_p(y[0]), u8), y[1],
                                                C.cast(C.c_void_p(uv[0]), u8), uv[1], 1))

    def ingest_gray(self, y):
        y = np.ascontiguousarray(y, np.uint8)
        self._keep = (y,)
        _check(lib().gklt_frame_ingest_gray(self.ctx.h, self.h, y.ctypes.data_as(C.POINTER(C.c_uint8)), y.strides[0], 0))
        _check(lib().gklt_sync(self.ctx.h))

    def luma(self):
        p, s = C.POINTER(C.c_uint8)(), C.c_int()
        _check(lib().gklt_frame_luma(self.ctx.h, self.h, C.byref(p), C.byref(s)))
        return C.cast(p, C.c_void_p).value, s.value

    def color(self):
        out = np.empty((self.H // 2, self.W // 2, 3), np.uint8)
        _check(lib().gklt_frame_download_color(self.ctx.h, self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               out.strides[0]))
        return out

    def feature_colors(self, fl):
        out = np.zeros((len(fl), 3), np.uint8)
        _check(lib().gklt_flist_colors(self.ctx.h, self.h, fl.h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def close(self):
        if self.h:
            lib().gklt_frame_destroy(self.h)
            self.h = C.c_void_p()


def pyramid_build_frame(ctx, pyr, frame):
    _check(lib().gklt_pyramid_build_frame(ctx.h, pyr.h, frame.h))


def pipeline_step_frame(pipe, frame, host_out=None):
    _check(lib().gklt_pipeline_step_frame(pipe.h, frame.h, C.c_void_p(host_out)))


def relative_ransac(ctx, key_norm, cur_norm, val, is_valid, R, K, min_view_angle, med_view_angle, threshold=10.0,
                    prob=0.98, max_iters=1000, seed=0):
    """gklt_relative_ransac (Estimation.cpp:621-683) -> (inliers or -1, is_valid after, F, examined)."""
    kn = np.ascontiguousarray(key_norm, np.float64).reshape(-1, 3)
    cn = np.ascontiguousarray(cur_norm, np.float64).reshape(-1, 3)
    val = np.ascontiguousarray(val, np.int32)
    valid = np.array(is_valid, np.uint8)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    F = np.zeros(9)
    inl, it = C.c_int(), C.c_int()
    dp = C.POINTER(C.c_double)
    _check(lib().gklt_relative_ransac(
        ctx.h, len(val), kn.ctypes.data_as(dp), cn.ctypes.data_as(dp), _ip(val),
        valid.ctypes.data_as(C.POINTER(C.c_uint8)), R.ctypes.data_as(dp), K.ctypes.data_as(dp),
        C.c_double(min_view_angle), C.c_double(med_view_angle), C.c_double(threshold), C.c_double(prob),
        int(max_iters), C.c_uint64(seed), C.byref(inl), F.ctypes.data_as(dp), C.byref(it)))
    return inl.value, valid, F.reshape(3, 3), it.value
