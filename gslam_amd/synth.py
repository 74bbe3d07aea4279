"""Synthetic video streams for tests and bench.

The reference's sample sequence (robot/shake.mov) is a Google-Drive download that is not in
the repository (README.md:10,33-42), so every configuration runs on this generator.
"""
import numpy as np

BASE_SEED = 20260711


def base_texture(H, W, seed=BASE_SEED, margin=256, sigma=2.0):
    """Band-limited noise, mean 128 / std 40, float64, (H+margin) x (W+margin)."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    T = rng.standard_normal((H + margin, W + margin))
    T = gaussian_filter(T, sigma, mode="wrap")
    T = (T - T.mean()) / T.std() * 40.0 + 128.0
    return T


def frame_offsets(k, period=64):
    ox = 128.0 + 24.0 * np.sin(2 * np.pi * k / period)
    oy = 128.0 + 16.0 * np.cos(2 * np.pi * k / period)
    return ox, oy


def render_frame(T, H, W, k, seed=BASE_SEED, period=64, noise_sigma=1.5):
    """Frame k = H x W crop of T at a sub-pixel offset (bilinear, fp64) + N(0, noise_sigma^2), u8."""
    ox, oy = frame_offsets(k, period)
    x0, y0 = int(np.floor(ox)), int(np.floor(oy))
    ax, ay = ox - x0, oy - y0
    a = T[y0:y0 + H, x0:x0 + W]
    b = T[y0:y0 + H, x0 + 1:x0 + W + 1]
    c = T[y0 + 1:y0 + H + 1, x0:x0 + W]
    d = T[y0 + 1:y0 + H + 1, x0 + 1:x0 + W + 1]
    img = (1 - ax) * (1 - ay) * a + ax * (1 - ay) * b + (1 - ax) * ay * c + ax * ay * d
    if noise_sigma > 0:
        rng = np.random.default_rng(seed + 1 + k)
        img = img + rng.normal(0.0, noise_sigma, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_stream(H, W, nframes=64, seed=BASE_SEED, period=64, noise_sigma=1.5):
    T = base_texture(H, W, seed)
    return [render_frame(T, H, W, k, seed, period, noise_sigma) for k in range(nframes)]


def lattice_seeds(H, W, n, border, seed=BASE_SEED, nx=None, ny=None):
    """First n points of a jittered nx x ny lattice strictly inside the border (float32 x, y)."""
    if nx is None:
        nx = int(np.ceil(np.sqrt(n * W / float(H))))
        ny = int(np.ceil(n / float(nx)))
    rng = np.random.default_rng(seed + 7)
    m = border + 4.0  # keep clear of the border test even after a few pixels of drift
    gx = np.linspace(m, W - 1 - m, nx)
    gy = np.linspace(m, H - 1 - m, ny)
    xx, yy = np.meshgrid(gx, gy)
    pts = np.stack([xx.ravel(), yy.ravel()], 1)
    jit = 0.35 * min((gx[1] - gx[0]) if nx > 1 else 1.0, (gy[1] - gy[0]) if ny > 1 else 1.0)
    pts = pts + rng.uniform(-jit, jit, pts.shape)
    pts[:, 0] = np.clip(pts[:, 0], m, W - 1 - m)
    pts[:, 1] = np.clip(pts[:, 1], m, H - 1 - m)
    # spread the first n over the whole image rather than the top rows
    perm = rng.permutation(len(pts))[:n]
    perm.sort()
    pts = pts[perm]
    return pts[:, 0].astype(np.float32), pts[:, 1].astype(np.float32)


# BASELINE.json configs (SURVEY.md section 8 shorthand)
CONFIGS = {
    "C2": dict(H=1080, W=1920, levels=4, window=7, n=4000, lattice=(80, 50)),
    "C3": dict(H=2160, W=3840, levels=5, window=11, n=16000, lattice=(160, 100)),
    "Cship": dict(H=1080, W=1920, levels=2, window=21, n=2000, lattice=(60, 34), border=50),
    # re-detection every frame: Uniform selector with gridSize (32, 32), bordery >= 24
    # (SURVEY A.3 hazard 1), then track
    "C5": dict(H=1080, W=1920, levels=4, window=7, n=4000, lattice=(80, 50),
               select=dict(min_eigenvalue=100, mindist=10, grid=(32, 32), min_bordery=24)),
}
