"""Synthetic two-view scenes for the outlier-rejection tests."""
import numpy as np


def scene(n=2000, out_frac=0.3, seed=0, noise=0.5, baseline=0.4, rot=0.05):
    """Keyframe bearings p1, current-frame bearings (rotated by R, pixel noise, a fraction displaced
    by up to 80 px), list status, isValid flags, R, K, and which pairs are geometric inliers."""
    rng = np.random.default_rng(seed)
    K = np.array([[1000.0, 0, 960], [0, 1000.0, 540], [0, 0, 1]])
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(3, 10, n)], 1)
    R = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1.0]])
    C = baseline * np.array([1.0, 0.125, 0.075])
    p1 = X / np.linalg.norm(X, axis=1, keepdims=True)
    d = X - C
    pix = (K @ (d / d[:, 2:]).T).T
    pix[:, :2] += rng.normal(0, noise, (n, 2))
    nout = int(out_frac * n)
    bad = rng.choice(n, nout, replace=False)
    pix[bad, :2] += rng.uniform(-80, 80, (nout, 2))
    d2 = (np.linalg.inv(K) @ pix.T).T
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    cur = (R @ d2.T).T
    val = np.zeros(n, np.int32)
    val[::50] = -2
    valid = np.ones(n, np.uint8)
    valid[::31] = 0
    truth = np.ones(n, bool)
    truth[bad] = False
    return p1, cur, val, valid, R, K, truth


MIN_VA = float(np.cos(np.radians(0.3)))
MED_VA = float(np.cos(np.radians(1.0)))
