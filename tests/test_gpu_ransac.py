"""gklt_relative_ransac against oracle/ransac_oracle.c (the restatement of
Estimation.cpp:621-683 with the RANSAC this build defines in place of cv::findFundamentalMat, whose
sampler is OpenCV-internal: parity with the reference's random choices is unpinned).  The GPU path
evaluates all hypotheses at once and replays the sequential bookkeeping; masks, counts, the number
of hypotheses examined and the matrix must equal the sequential oracle exactly."""
import numpy as np
import pytest

from ransac_scene import MED_VA, MIN_VA, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(klt):
    c = klt.Context(0, klt.make_params(window=7, levels=2))
    yield c
    c.close()


@pytest.mark.parametrize("n,out_frac,seed,noise,max_iters", [
    (2000, 0.3, 0, 0.5, 1000), (4000, 0.5, 1, 1.0, 1000), (300, 0.1, 2, 0.2, 1000), (2000, 0.7, 3, 0.5, 1000),
    (64, 0.2, 4, 0.3, 200), (16000, 0.4, 5, 0.8, 500), (2000, 0.0, 6, 0.0, 1000),
])
def test_ransac_equals_sequential_oracle(klt, ok, ctx, n, out_frac, seed, noise, max_iters):
    p1, cur, val, valid, R, K, truth = scene(n, out_frac, seed, noise)
    for s in (11, 12345678901234567):
        want = ok.relative_ransac(p1, cur, val, valid, R, K, MIN_VA, MED_VA, 10.0, 0.98, max_iters, s)
        got = klt.relative_ransac(ctx, p1, cur, val, valid, R, K, MIN_VA, MED_VA, 10.0, 0.98, max_iters, s)
        assert got[0] == want[0] and got[3] == want[3]
        assert (got[1] == want[1]).all()
        assert (got[2].view(np.uint64) == want[2].view(np.uint64)).all()     # same matrix, bit for bit
    cand = (val == 0) & (valid == 1)
    kept = got[1].astype(bool) & cand
    assert got[0] == kept.sum()
    if out_frac <= 0.5:       # it does its job: true inliers stay, gross outliers go
        assert kept[truth & cand].mean() > 0.9     # (no refit on the inliers, as in OpenCV's RANSAC)
        assert truth[kept].mean() > 0.85
    assert (got[1][~cand] == valid[~cand]).all()                              # other entries untouched


def test_view_angle_gate_and_minus_one(klt, ok, ctx):
    """baseline ~0: every view angle is above minViewAngle -> all candidates lose isValid, none is left
    below medViewAngle: 0 < 0.5 * overall -> -1 (Estimation.cpp:643-646,668-670)."""
    p1, cur, val, valid, R, K, _ = scene(500, 0.0, 7, 0.0, baseline=1e-4)
    want = ok.relative_ransac(p1, cur, val, valid, R, K, MIN_VA, MED_VA)
    got = klt.relative_ransac(ctx, p1, cur, val, valid, R, K, MIN_VA, MED_VA)
    assert got[0] == want[0] == -1
    assert (got[1] == want[1]).all() and got[1][(val == 0) & (valid == 1)].sum() == 0
    # moderate parallax below the median threshold for most, but with minViewAngle = -1 nothing passes the gate
    p1, cur, val, valid, R, K, _ = scene(500, 0.0, 8, 0.0)
    got = klt.relative_ransac(ctx, p1, cur, val, valid, R, K, -1.0, MED_VA)
    assert got[0] == -1 and got[1][val == 0].sum() == 0


def test_fewer_than_seven_pairs_and_empty(klt, ok, ctx):
    p1, cur, val, valid, R, K, _ = scene(6, 0.0, 9, 0.0)
    val[:] = 0
    valid[:] = 1
    want = ok.relative_ransac(p1, cur, val, valid, R, K, MIN_VA, MED_VA)
    got = klt.relative_ransac(ctx, p1, cur, val, valid, R, K, MIN_VA, MED_VA)
    assert got[0] == want[0] == 0 and (got[1] == 1).all() and got[3] == 0
    got = klt.relative_ransac(ctx, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.uint8),
                              R, K, MIN_VA, MED_VA)
    assert got[0] == 0
    with pytest.raises(klt.GkltError):
        klt.relative_ransac(ctx, p1, cur, val, valid, R, K, MIN_VA, MED_VA, max_iters=0)
