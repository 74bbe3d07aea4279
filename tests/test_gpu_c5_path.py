"""GPU parity of the path `bench.py --config C5` runs (BASELINE config 5), frame after frame on ONE
context: gklt_select_from_pyramid(SELECT_ALL, UNIFORM, 32x32-px cells) on a device list of capacity
100*n (KLTUtil.cpp:53-55) that the selector GROWS (selectGoodFeatures.hpp:1257) and that keeps its
grown size for the next selection, then gklt_track_flist on the grown list.  The selector keeps state
between calls (round hint of the previous selection, cell ranks cached on the device, scratch sized
for the largest image so far); a different image size must reset what depends on it.  Every list and
every tracked result must equal the loop `bench.py::cpu_baseline_c5` runs on the oracle, slot by slot.
"""
import numpy as np
import pytest

from gslam_amd import synth

pytestmark = pytest.mark.gpu

GRID = (32, 32)


def _same(got, ref, what):
    for a, b, name in zip(got, ref, "xyv"):
        assert len(a) == len(b), "%s: %s has %d entries, oracle %d" % (what, name, len(a), len(b))
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, "%s: %s differs at %d slots, first %s: gpu %s ref %s" % (
            what, name, len(bad), bad[:5], a[bad[:5]], b[bad[:5]])


def _params(klt, ok, bordery):
    kw = dict(window=7, levels=4, min_eigenvalue=100, mindist=10)
    gp, op = klt.make_params(**kw), ok.make_params(**kw)
    gp.bordery = max(gp.bordery, bordery)       # SURVEY A.3 hazard 1, as bench.py does
    op.bordery = max(op.bordery, bordery)
    return gp, op


def _run_stream(klt, ok, ctx, fl, frames, n_list, what, rounds_seen, prepare=False):
    """C5 loop over `frames` with the list at `n_list` entries; returns the list length afterwards.
    prepare: the integral images of frame t+1 are computed ahead (gklt_select_prepare), on the build
    stream, while the selection on frame t runs -- as bench.py does; the first selection finds
    nothing prepared and computes its own."""
    H, W = frames[0].shape
    gp, op = _params(klt, ok, H - GRID[1] * (H // GRID[1]))
    ctx.set_params(gp)
    ring = [klt.Pyramid(ctx, W, H) for _ in range(3)]
    try:
        ring[0].build(frames[0])
        o_cur = ok.compute_pyramid(op, frames[0])
        for t in range(len(frames) - 1):
            cur, nxt = ring[t % 3], ring[(t + 1) % 3]
            nxt.build(frames[t + 1])
            if prepare:
                klt.select_prepare(ctx, nxt)
            klt.select_from_pyramid(ctx, cur, klt.SELECT_ALL, klt.VARIANT_UNIFORM, fl, grid=GRID)
            rounds_seen.append(ctx.select_rounds())
            sel = fl.download()
            o_nxt = ok.compute_pyramid(op, frames[t + 1])
            sums = ok.integral_products(o_cur[1], o_cur[2])
            ref = ok.select_features(op, ok.SELECT_ALL, ok.VARIANT_UNIFORM, *sums, n_list, grid=GRID,
                                     capacity=fl.capacity)
            _same(sel, ref, "%s frame %d selection" % (what, t))
            klt.track_flist(ctx, cur, nxt, fl)
            trk = fl.download()
            rt = ok.track_features(op, o_cur, o_nxt, *ref)[:3]
            _same(trk, rt, "%s frame %d track" % (what, t))
            assert (trk[2] == 0).sum() > 0.5 * (ref[2] > 0).sum()
            n_list = len(ref[0])               # the list keeps the size the selection grew it to
            assert len(fl) == n_list
            o_cur = o_nxt
    finally:
        for p in ring:
            p.close()
    return n_list


@pytest.mark.parametrize("prepare", [False, True])
def test_c5_bench_path_matches_oracle(klt, ok, gpu_ctx, prepare):
    n = 100
    fl = klt.FeatureList(gpu_ctx, 100 * n)
    rounds = []
    try:
        fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
        small = synth.make_stream(270, 480, 5, seed=synth.BASE_SEED + 5)
        n1 = _run_stream(klt, ok, gpu_ctx, fl, small, n, "270x480", rounds, prepare)
        assert n1 > n                                       # the first selection grew the list (:1257)
        # the slots ran out in the first selection: capped phase (phase B) after the uncapped one
        assert rounds[0][0] >= 1 and rounds[0][1] >= 1, rounds
        assert all(r[0] >= 1 for r in rounds)
        # another image size on the same context and list: the round hint and the cell ranks of the
        # 270x480 grid must not be used (gklt_select.hip sel_get / uni ranks)
        big = synth.make_stream(1080, 1920, 2, seed=synth.BASE_SEED + 6)
        rounds_big = []
        n2 = _run_stream(klt, ok, gpu_ctx, fl, big, n1, "1080p", rounds_big, prepare)
        assert n2 > n1
        # and back to the small size with the 1080p list length: every slot beyond the accepted
        # candidates is KLT_NOT_FOUND, the list does not shrink
        again = []
        n3 = _run_stream(klt, ok, gpu_ctx, fl, small[:3], n2, "270x480 again", again, prepare)
        assert n3 == n2
        assert all(r[0] >= 1 and r[1] == 0 for r in again), again   # slots never ran out: no phase B
    finally:
        fl.close()


@pytest.mark.parametrize("hint", [1, 40])
def test_c5_round_hint_too_small_or_too_large(klt, ok, gpu_ctx, hint):
    """The number of suppression launches enqueued before the first host check comes from the
    previous selection; a selection that needs more (or far fewer) launches than the hint gives the
    same list.  One local iteration per launch makes the first selection need many launches, 40 per
    launch makes the next one need few."""
    H, W, n = 270, 480, 100
    frames = synth.make_stream(H, W, 3, seed=synth.BASE_SEED + 9)
    gp, op = _params(klt, ok, H - GRID[1] * (H // GRID[1]))
    gpu_ctx.set_params(gp)
    fl = klt.FeatureList(gpu_ctx, 100 * n)
    pyr = klt.Pyramid(gpu_ctx, W, H)
    try:
        fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
        n_list = n
        for t, iters in enumerate([hint, 41 - hint, hint]):
            gpu_ctx.set_select_iters(iters)
            pyr.build(frames[t])
            klt.select_from_pyramid(gpu_ctx, pyr, klt.SELECT_ALL, klt.VARIANT_UNIFORM, fl, grid=GRID)
            sel = fl.download()
            o = ok.compute_pyramid(op, frames[t])
            ref = ok.select_features(op, ok.SELECT_ALL, ok.VARIANT_UNIFORM, *ok.integral_products(o[1], o[2]),
                                     n_list, grid=GRID, capacity=fl.capacity)
            _same(sel, ref, "frame %d (%d iterations per launch)" % (t, iters))
            n_list = len(ref[0])
    finally:
        gpu_ctx.set_select_iters(3)
        pyr.close()
        fl.close()


def test_select_prepare_is_tied_to_the_build(klt, ok, gpu_ctx):
    """Prepared integral images are used only for the pyramid and the build they were computed from:
    rebuilding the pyramid with another frame, or selecting on another pyramid, computes fresh ones."""
    H, W, n = 135, 240, 60
    f = synth.make_stream(H, W, 3, seed=synth.BASE_SEED + 11)
    gp, op = _params(klt, ok, 8)
    gpu_ctx.set_params(gp)
    pa, pb = klt.Pyramid(gpu_ctx, W, H), klt.Pyramid(gpu_ctx, W, H)
    fl = klt.FeatureList(gpu_ctx, 100 * n)
    try:
        def check(pyr, frame, what):
            fl.upload(np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
            klt.select_from_pyramid(gpu_ctx, pyr, klt.SELECT_ALL, klt.VARIANT_PLAIN, fl)
            o = ok.compute_pyramid(op, frame)
            ref = ok.select_features(op, ok.SELECT_ALL, ok.VARIANT_PLAIN, *ok.integral_products(o[1], o[2]), n)
            _same(fl.download(), ref, what)
        pa.build(f[0]); klt.select_prepare(gpu_ctx, pa)
        check(pa, f[0], "prepared and used")
        pa.build(f[0]); klt.select_prepare(gpu_ctx, pa)
        pa.build(f[1])                                   # rebuilt after the preparation: stale
        check(pa, f[1], "stale preparation ignored")
        pb.build(f[2]); pa.build(f[0]); klt.select_prepare(gpu_ctx, pa)
        check(pb, f[2], "another pyramid")
        check(pa, f[0], "still there for its own pyramid")
        klt.select_prepare(gpu_ctx, pa); klt.select_prepare(gpu_ctx, pb); klt.select_prepare(gpu_ctx, pa)
        check(pb, f[2], "two preparations kept")
        check(pa, f[0], "the newer one of the same pyramid")
    finally:
        fl.close(); pa.close(); pb.close()
