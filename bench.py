#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: tracked features * frames / s on the synthetic 1080p stream
(config C2: 4000 features, 4-level pyramid, 7x7 window), one independent stream per GPU.

A step = one frame of one stream: its pyramid is built and the feature list is tracked from the
previous frame to it, the tracker writing the frame's list into pinned host memory.  Frames are
resident in HBM before the timed region.  The timed leg hands the frames to gklt_lookahead in
batches of --lookahead frames (pyramids of a batch: one launch per level; tracking of a batch: one
launch; build(batch k) overlaps track(batch k-1) -- the overlap of System.cpp:501-518 with a
deeper preload).  --lookahead 0 times gklt_pipeline_step instead (one frame per call, the
reference's preload depth of one); that mode is also what the latency leg and the
"frame_at_a_time" block report.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|Cship]
N > 1 is launched by torch.distributed.run, one rank per GPU.  Streams are independent, so
there is no data-path collective; the harness's barrier / max-time / sum-count reductions run on
CPU tensors (gloo).

libgklt.so (system ROCm 7.2 HIP runtime) is loaded BEFORE torch is imported: the torch wheel
bundles an older libamdhip64 with the same soname, and HIP-graph capture crashes on it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gslam_amd import klt, synth  # noqa: E402

klt.lib()  # load libgklt.so (and the system HIP runtime) before anything can import torch

RESEED_EVERY = 32
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes per FRAME of a launch, from the rocprofv3 --pmc passes committed under profiles/
# (r01_c2_lookahead16_pmc_by_kernel.txt: FETCH_SIZE and WRITE_SIZE in separate passes, KiB per
# 16-frame launch).  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
# so reads are doubled; WRITE_SIZE is exact.  None where no PMC pass has been committed.
def _pmc_bytes(fetch_kib, write_kib, frames):
    return (2.0 * fetch_kib + write_kib) * 1024.0 / frames


PMC_TRAFFIC = {("C2", "track"): _pmc_bytes(122400.0, 1270.0, 16),
               ("C2", "level0"): _pmc_bytes(76630.0, 421200.0, 16)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--frames", type=int, default=64, help="distinct synthetic frames (looped)")
    ap.add_argument("--lookahead", type=int, default=16,
                    help="frames per gklt_lookahead_submit (0 = gklt_pipeline_step, frame at a time)")
    ap.add_argument("--streams", type=int, default=0,
                    help="BASELINE config 4: this many independent streams IN TOTAL, sharded over the ranks "
                         "(streams/gpus per GPU, stepped together through gklt_multistream, one frame of "
                         "each per call, no lookahead); a step is then one frame of every stream")
    ap.add_argument("--track-waves", type=int, default=0, help="wavefronts per tracker launch (0 = library default)")
    ap.add_argument("--track-impl", type=int, default=0, help="gklt_ctx_set_track_impl (A/B runs; 0 = default)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (numbers are not a scaling result)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--main-only", action="store_true",
                    help="timed + roofline legs only (for rocprofv3 runs: the frame-at-a-time and latency "
                         "legs launch the same kernels with one frame per launch)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto)")
    return ap.parse_args()


class Stream:
    """One video stream on one GPU: frames in HBM, gklt_pipeline (3-deep pyramid ring + device
    feature list)."""

    def __init__(self, ctx, cfg, frames, seeds):
        self.ctx, self.cfg = ctx, cfg
        H, W, n = cfg["H"], cfg["W"], cfg["n"]
        self.pipe = klt.Pipeline(ctx, W, H, n)
        self.fl = self.pipe.flist
        self.ptrs = [ctx.device_upload(f) for f in frames]  # frames resident in HBM
        self.seeds = seeds
        self.t = 0
        # pinned result ring: what the host (System::Track) reads each frame
        self.host = [ctx.pinned_array((3, n), np.float32) for _ in range(4)]

    def frame(self, t):
        return (self.ptrs[t % len(self.ptrs)], self.cfg["W"])

    def start(self):
        self.pipe.upload_features(*self.seeds)
        self.pipe.start(self.frame(0), self.frame(1))
        self.t = 1
        self.next_reseed = 1 + RESEED_EVERY

    def step(self):
        """pyramid(t+1) || track(t-1 -> t) -> async D2H of the list."""
        t = self.t
        if t >= self.next_reseed:  # keep the tracked fraction constant
            self.pipe.upload_features(*self.seeds)
            self.next_reseed += RESEED_EVERY
        self.pipe.step(self.frame(t + 1), self.host[t % 4][1])
        self.t = t + 1

    def run(self, nframes):
        for _ in range(nframes):
            self.step()

    def sync(self):
        self.pipe.sync()

    def close(self):
        self.pipe.close()
        for p in self.ptrs:
            self.ctx.device_free(p)
        for _, p in self.host:
            self.ctx.pinned_free(p)


class LookaheadStream:
    """The same stream through gklt_lookahead: batches of B frames per call."""

    def __init__(self, ctx, cfg, frames, seeds, B):
        assert RESEED_EVERY % B == 0
        self.ctx, self.cfg, self.B = ctx, cfg, B
        H, W, n = cfg["H"], cfg["W"], cfg["n"]
        self.la = klt.Lookahead(ctx, W, H, n, B)
        self.fl = self.la.flist
        self.ptrs = [ctx.device_upload(f) for f in frames]
        self.seeds = seeds
        self.t = 1           # next frame to submit
        self.pending = None  # first frame of the batch that the next submit will track
        self.host = [ctx.pinned_array((3, n), np.float32) for _ in range(4 * B)]

    def frame(self, t):
        return (self.ptrs[t % len(self.ptrs)], self.cfg["W"])

    def start(self):
        self.la.upload_features(*self.seeds)
        self.la.start(self.frame(0))
        self.t, self.pending = 1, None
        self.next_reseed = 1 + RESEED_EVERY

    def _reseed_if_due(self):
        # the next tracker launch starts at frame self.pending: same rule as Stream.step (a
        # re-seed can only fall on the first frame of a batch; B divides RESEED_EVERY)
        t = self.pending
        if t is not None and t >= self.next_reseed:
            self.la.upload_features(*self.seeds)
            self.next_reseed += RESEED_EVERY

    def run(self, nframes):
        """submits nframes frames in batches of B (last one ragged)"""
        done = 0
        while done < nframes:
            b = min(self.B, nframes - done)
            t = self.t
            self._reseed_if_due()
            self.la.submit([self.frame(k) for k in range(t, t + b)],
                           [self.host[k % len(self.host)][1] for k in range(t, t + b)])
            self.pending = t
            self.t = t + b
            done += b

    def sync(self):
        """tracks the pending batch and waits"""
        self._reseed_if_due()
        self.la.flush()
        self.pending = None

    def close(self):
        self.la.close()
        for p in self.ptrs:
            self.ctx.device_free(p)
        for _, p in self.host:
            self.ctx.pinned_free(p)


def algorithmic_bytes(cfg):
    """compulsory pyramid bytes per frame; tracker bytes per plane footprint."""
    H, W = cfg["H"], cfg["W"]
    px, w, h = 0, W, H
    for _ in range(cfg["levels"]):
        px += w * h
        w //= 2
        h //= 2
    b_pyr = W * H + 12 * px
    foot = (cfg["window"] + 1) ** 2 * 4  # one plane, bilinear footprint (w+1)(h+1) fp32
    return b_pyr, foot, px


def cpu_baseline(params_kw, frames, seeds, nframes, threads):
    """The CPU restatement (oracle/, kind 'port') on a bounded sample of the same workload."""
    from oracle import oracle as ok

    ok.build()
    op = ok.make_params(**params_kw)
    ok.set_threads(threads)
    x, y, val = (a.copy() for a in seeds)
    prev = ok.compute_pyramid(op, frames[0])
    t0 = time.perf_counter()
    tracked = 0
    for t in range(1, nframes + 1):
        if t % RESEED_EVERY == 0:
            x, y, val = (a.copy() for a in seeds)
        cur = ok.compute_pyramid(op, frames[t % len(frames)])
        x, y, val = ok.track_features(op, prev, cur, x, y, val)[:3]
        tracked += int((val == 0).sum())
        prev = cur
    dt = time.perf_counter() - t0
    ok.set_threads(1)
    return tracked / dt, dt


def latency_leg(st, n):
    """per-frame latency of gklt_pipeline_step, ms: call -> the frame's list is in pinned host
    memory (gklt_pipeline_wait_tracked; the next frame's build keeps running), and call -> both
    streams drained."""
    lat, full = [], []
    for _ in range(n):
        q0 = time.perf_counter()
        st.step()
        st.pipe.wait_tracked()
        q1 = time.perf_counter()
        st.sync()
        lat.append((q1 - q0) * 1e3)
        full.append((time.perf_counter() - q0) * 1e3)
    return lat, full


def reduce_over_ranks(dist, dt, tracked, iters):
    """max wall time, summed counts (CPU tensors: works on gloo and needs no GPU)."""
    if dist is None:
        return dt, float(tracked), float(iters)
    import torch

    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([float(tracked), float(iters)], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return tt.item(), cnt[0].item(), cnt[1].item()


def rank_seed(rank):
    """Stream sharding: rank r tracks its own stream (BASELINE config 4)."""
    return synth.BASE_SEED + 1000 * rank


def shard_streams(total, world, rank):
    """Config 4 partition: stream s -> rank s mod world.  Returns this rank's stream ids."""
    return [s for s in range(total) if s % world == rank]


class MultiStreamSet:
    """This rank's share of the config-4 streams behind one gklt_multistream.  To bound host-side
    generation time the rank renders at most 2 distinct synthetic sequences; stream s plays
    sequence s mod 2 from phase 5*s (frames are read-only in HBM, so streams share the copies).
    Every stream has its own pyramid ring and feature list and does the full per-frame work."""

    NSEQ = 2

    def __init__(self, ctx, cfg, ids, nframes, gp):
        H, W, n = cfg["H"], cfg["W"], cfg["n"]
        self.ctx, self.cfg, self.ids, self.S, self.nf = ctx, cfg, ids, len(ids), nframes
        self.ms = klt.MultiStream(ctx, self.S, W, H, n)
        self.seq = {}
        for s in ids:
            q = s % self.NSEQ
            if q not in self.seq:
                fr = synth.make_stream(H, W, nframes, seed=rank_seed(q))
                self.seq[q] = [ctx.device_upload(f) for f in fr]
        sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=synth.BASE_SEED,
                                     nx=cfg["lattice"][0], ny=cfg["lattice"][1])
        self.seeds = (sx, sy, np.zeros(len(sx), np.int32))
        self.host = [ctx.pinned_array((3, self.S * n), np.float32) for _ in range(4)]
        self.t = 0

    def frames(self, t):
        W = self.cfg["W"]
        return [(self.seq[s % self.NSEQ][(t + 5 * s) % self.nf], W) for s in self.ids]

    def seed(self):
        for i in range(self.S):
            self.ms.upload_features(i, *self.seeds)

    def start(self):
        self.seed()
        self.ms.start(self.frames(0), self.frames(1))
        self.t = 1

    def step(self):
        t = self.t
        if t % RESEED_EVERY == 0:
            self.seed()
        self.ms.step(self.frames(t + 1), self.host[t % 4][1])
        self.t = t + 1

    def run(self, k):
        for _ in range(k):
            self.step()

    def sync(self):
        self.ms.sync()

    def close(self):
        self.ms.close()
        for fr in self.seq.values():
            for p in fr:
                self.ctx.device_free(p)
        for _, p in self.host:
            self.ctx.pinned_free(p)


def main_multistream(a, ctx, cfg, gp, dist, rank, world):
    """--streams S: S streams in total, S/world per GPU, one frame of each per step."""
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    ids = shard_streams(a.streams, world, rank)
    if not ids:
        raise SystemExit("--streams must be at least the number of GPUs")
    st = MultiStreamSet(ctx, cfg, ids, a.frames, gp)
    st.start()
    steps, warm = a.steps, a.warmup

    def barrier():
        ctx.device_synchronize()
        if dist is not None:
            dist.barrier()

    st.run(warm)
    st.sync()
    st.ms.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    st.run(steps)
    st.sync()
    ctx.device_synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tracked, iters = st.ms.stats(reset=True)
    dt_max, tracked_all, iters_all = reduce_over_ranks(dist, dt, tracked, iters)

    psteps = min(steps, 64)
    ctx.profile_enable(True)
    ctx.profile_reset()
    st.run(psteps)
    st.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    p_tracked, p_iters = st.ms.stats(reset=True)
    lat = []
    for _ in range(min(steps, 64)):
        q0 = time.perf_counter()
        st.step()
        st.ms.wait_tracked()
        lat.append((time.perf_counter() - q0) * 1e3)
        st.sync()
    lat = np.array(lat)

    b_pyr, foot, px = algorithmic_bytes(cfg)
    S = len(ids)
    per_kernel = {k: {"avg_us": ms / cnt * 1e3, "launches": cnt, "total_ms": ms}
                  for k, (ms, cnt) in prof.items() if cnt}
    fpl0 = min(S, 16)  # frames per level launch (streams beyond 16 take further launches)
    alg_by_kernel = {
        "track": (4.0 * n * cfg["levels"] * S + 3.0 * p_iters / psteps) * foot,
        "level0": W * H * (1 + 12 + 1.0) * S * psteps / max(per_kernel.get("level0", {}).get("launches", 1), 1),
    }
    dom = max((k for k in per_kernel if k in alg_by_kernel), key=lambda k: per_kernel[k]["total_ms"])
    alg = alg_by_kernel[dom]
    achieved = alg / (per_kernel[dom]["avg_us"] * 1e-6) / 1e9
    names = {"track": "k_track_w", "level0": "k_level<u8>"}
    out = {
        "metric": "tracked features*frames/sec at 1080p, 4000 pts, 4-level pyramid; per-frame latency",
        "value": tracked_all / dt_max,
        "unit": "tracked_features*frames/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warm,
        "ms_per_step": dt_max / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C4: %d independent streams of %s (%dx%d u8, %d features, %d-level pyramid, "
                               "%dx%d window) sharded stream s -> GPU s mod %d, %d per GPU, one frame of "
                               "every stream per step (no lookahead)" % (
                                   a.streams, a.config, W, H, n, cfg["levels"], cfg["window"], cfg["window"],
                                   world, S),
                   "streams": a.streams, "streams_per_gpu": S,
                   "frames_per_s": a.streams * steps / dt_max,
                   "tracked_fraction": tracked_all / (a.streams * steps * n),
                   "newton_iters_per_feature_frame": iters_all / (a.streams * steps * n),
                   "launch": "gklt_multistream: one launch per pyramid level per 16 streams, one tracker "
                             "launch for all lists, build(t+1) || track(t)",
                   "sequences": "each rank renders %d distinct sequences; stream s plays sequence s mod %d "
                                "from phase 5*s" % (len(st.seq), MultiStreamSet.NSEQ),
                   "parallelism": "stream-sharded x%d, no collective" % world},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                       "mode": "gklt_multistream_step call -> the lists of all of the rank's streams in pinned "
                               "host memory (rank 0)"},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_us": per_kernel[dom]["avg_us"],
                     "frames_per_launch": fpl0 if dom == "level0" else S},
        "kernels": per_kernel,
    }
    st.close()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    dist = None
    if world > 1:
        import torch.distributed as dist  # after libgklt.so: see the module docstring

        dist.init_process_group("gloo")

    cfg = dict(synth.CONFIGS[a.config])
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    params_kw = dict(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
    gp = klt.make_params(**params_kw)
    ctx = klt.Context(0 if a.share_gpu else local_rank, gp)
    ctx.set_track_waves(a.track_waves)
    ctx.set_track_impl(a.track_impl)
    if a.streams > 0:
        return main_multistream(a, ctx, cfg, gp, dist, rank, world)

    seed = rank_seed(rank)
    frames = synth.make_stream(H, W, a.frames, seed=seed)
    sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=seed, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    seeds = (sx, sy, np.zeros(len(sx), np.int32))

    B = a.lookahead

    def barrier():
        ctx.device_synchronize()
        if dist is not None:
            dist.barrier()

    # ---- frame at a time (gklt_pipeline_step): throughput, and synchronous per-frame latency
    # (what System::Track sees).  Runs first, on its own 3-pyramid ring; secondary numbers.
    fa1 = None
    lat, lat_full = [float("nan")], [float("nan")]
    if B and not a.main_only:
        st1 = Stream(ctx, cfg, frames, seeds)
        st1.start()
        st1.run(a.warmup)
        st1.sync()
        st1.fl.stats(reset=True)
        k1 = min(a.steps, 1024)
        q0 = time.perf_counter()
        st1.run(k1)
        st1.sync()
        dt1 = time.perf_counter() - q0
        tracked1, _ = st1.fl.stats(reset=True)
        fa1 = (tracked1, dt1, k1)
        lat, lat_full = latency_leg(st1, min(a.steps, 256))
        st1.close()

    if B:
        st = LookaheadStream(ctx, cfg, frames, seeds, B)
    else:
        st = Stream(ctx, cfg, frames, seeds)
    st.start()
    st.run(a.warmup)
    st.sync()
    st.fl.stats(reset=True)

    barrier()
    t0 = time.perf_counter()
    st.run(a.steps)
    st.sync()
    ctx.device_synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tracked, iters = st.fl.stats(reset=True)
    dt_max, tracked_all, iters_all = reduce_over_ranks(dist, dt, tracked, iters)

    # ---- roofline leg: the same steps, every launch bracketed by HIP events on its own stream
    psteps = min(a.steps, 512)
    ctx.profile_enable(True)
    ctx.profile_reset()
    st.run(psteps)
    st.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    p_tracked, p_iters = st.fl.stats(reset=True)

    # ---- the level kernels alone (empty list: no tracker launch shares the CUs): how far the
    # build is from its own HBM bound when nothing competes with it.  Reported beside the roofline
    # block, never used for `value`.
    alone = {}
    if B and not a.main_only:
        st.seeds = (np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32))
        st.la.upload_features(*st.seeds)  # (the periodic re-seed then uploads the empty list too)
        st.run(2 * B)
        st.sync()
        ctx.profile_enable(True)
        ctx.profile_reset()
        st.run(8 * B)
        st.sync()
        alone = {k: ms / cnt * 1e3 for k, (ms, cnt) in ctx.profile_get().items() if cnt}
        ctx.profile_enable(False)

    if not B:
        fa1 = (tracked, dt, a.steps)
        if not a.main_only:
            lat, lat_full = latency_leg(st, min(a.steps, 256))
    elif fa1 is None:
        fa1 = (float("nan"), 1.0, 1)
    tracked1, dt1, k1 = fa1
    lat, lat_full = np.array(lat), np.array(lat_full)

    b_pyr, foot, px = algorithmic_bytes(cfg)
    per_kernel = {k: {"avg_us": ms / cnt * 1e3, "launches": cnt, "total_ms": ms}
                  for k, (ms, cnt) in prof.items() if cnt}
    fpl = max(B, 1)  # frames per launch
    # Algorithmic bytes per launch.  Tracker: n features x L levels x (3 template
    # planes + 1 residue plane) + 3 planes per Newton iteration, each plane one (w+1)(h+1) fp32
    # bilinear footprint.  Level-0 kernel: u8 in, (I, gx, gy) out, decimated plane out.
    alg_by_kernel = {
        "track": (4.0 * n * cfg["levels"] * fpl + 3.0 * p_iters / psteps * fpl) * foot,
        "level0": fpl * W * H * (1 + 12 + 1.0),
    }
    dom = max((k for k in per_kernel if k in alg_by_kernel), key=lambda k: per_kernel[k]["total_ms"])
    avg_s = per_kernel[dom]["avg_us"] * 1e-6
    alg = alg_by_kernel[dom]
    achieved = alg / avg_s / 1e9
    # HBM bytes per frame from the rocprofv3 PMC passes committed under profiles/ (see PMC_TRAFFIC)
    pmc_traffic = PMC_TRAFFIC.get((a.config, dom))

    names = {"track": "k_track_w", "level0": "k_level<u8>"}
    out = {
        "metric": "tracked features*frames/sec at 1080p, 4000 pts, 4-level pyramid; per-frame latency",
        "value": tracked_all / dt_max,
        "unit": "tracked_features*frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt_max / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: synthetic %dx%d u8 stream, %d features, %d-level pyramid, %dx%d window, "
                               "1 independent stream per GPU" % (a.config, W, H, n, cfg["levels"],
                                                                 cfg["window"], cfg["window"]),
                   "frames_per_s": world * a.steps / dt_max,
                   "tracked_fraction": tracked_all / (world * a.steps * n),
                   "newton_iters_per_feature_frame": iters_all / (world * a.steps * n),
                   "lookahead_frames": B,
                   "launch": ("gklt_lookahead: %d frames per call, one launch per pyramid level and one "
                              "tracker launch per batch, build(k) || track(k-1)" % B) if B else
                             "gklt_pipeline_step: eager, 2 alternating streams, 1 event wait/frame",
                   "parallelism": "stream-sharded x%d, no collective" % world},
        "frame_at_a_time": {"value": tracked1 / dt1, "ms_per_step": dt1 / k1 * 1e3, "steps": k1,
                            "mode": "gklt_pipeline_step (preload depth 1, as System.cpp:501-518), rank 0"},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                       "step_complete_p50": float(np.percentile(lat_full, 50)),
                       "step_complete_p99": float(np.percentile(lat_full, 99)),
                       "mode": "gklt_pipeline_step call -> list of the frame in pinned host memory "
                               "(gklt_pipeline_wait_tracked); step_complete = pyramid(t+1) built as well"},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic * fpl if pmc_traffic else None,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_us": per_kernel[dom]["avg_us"],
                     "frames_per_launch": fpl,
                     "level0_alone": ({"avg_launch_us": alone["level0"],
                                       "achieved": alg_by_kernel["level0"] / (alone["level0"] * 1e-6) / 1e9,
                                       "frac": alg_by_kernel["level0"] / (alone["level0"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       "what": "k_level<u8> with no tracker launch sharing the CUs (empty "
                                               "feature list), same batches"} if "level0" in alone else None),
                     "note": "dominant kernel by total time in the profiled leg; tracker bytes = "
                             "[4*N*L + 3*iters] footprints of (w+1)(h+1)*4 B per frame "
                             "(iters/feature/frame=%.2f), level-0 bytes = W*H*(1+12+1) per frame; "
                             "traffic = (2*FETCH_SIZE + WRITE_SIZE) per frame from the rocprofv3 --pmc passes "
                             "in profiles/r01_c2_lookahead16_pmc_by_kernel.txt (16-frame launches) x "
                             "frames_per_launch; see DESIGN.md 4" % (p_iters / psteps / n)},
        "kernels": per_kernel,
        "pyramid_algorithmic_bytes_per_frame": b_pyr,
    }

    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        ncpu = a.cpu_frames or {"C2": 160, "Cship": 160, "C3": 48}.get(a.config, 24)
        v, cdt = cpu_baseline(params_kw, frames, seeds, ncpu, threads)
        v1, cdt1 = cpu_baseline(params_kw, frames, seeds, max(ncpu // 16, 4), 1)
        out["cpu_baseline"] = {"value": v, "unit": "tracked_features*frames/s", "cores": threads,
                               "kind": "port",
                               "one_core": {"value": v1, "sample": "%d frames, %.1f s wall" % (max(ncpu // 16, 4), cdt1)},
                               "sample": "%d frames of the same stream (pyramid + track), oracle/ C "
                                         "restatement with OpenMP, %.1f s wall" % (ncpu, cdt)}
    st.close()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
