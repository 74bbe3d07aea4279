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

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|Cship|C5] [--streams S]
N > 1: one rank per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) the
process is one of the N ranks; started plainly with --gpus N > 1 it is the launcher: before any HIP
call it starts N fresh child processes (never an exec of a process that has touched the GPU),
forwards rank 0's JSON line and fails if any child failed.  Streams are independent, so there is no
data-path collective; the harness's barrier / max-time / sum-count reductions run on CPU tensors
(gloo).

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
# libgklt.so (and with it the system HIP runtime) is loaded by main() -- before torch can be
# imported, and only in a process that is a rank: the launcher never touches HIP.

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


def parse(argv=None):
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
    ap.add_argument("--pyramid-impl", type=int, default=0, help="gklt_ctx_set_pyramid_impl (A/B runs; 0 = default)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (numbers are not a scaling result)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--main-only", action="store_true",
                    help="timed + roofline legs only (for rocprofv3 runs: the frame-at-a-time and latency "
                         "legs launch the same kernels with one frame per launch)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the pcie_inclusive legs")
    ap.add_argument("--no-shim", action="store_true", help="skip the shim_drop_in block")
    ap.add_argument("--select-scores", default="integral", choices=["integral", "direct"],
                    help="C5: candidate scores from fp32 integral images (the reference's inputs) or from window "
                         "sums formed directly (SURVEY section 7's fast path; gklt_ctx_set_select_scores)")
    ap.add_argument("--no-select-prepare", action="store_true",
                    help="C5: compute the integral images inside the selection (no gklt_select_prepare)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N launcher: seconds after which ranks still running are killed")
    ap.add_argument("--launch-grace", type=float, default=10.0,
                    help="--gpus N launcher: seconds the other ranks get after one rank has failed")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # tests: this rank exits 3 at once
    ap.add_argument("--select-variant", default="uniform", choices=["uniform", "plain", "round2"],
                    help="--config C5: selector flavour run every frame")
    ap.add_argument("--dry-launch", action="store_true",
                    help="ranks print their rank / world / device and exit without touching the GPU "
                         "(CPU test of the launcher)")
    return ap.parse_args(argv)


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a, argv):
    """--gpus N > 1 outside torch.distributed.run: this process starts N ranks of this script as
    fresh children (it has made no HIP call and makes none), relays rank 0's stdout and returns a
    non-zero code if any rank failed."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, text=True))
    # A rank that dies before the first barrier leaves the others waiting in it for ever: when one
    # rank has failed the rest get a short grace period, and the whole job has a deadline; what is
    # still running then is killed (these exact children, by handle).
    rc = 0
    deadline = time.monotonic() + a.launch_timeout
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now
        if now > deadline or (failed_at is not None and now > failed_at + a.launch_grace):
            for r, p in enumerate(procs):
                if p.poll() is None:
                    print("bench.py: killing rank %d (%s)" % (
                        r, "deadline of %.0f s passed" % a.launch_timeout if now > deadline else "another rank failed"),
                        file=sys.stderr)
                    p.kill()
            break
        time.sleep(0.05)
    outs = []
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        outs.append(out)
        if p.returncode != 0:
            print("bench.py: rank %d exited with code %d" % (r, p.returncode), file=sys.stderr)
            rc = rc or (p.returncode if p.returncode > 0 else 1)
    if a.dry_launch:
        sys.stdout.write("".join(outs))
    else:
        sys.stdout.write(outs[0])
        for r, out in enumerate(outs[1:], 1):
            if out.strip():
                sys.stderr.write("[rank %d] %s" % (r, out))
    sys.stdout.flush()
    return rc


class FrameStore:
    """The synthetic frames of one stream where the timed region finds them: in HBM (`value` and
    every roofline figure), or in page-locked host memory (the pcie_inclusive legs: the library
    stages each frame H2D itself, as at the computePyramid boundary of SURVEY 3.1)."""

    def __init__(self, ctx, frames, host=False):
        self.ctx, self.host = ctx, host
        self.W = frames[0].shape[1]
        if host:
            self.bufs = []
            for f in frames:
                arr, p = ctx.pinned_array(f.shape, np.uint8)
                arr[:] = f
                self.bufs.append(p)
        else:
            self.bufs = [ctx.device_upload(f) for f in frames]

    def frame(self, t):
        return (self.bufs[t % len(self.bufs)], self.W, 0 if self.host else 1)

    def close(self):
        for p in self.bufs:
            (self.ctx.pinned_free if self.host else self.ctx.device_free)(p)
        self.bufs = []


class Stream:
    """One video stream on one GPU: frames in HBM, gklt_pipeline (3-deep pyramid ring + device
    feature list)."""

    def __init__(self, ctx, cfg, frames, seeds, host_frames=False):
        self.ctx, self.cfg = ctx, cfg
        H, W, n = cfg["H"], cfg["W"], cfg["n"]
        self.pipe = klt.Pipeline(ctx, W, H, n)
        self.fl = self.pipe.flist
        self.frames = FrameStore(ctx, frames, host_frames)  # resident in HBM, or page-locked host
        self.seeds = seeds
        self.t = 0
        # pinned result ring: what the host (System::Track) reads each frame
        self.host = [ctx.pinned_array((3, n), np.float32) for _ in range(4)]

    def frame(self, t):
        return self.frames.frame(t)

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
        self.frames.close()
        for _, p in self.host:
            self.ctx.pinned_free(p)


class LookaheadStream:
    """The same stream through gklt_lookahead: batches of B frames per call."""

    def __init__(self, ctx, cfg, frames, seeds, B, host_frames=False):
        assert RESEED_EVERY % B == 0
        self.ctx, self.cfg, self.B = ctx, cfg, B
        H, W, n = cfg["H"], cfg["W"], cfg["n"]
        self.la = klt.Lookahead(ctx, W, H, n, B)
        self.fl = self.la.flist
        self.frames = FrameStore(ctx, frames, host_frames)
        self.seeds = seeds
        self.t = 1           # next frame to submit
        self.pending = None  # first frame of the batch that the next submit will track
        self.host = [ctx.pinned_array((3, n), np.float32) for _ in range(4 * B)]

    def frame(self, t):
        return self.frames.frame(t)

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
        self.frames.close()
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


CPU_SAMPLE_SECONDS = 12.0  # the cpu_baseline legs run about this long (bounded: 10-30 s of CPU work)


def cpu_baseline(params_kw, frames, seeds, nframes, threads, seconds=CPU_SAMPLE_SECONDS):
    """The CPU restatement (oracle/, kind 'port') on a bounded sample of the same workload: at
    least `nframes` frames, then on until `seconds` of wall time have gone.  -> (rate, wall, frames)"""
    from oracle import oracle as ok

    ok.build()
    op = ok.make_params(**params_kw)
    ok.set_threads(threads)
    x, y, val = (a.copy() for a in seeds)
    prev = ok.compute_pyramid(op, frames[0])
    t0 = time.perf_counter()
    tracked, t = 0, 0
    while t < nframes or time.perf_counter() - t0 < seconds:
        t += 1
        if t % RESEED_EVERY == 0:
            x, y, val = (a.copy() for a in seeds)
        cur = ok.compute_pyramid(op, frames[t % len(frames)])
        x, y, val = ok.track_features(op, prev, cur, x, y, val)[:3]
        tracked += int((val == 0).sum())
        prev = cur
    dt = time.perf_counter() - t0
    ok.set_threads(1)
    return tracked / dt, dt, t


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


def pcie_legs(ctx, cfg, frames, seeds, B, steps, warmup):
    """PCIe-inclusive rates (never `value`): frames start in page-locked HOST memory and are staged
    H2D by the library inside the timed region; lists come back into pinned host memory.  Both
    call patterns: gklt_pipeline_step (frame at a time) and gklt_lookahead_submit (B per call)."""
    out = {}
    legs = [("frame_at_a_time", 0)] + ([("lookahead", B)] if B else [])
    for name, b in legs:
        st = LookaheadStream(ctx, cfg, frames, seeds, b, host_frames=True) if b else \
            Stream(ctx, cfg, frames, seeds, host_frames=True)
        st.start()
        st.run(warmup)
        st.sync()
        st.fl.stats(reset=True)
        k = min(steps, 1024)
        q0 = time.perf_counter()
        st.run(k)
        st.sync()
        ctx.device_synchronize()
        dt = time.perf_counter() - q0
        tracked, _ = st.fl.stats(reset=True)
        out[name] = {"value": tracked / dt, "ms_per_step": dt / k * 1e3, "steps": k,
                     "h2d_GBps": k * cfg["H"] * cfg["W"] / dt / 1e9}
        st.close()
    out["mode"] = ("u8 frames in page-locked host memory, staged H2D by the library inside the timed "
                   "region (img_on_device = 0); lists written to pinned host memory; rank 0")
    return out


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


def shim_drop_in(frames, steps=256, warmup=16):
    """The path an unchanged GSLAM gets (north_star: "System code calls it unchanged through a thin
    C-ABI layer"): shim/gklt_shim.cpp's computePyramid on a preload std::thread + KLTTrackFeatures per
    frame, the loop of System.cpp:501-518, frames in PAGEABLE host memory, AoS KLT_FeatureList,
    bearing vectors included -- timed inside tests/shim_mock/shim_driver.cpp (drv_track_stream), no
    Python in the loop.  C2 and as-shipped (Cship) parameters on the same 1080p frames."""
    import ctypes as C

    from gslam_amd import build as b

    if not os.path.exists(b.SHIM_TEST_LIB):
        return {"error": "%s not built (python gslam_amd/build.py)" % b.SHIM_TEST_LIB}
    drv = C.CDLL(b.SHIM_TEST_LIB)
    H, W = frames[0].shape
    stack = np.ascontiguousarray(np.stack(frames))
    invK = np.linalg.inv(np.array([[1400.0, 0, W / 2], [0, 1400.0, H / 2], [0, 0, 1]]))
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    res = {}
    for name in ("C2", "Cship"):
        cfg = synth.CONFIGS[name]
        if (cfg["H"], cfg["W"]) != (H, W):
            continue
        n = cfg["n"]
        p = klt.make_params(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
        sx, sy = synth.lattice_seeds(H, W, n, p.borderx, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
        out = np.zeros(4)
        x, y, v = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        rc = drv.drv_track_stream(W, H, len(frames), stack.ctypes.data_as(C.POINTER(C.c_uint8)), cfg["window"],
                                  cfg["levels"], cfg.get("border", -1) or -1, n, sx.ctypes.data_as(fp),
                                  sy.ctypes.data_as(fp), RESEED_EVERY, warmup, steps, invK.ctypes.data_as(dp),
                                  out.ctypes.data_as(dp), x.ctypes.data_as(fp), y.ctypes.data_as(fp),
                                  v.ctypes.data_as(ip))
        if rc:
            res[name] = {"error": "drv_track_stream rc=%d" % rc}
            continue
        res[name] = {"value": out[1] / out[0], "unit": "tracked_features*frames/s", "ms_per_frame": out[0] / steps * 1e3,
                     "frames_per_s": steps / out[0], "p50_ms": out[2], "p99_ms": out[3], "steps": steps,
                     "tracked_fraction": out[1] / (steps * n)}
    res["mode"] = ("shim/gklt_shim.cpp: computePyramid (pageable cv::Mat frame -> staged H2D -> build -> coarsest "
                   "plane back to the host) on a preload std::thread, KLTTrackFeatures (AoS list <-> gklt_track, fp64 "
                   "bearing vectors) on the calling thread; C++ loop, wall clock around `steps` frames")
    return res


def like_for_like(config, pcie, shim):
    """The host-resident figures of the line in one sentence (they lead `config.workload`)."""
    parts = []
    if shim and isinstance(shim.get(config), dict) and "value" in shim[config]:
        parts.append("unchanged-GSLAM drop-in path through the shim (pageable frames) %.1f M feature*frames/s, "
                     "%.3f ms/frame [shim_drop_in]" % (shim[config]["value"] / 1e6, shim[config]["ms_per_frame"]))
    if pcie:
        parts.append("library API with pinned host frames copied inside the timed region %.1f M frame at a time, "
                     "%.1f M with lookahead [pcie_inclusive]" % (
                         pcie["frame_at_a_time"]["value"] / 1e6, pcie.get("lookahead", pcie["frame_at_a_time"])["value"] / 1e6))
    return "; ".join(parts) if parts else "not measured in this run (--main-only / --no-pcie / --no-shim / rank > 0)"


def cpu_model():
    """CPU model name of the box the cpu_baseline leg ran on (core count AND model)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def c5_params(cfg):
    sel = cfg["select"]
    kw = dict(window=cfg["window"], levels=cfg["levels"], min_eigenvalue=sel["min_eigenvalue"],
              mindist=sel["mindist"])
    return kw, sel


def cpu_baseline_c5(cfg, frames, nframes, threads, variant, seconds=CPU_SAMPLE_SECONDS):
    """C5 on the CPU restatement: pyramid, integral images, selection, track -- per frame."""
    from oracle import oracle as ok

    ok.build()
    kw, sel = c5_params(cfg)
    op = ok.make_params(**kw)
    op.bordery = max(op.bordery, sel["min_bordery"])
    ok.set_threads(threads)
    n = cfg["n"]
    cur = ok.compute_pyramid(op, frames[0])
    t0 = time.perf_counter()
    tracked, t = 0, 0
    x = y = val = None
    while t < nframes or time.perf_counter() - t0 < seconds:
        nxt = ok.compute_pyramid(op, frames[(t + 1) % len(frames)])
        sums = ok.integral_products(cur[1], cur[2])
        # the list keeps the size the previous selection grew it to, as the device list does
        x, y, val = ok.select_features(op, 0, variant, *sums, n if val is None else len(val), grid=sel["grid"])
        val = ok.track_features(op, cur, nxt, x, y, val)[2]
        tracked += int((val == 0).sum())
        cur = nxt
        t += 1
    dt = time.perf_counter() - t0
    ok.set_threads(1)
    return tracked / dt, dt, t


def main_c5(a, ctx, cfg, dist, rank, world):
    """--config C5 (BASELINE config 5): the selector runs on the current pyramid EVERY frame
    (SELECT_ALL, Uniform flavour with 32x32-pixel cells by default), then the fresh list is tracked
    to the next frame, whose pyramid was built meanwhile.  Everything stays on the device."""
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    sel = cfg["select"]
    variant = {"uniform": klt.VARIANT_UNIFORM, "plain": klt.VARIANT_PLAIN, "round2": klt.VARIANT_ROUND2}[
        a.select_variant]
    frames = synth.make_stream(H, W, min(a.frames, 16), seed=rank_seed(rank))
    store = FrameStore(ctx, frames)
    ring = [klt.Pyramid(ctx, W, H) for _ in range(3)]
    fl = klt.FeatureList(ctx, 100 * n)  # KLTCreateFeatureList's capacity (KLTUtil.cpp:53-55)
    empty = (np.zeros(n, np.float32), np.zeros(n, np.float32), np.full(n, -1, np.int32))
    fl.upload(*empty)
    state = {"t": 0}

    def run(k):
        for t in range(state["t"], state["t"] + k):
            cur, nxt = ring[t % 3], ring[(t + 1) % 3]
            ptr, stride, _ = store.frame(t + 1)
            nxt.build_from_device(ptr, stride)
            if prepare:   # integral images of frame t+1 behind its build, beside select(t)
                klt.select_prepare(ctx, nxt)
            klt.select_from_pyramid(ctx, cur, klt.SELECT_ALL, variant, fl, grid=sel["grid"])
            klt.track_flist(ctx, cur, nxt, fl)
        state["t"] += k

    def barrier():
        ctx.device_synchronize()
        if dist is not None:
            dist.barrier()

    ptr, stride, _ = store.frame(0)
    ring[0].build_from_device(ptr, stride)
    # how far the two score forms agree ("selection overlap"): one selection of
    # n slots each way on frame 0, outside the timed region
    picks = []
    probe = klt.FeatureList(ctx, 100 * n)
    for mode in (0, 1):
        ctx.set_select_scores(mode)
        probe.upload(*empty)
        klt.select_from_pyramid(ctx, ring[0], klt.SELECT_ALL, variant, probe, grid=sel["grid"])
        px, py_, pv = probe.download()
        picks.append({(x, y) for x, y, v in zip(px, py_, pv) if v > 0})
    probe.close()
    overlap = len(picks[0] & picks[1]) / max(len(picks[0] | picks[1]), 1)
    direct = a.select_scores == "direct"
    ctx.set_select_scores(1 if direct else 0)
    prepare = not a.no_select_prepare and not direct
    run(a.warmup)
    ctx.sync()
    fl.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    ctx.sync()
    ctx.device_synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tracked, iters = fl.stats(reset=True)
    nlist = len(fl)
    dt_max, tracked_all, iters_all = reduce_over_ranks(dist, dt, tracked, iters)

    psteps = min(a.steps, 64)
    ctx.profile_enable(True)
    ctx.profile_reset()
    run(psteps)
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    rounds = ctx.select_rounds()
    p_tracked, p_iters = fl.stats(reset=True)

    lat = []
    for _ in range(min(a.steps, 64)):
        q0 = time.perf_counter()
        run(1)
        ctx.sync()
        lat.append((time.perf_counter() - q0) * 1e3)
    lat = np.array(lat)

    b_pyr, foot, px = algorithmic_bytes(cfg)
    per_kernel = {k: {"avg_us": ms / cnt * 1e3, "launches": cnt, "total_ms": ms, "us_per_frame": ms / psteps * 1e3}
                  for k, (ms, cnt) in prof.items() if cnt}
    # SURVEY 8(d), bytes per frame.  Selector (API form): three (H+1)(W+1) fp32 integral images read
    # + the H x W featuremap read and written.  Integral producer: gx, gy read, three sums written.
    alg_frame = {
        "select": 3.0 * (W + 1) * (H + 1) * 4 + 2.0 * W * H,
        "integral": 8.0 * W * H + 3.0 * (W + 1) * (H + 1) * 4,
        "track": (4.0 * nlist * cfg["levels"] + 3.0 * p_iters / psteps) * foot,
        "level0": W * H * (1 + 12 + 1.0),
    }
    dom = max((k for k in per_kernel if k in alg_frame), key=lambda k: per_kernel[k]["total_ms"])
    achieved = alg_frame[dom] * psteps / (per_kernel[dom]["total_ms"] * 1e-3) / 1e9
    names = {"track": "k_track_w", "level0": "k_level<u8>", "integral": "k_integral_rows/cols",
             "select": "selector kernels (k_score, suppression rounds, rank/assign) of one selection"}
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
        "config": {"workload": "C5: re-detection every frame on the synthetic %dx%d u8 stream: "
                               "gklt_select_from_pyramid (SELECT_ALL, %s, gridSize %dx%d px, min_eigenvalue %d, "
                               "mindist %d, %d features) on the current pyramid, then gklt_track_flist to the "
                               "next frame (%d-level pyramid, %dx%d window), pyramid(t+1) and its integral images built meanwhile; "
                               "frames resident in HBM, one frame per call, 1 stream per GPU" % (
                                   W, H, a.select_variant, sel["grid"][0], sel["grid"][1], sel["min_eigenvalue"],
                                   sel["mindist"], n, cfg["levels"], cfg["window"], cfg["window"]),
                   "frames_per_s": world * a.steps / dt_max,
                   "list_size": nlist,
                   "tracked_fraction": tracked_all / (world * a.steps * max(nlist, 1)),
                   "newton_iters_per_feature_frame": iters_all / (world * a.steps * max(nlist, 1)),
                   "suppression_launches_last_selection": {"phase_a": rounds[0], "phase_b": rounds[1]},
                   "select_scores": a.select_scores,
                   "selection_overlap_direct_vs_integral": {
                       "jaccard": overlap, "selected_integral": len(picks[0]), "selected_direct": len(picks[1]),
                       "what": "one selection on frame 0 with scores from integral images and one with direct "
                               "window sums: |A & B| / |A | B| of the selected positions"},
                   "parallelism": "stream-sharded x%d, no collective" % world},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                       "mode": "build(t+1) + select(t) + track(t -> t+1) enqueued -> both streams drained"},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_frame[dom] * psteps / per_kernel[dom]["launches"],
                     "algorithmic_bytes_per_frame": alg_frame[dom],
                     "avg_launch_us": per_kernel[dom]["avg_us"],
                     "frames_in_profiled_launches": psteps, "profiled_launches": per_kernel[dom]["launches"],
                     "note": "dominant kernel group by total time in the profiled leg; achieved = bytes per "
                             "frame (SURVEY 8d) x frames / total time of that group's launches"},
        "kernels": per_kernel,
    }
    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        v, cdt, ncpu = cpu_baseline_c5(cfg, frames, a.cpu_frames or 4, threads, variant,
                                       0.0 if a.cpu_frames else CPU_SAMPLE_SECONDS)
        out["cpu_baseline"] = {"value": v, "unit": "tracked_features*frames/s", "cores": threads, "kind": "port",
                               "cpu_model": cpu_model(),
                               "sample": "%d frames of the same stream (pyramid + integral images + selection + "
                                         "track), oracle/ C restatement with OpenMP, %.1f s wall" % (ncpu, cdt)}
    for p in ring:
        p.close()
    fl.close()
    store.close()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if "RANK" not in os.environ and a.gpus > 1:
        return launch_ranks(a, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it with --gpus equal to the "
                         "number of ranks" % (a.gpus, world))
    device = 0 if a.share_gpu else local_rank
    if a.dry_launch:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "device": device,
                          "streams": shard_streams(a.streams, world, rank) if a.streams else None}), flush=True)
        if a.fail_rank >= 0:  # launcher test: one rank dies, the others wait as if in a barrier
            if rank == a.fail_rank:
                return 3
            time.sleep(600)
        return 0
    klt.lib()  # libgklt.so and the system HIP runtime, before torch can be imported

    dist = None
    if world > 1:
        import torch.distributed as dist  # after libgklt.so: see the module docstring

        # gloo announces its connections on the C-level stdout ("[Gloo] Rank 0 is connected to ..."),
        # which would precede rank 0's ONE JSON line: fd 1 points at stderr while the group is set up.
        # (No collective here: a torch collective before the library's first HIP call leaves
        # hipGetDeviceCount without devices -- two HIP runtimes in the process, see the module docstring.)
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
        finally:
            ctypes.CDLL(None).fflush(None)  # the notice may sit in libc's stdout buffer
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    cfg = dict(synth.CONFIGS[a.config])
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    params_kw = dict(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
    if "select" in cfg:
        params_kw = c5_params(cfg)[0]
    gp = klt.make_params(**params_kw)
    if "select" in cfg:
        gp.bordery = max(gp.bordery, cfg["select"]["min_bordery"])
    ctx = klt.Context(device, gp)
    ctx.set_track_waves(a.track_waves)
    ctx.set_track_impl(a.track_impl)
    ctx.set_pyramid_impl(a.pyramid_impl)
    if "select" in cfg and not a.streams:
        return main_c5(a, ctx, cfg, dist, rank, world)
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
    pcie = None
    if rank == 0 and not a.main_only and not a.no_pcie:
        pcie = pcie_legs(ctx, cfg, frames, seeds, B, a.steps, a.warmup)

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

    # ---- roofline leg: the same steps, every launch bracketed by HIP events on its own stream.
    # Whole batches only (the leg starts with nothing pending: st.sync() above tracked the last
    # batch), so every profiled launch of a kernel carries the same number of frames and
    # achieved = sum of bytes / sum of kernel time.  A leg of its own because the event pairs cost
    # 5 % of the throughput (210 -> 198 M at C2 with them inside the timed region): `value` is
    # measured without them, the kernel durations with them, over the same number of steps.
    psteps = min(a.steps, 2048)
    if B:
        psteps = max(B, psteps // B * B)
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
        ctx.set_coarse_split(0)           # one build stream: no coarse level of the batch before beside level 0
        st.run(2 * B)
        st.sync()
        ctx.profile_enable(True)
        ctx.profile_reset()
        st.run(8 * B)
        st.sync()
        alone = {k: ms / cnt * 1e3 for k, (ms, cnt) in ctx.profile_get().items() if cnt}
        ctx.profile_enable(False)
        ctx.set_coarse_split(1)

    if not B:
        fa1 = (tracked, dt, a.steps)
        if not a.main_only:
            lat, lat_full = latency_leg(st, min(a.steps, 256))
    elif fa1 is None:
        fa1 = (float("nan"), 1.0, 1)
    tracked1, dt1, k1 = fa1
    lat, lat_full = np.array(lat), np.array(lat_full)

    shim = None
    if rank == 0 and world == 1 and not a.main_only and not a.no_shim:
        shim = shim_drop_in(frames[:16], steps=min(max(a.steps, 64), 512))

    b_pyr, foot, px = algorithmic_bytes(cfg)
    per_kernel = {k: {"avg_us": ms / cnt * 1e3, "launches": cnt, "total_ms": ms}
                  for k, (ms, cnt) in prof.items() if cnt}
    # Algorithmic bytes of the profiled leg = bytes per frame x the psteps frames its
    # launches processed.  Tracker: n features x L levels x (3 template planes + 1 residue plane)
    # + 3 planes per Newton iteration, each plane one (w+1)(h+1) fp32 bilinear footprint.  Level-0
    # kernel: u8 in, (I, gx, gy) out, decimated plane out.
    alg_leg = {
        "track": (4.0 * n * cfg["levels"] * psteps + 3.0 * p_iters) * foot,
        "level0": float(psteps) * W * H * (1 + 12 + 1.0),
    }
    dom = max((k for k in per_kernel if k in alg_leg), key=lambda k: per_kernel[k]["total_ms"])
    launches = per_kernel[dom]["launches"]
    fpl = psteps / launches                       # frames per launch (whole batches: an integer)
    alg = alg_leg[dom] / launches                 # algorithmic bytes per launch
    achieved = alg_leg[dom] / (per_kernel[dom]["total_ms"] * 1e-3) / 1e9
    alg_by_kernel = {k: v / psteps * max(B, 1) for k, v in alg_leg.items()}  # per full batch
    # HBM bytes per frame from the rocprofv3 PMC passes committed under profiles/ (an OFFLINE
    # constant, not collected by this run): only quoted for the configuration it was collected on
    pmc_traffic = PMC_TRAFFIC.get((a.config, dom)) if fpl == 16 else None

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
                               "1 independent stream per GPU.  LIKE-FOR-LIKE with the reference's frame loop "
                               "(frames start in HOST memory, SURVEY 3.1): %s.  `value` = frames RESIDENT IN HBM "
                               "before the timed region, handed over %s; the drop-in call pattern with "
                               "HBM-resident frames is `frame_at_a_time`" % (
                                   a.config, W, H, n, cfg["levels"], cfg["window"], cfg["window"],
                                   like_for_like(a.config, pcie, shim),
                                   ("%d frames per gklt_lookahead_submit call" % B) if B else
                                   "one frame per gklt_pipeline_step call"),
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
                     "frames_per_launch": fpl, "frames_in_profiled_launches": psteps,
                     "profiled_launches": launches,
                     # both kernels of the leg, whichever is the dominant one (at the driver's 20 steps
                     # the tracker launch is, since round 3; in steady state the level-0 launch)
                     "by_kernel": {names[k]: {"avg_launch_us": per_kernel[k]["avg_us"],
                                              "achieved": alg_leg[k] / (per_kernel[k]["total_ms"] * 1e-3) / 1e9,
                                              "frac": alg_leg[k] / (per_kernel[k]["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                   for k in alg_leg if k in per_kernel},
                     "level0_alone": ({"avg_launch_us": alone["level0"],
                                       "achieved": alg_by_kernel["level0"] / (alone["level0"] * 1e-6) / 1e9,
                                       "frac": alg_by_kernel["level0"] / (alone["level0"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       "what": "k_level<u8> with no tracker launch sharing the CUs (empty "
                                               "feature list), same batches"} if "level0" in alone else None),
                     "note": "dominant kernel by total time in the profiled leg; tracker bytes = "
                             "[4*N*L + 3*iters] footprints of (w+1)(h+1)*4 B per frame "
                             "(iters/feature/frame=%.2f), level-0 bytes = W*H*(1+12+1) per frame; "
                             "achieved = bytes of the frames_in_profiled_launches frames / total kernel time "
                             "of the profiled launches (HIP events on the kernel's own stream); "
                             "traffic = offline constant: (2*FETCH_SIZE + WRITE_SIZE) per frame from the "
                             "rocprofv3 --pmc passes in profiles/ (16-frame launches) x frames_per_launch, "
                             "null for other launch sizes; see DESIGN.md 4" % (p_iters / psteps / n)},
        "pcie_inclusive": pcie,
        "shim_drop_in": shim,
        "kernels": per_kernel,
        "pyramid_algorithmic_bytes_per_frame": b_pyr,
    }

    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        secs = CPU_SAMPLE_SECONDS if not a.cpu_frames else 0.0
        v, cdt, nf = cpu_baseline(params_kw, frames, seeds, a.cpu_frames or 8, threads, secs)
        v1, cdt1, nf1 = cpu_baseline(params_kw, frames, seeds, max((a.cpu_frames or 8) // 16, 2), 1, secs / 2)
        out["cpu_baseline"] = {"value": v, "unit": "tracked_features*frames/s", "cores": threads,
                               "kind": "port", "cpu_model": cpu_model(),
                               "one_core": {"value": v1, "sample": "%d frames, %.1f s wall" % (nf1, cdt1)},
                               "sample": "%d frames of the same stream (pyramid + track), oracle/ C "
                                         "restatement with OpenMP, %.1f s wall" % (nf, cdt)}
    st.close()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
