#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: tracked features * frames / s on the synthetic 1080p stream
(config C2: 4000 features, 4-level pyramid, 7x7 window), one independent stream per GPU.

A step = one frame of one stream through gklt_pipeline_step: build pyramid(t+1) while tracking
(t-1 -> t), the tracker writing the frame's feature list into pinned host memory (the overlap of
System.cpp:501-518).  Frames are resident in HBM before the timed region.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--frames", type=int, default=64, help="distinct synthetic frames (looped)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
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

    def step(self):
        """pyramid(t+1) || track(t-1 -> t) -> async D2H of the list."""
        t = self.t
        if t % RESEED_EVERY == 0:  # keep the tracked fraction constant
            self.pipe.upload_features(*self.seeds)
        self.pipe.step(self.frame(t + 1), self.host[t % 4][1])
        self.t = t + 1

    def sync(self):
        self.pipe.sync()

    def close(self):
        self.pipe.close()
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
    ctx = klt.Context(local_rank, gp)

    seed = rank_seed(rank)
    frames = synth.make_stream(H, W, a.frames, seed=seed)
    sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=seed, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    seeds = (sx, sy, np.zeros(len(sx), np.int32))

    st = Stream(ctx, cfg, frames, seeds)
    st.start()
    for _ in range(a.warmup):
        st.step()
    st.sync()
    st.fl.stats(reset=True)

    def barrier():
        ctx.device_synchronize()
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
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
    for _ in range(psteps):
        st.step()
    st.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    p_tracked, p_iters = st.fl.stats(reset=True)

    # ---- per-frame latency, synchronous use (what System::Track sees)
    lat = []
    for _ in range(min(a.steps, 256)):
        q0 = time.perf_counter()
        st.step()
        st.sync()
        lat.append((time.perf_counter() - q0) * 1e3)
    lat = np.array(lat)

    b_pyr, foot, px = algorithmic_bytes(cfg)
    per_kernel = {k: {"avg_us": ms / cnt * 1e3, "launches": cnt, "total_ms": ms}
                  for k, (ms, cnt) in prof.items() if cnt}
    dom = "track"  # the largest single launch of the pipeline (rocprofv3 summary in profiles/)
    avg_s = per_kernel[dom]["avg_us"] * 1e-6
    # per launch: n features x L levels x (3 template planes + 1 residue plane) + 3 planes per
    # Newton iteration, each plane one (w+1)(h+1) fp32 bilinear footprint  
    passes = 4.0 * n * cfg["levels"] + 3.0 * p_iters / psteps
    alg = passes * foot
    achieved = alg / avg_s / 1e9

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
                   "launch": "eager, 2 alternating streams, 1 event wait/frame",
                   "parallelism": "stream-sharded x%d, no collective" % world},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                       "mode": "synchronous step: pyramid(t+1) || track(t-1->t) + D2H of the list"},
        "roofline": {"bound": "hbm", "kernel": "k_track", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_us": per_kernel[dom]["avg_us"],
                     "note": "tracker gather bytes [4*N*L + 3*iters] footprints of (w+1)(h+1)*4 B; "
                             "iters/feature/frame=%.2f; latency-bound gathers served by L2/MALL, "
                             "see DESIGN.md 4.3" % (p_iters / psteps / n)},
        "kernels": per_kernel,
        "pyramid_algorithmic_bytes_per_frame": b_pyr,
    }

    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        ncpu = a.cpu_frames or (160 if a.config == "C2" else 24)
        v, cdt = cpu_baseline(params_kw, frames, seeds, ncpu, threads)
        out["cpu_baseline"] = {"value": v, "unit": "tracked_features*frames/s", "cores": threads,
                               "kind": "port",
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
