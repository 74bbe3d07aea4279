#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: tracked features * frames / s on the synthetic 1080p stream
(config C2: 4000 features, 4-level pyramid, 7x7 window), one independent stream per GPU.

A step = one frame of one stream: build pyramid(t+1) on the pyramid stream while tracking
(t-1 -> t) on the tracker stream and copying the frame's feature list back to pinned host memory
(the overlap of System.cpp:501-518).  Frames are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|Cship]
N > 1 is launched by torch.distributed.run, one rank per GPU, no data-path collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gslam_amd import klt, synth  # noqa: E402

RESEED_EVERY = 32
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--frames", type=int, default=64, help="distinct synthetic frames (looped)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto)")
    return ap.parse_args()


class Stream:
    """One video stream on one GPU: frames in HBM, 3-deep pyramid ring, device feature list."""

    def __init__(self, ctx, cfg, frames_dev, frame_ptrs, seeds):
        self.ctx, self.cfg = ctx, cfg
        H, W = cfg["H"], cfg["W"]
        self.ring = [klt.Pyramid(ctx, W, H) for _ in range(3)]
        self.fl = klt.FeatureList(ctx, cfg["n"])
        self.frames_dev, self.ptrs = frames_dev, frame_ptrs
        self.seeds = seeds
        self.t = 0
        import torch

        n = cfg["n"]
        # pinned result ring: what the host (System::Track) reads each frame
        self.host = [torch.empty(3 * n, dtype=torch.float32).pin_memory() for _ in range(4)]
        self.host_ptr = [h.data_ptr() for h in self.host]

    def frame_ptr(self, t):
        return self.ptrs[t % len(self.ptrs)]

    def start(self):
        W = self.cfg["W"]
        self.fl.upload(*self.seeds)
        self.ring[0].build_from_device(self.frame_ptr(0), W)
        self.ring[1].build_from_device(self.frame_ptr(1), W)
        self.t = 1

    def step(self):
        """pyramid(t+1) || track(t-1 -> t) -> async D2H of the list."""
        t, W = self.t, self.cfg["W"]
        self.ring[(t + 1) % 3].build_from_device(self.frame_ptr(t + 1), W)
        if t % RESEED_EVERY == 0:  # keep the tracked fraction constant
            klt._check(klt.lib().gklt_flist_upload(self.ctx.h, self.fl.h, self.cfg["n"],
                                                   klt._fp(self.seeds[0]), klt._fp(self.seeds[1]),
                                                   klt._ip(self.seeds[2])))
        klt.track_flist(self.ctx, self.ring[(t - 1) % 3], self.ring[t % 3], self.fl)
        self.fl.download_packed_async(self.host_ptr[t % 4])
        self.t = t + 1

    def close(self):
        self.fl.close()
        for r in self.ring:
            r.close()


def algorithmic_bytes(cfg, levels):
    """compulsory pyramid bytes per frame, and tracker gather bytes per
    (feature, level, pass) unit."""
    H, W = cfg["H"], cfg["W"]
    px = 0
    w, h = W, H
    for _ in range(levels):
        px += w * h
        w //= 2
        h //= 2
    b_pyr = W * H + 12 * px
    foot = (cfg["window"] + 1) ** 2 * 4  # one plane, bilinear footprint (w+1)(h+1) fp32
    return b_pyr, foot, px


def cpu_baseline(cfg, params_kw, frames, seeds, nframes, threads):
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
        cur = ok.compute_pyramid(op, frames[t % len(frames)])
        x, y, val = ok.track_features(op, prev, cur, x, y, val)[:3]
        tracked += int((val == 0).sum())
        prev = cur
    dt = time.perf_counter() - t0
    ok.set_threads(1)
    return tracked / dt, dt


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = dict(synth.CONFIGS[a.config])
    H, W, n = cfg["H"], cfg["W"], cfg["n"]
    params_kw = dict(window=cfg["window"], levels=cfg["levels"], border=cfg.get("border"))
    gp = klt.make_params(**params_kw)
    ctx = klt.Context(local_rank, gp)

    # one independent stream per rank (BASELINE config 4: streams sharded over GPUs)
    seed = synth.BASE_SEED + 1000 * rank
    frames = synth.make_stream(H, W, a.frames, seed=seed)
    sx, sy = synth.lattice_seeds(H, W, n, gp.borderx, seed=seed, nx=cfg["lattice"][0], ny=cfg["lattice"][1])
    seeds = (sx, sy, np.zeros(len(sx), np.int32))
    frames_dev = [torch.from_numpy(f).to(dev) for f in frames]
    ptrs = [f.data_ptr() for f in frames_dev]
    torch.cuda.synchronize()

    st = Stream(ctx, cfg, frames_dev, ptrs, seeds)
    st.start()
    for _ in range(a.warmup):
        st.step()
    ctx.sync()
    st.fl.stats(reset=True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
    ctx.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tracked, iters = st.fl.stats(reset=True)

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cnt = torch.tensor([tracked, iters], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dt_max, tracked_all, iters_all = tt.item(), cnt[0].item(), cnt[1].item()
    else:
        dt_max, tracked_all, iters_all = dt, float(tracked), float(iters)

    # ---- roofline leg: the same steps with every launch bracketed by HIP events on its stream
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(a.steps):
        st.step()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    p_tracked, p_iters = st.fl.stats(reset=True)

    # ---- per-frame latency, synchronous use (what System::Track sees): H2D-free variant
    lat = []
    for _ in range(min(a.steps, 128)):
        q0 = time.perf_counter()
        st.step()
        ctx.sync()
        lat.append((time.perf_counter() - q0) * 1e3)
    lat = np.array(lat)

    b_pyr, foot, px = algorithmic_bytes(cfg, cfg["levels"])
    per_kernel = {k: {"avg_us": (ms / cnt * 1e3 if cnt else 0.0), "launches": cnt,
                      "total_ms": ms} for k, (ms, cnt) in prof.items() if cnt}
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"])
    avg_s = per_kernel[dom]["avg_us"] * 1e-6
    if dom == "track":
        # per launch: n features x L levels x (3 template + 3*iters + 1 residue) plane footprints
        passes = (4.0 * n * cfg["levels"] + 3.0 * p_iters / a.steps)
        alg = passes * foot
        note = "tracker gather bytes, iters/feature/frame=%.2f" % (p_iters / a.steps / n)
    elif dom == "grad":
        alg = 12.0 * px / cfg["levels"]  # I in + gx,gy out per pixel, averaged over the L launches
        note = "gradients: 4 B in + 8 B out per pixel, mean over levels"
    elif dom == "smooth0":
        alg = 5.0 * W * H
        note = "u8 in + f32 out per pixel"
    else:
        alg = (4.0 + 1.0) * (px - (W * H) // (4 ** (cfg["levels"] - 1))) / max(cfg["levels"] - 1, 1)
        note = "f32 in + f32/4 out per source pixel, mean over levels"
    achieved = alg / avg_s / 1e9 if avg_s > 0 else 0.0

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
                   "parallelism": "stream-sharded x%d, no collective" % world},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                       "mode": "synchronous step: pyramid(t+1) || track(t-1->t) + D2H of the list"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "avg_launch_us": per_kernel[dom]["avg_us"],
                     "note": note},
        "kernels": per_kernel,
        "pyramid_algorithmic_bytes_per_frame": b_pyr,
    }

    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        ncpu = a.cpu_frames or (16 if a.config == "C2" else 6)
        v, cdt = cpu_baseline(cfg, params_kw, frames, seeds, ncpu, threads)
        out["cpu_baseline"] = {"value": v, "unit": "tracked_features*frames/s", "cores": threads,
                               "kind": "port",
                               "sample": "%d frames of the same stream (pyramid + track), oracle/ C "
                                         "restatement with OpenMP, %.1f s" % (ncpu, cdt)}
    st.close()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
