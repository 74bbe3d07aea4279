// gklt_lookahead: the frame loop of System::Track (System.cpp:501-518) for callers that have the
// next frames at hand (recorded video, a capture queue).  The reference preloads ONE pyramid ahead
// on a std::thread (PyramidBuffer, System.h:30-56); here the preload depth is a batch of up to
// GKLT_MAX_LOOKAHEAD frames, because pyramids do not depend on tracking results at all and the
// tracker chain of one feature does not depend on other features:
//   * the pyramids of a batch are built by ONE launch per level (blockIdx.z = frame);
//   * the previous batch is tracked by ONE launch in which every wavefront follows its feature
//     through all frame pairs of the batch (k_track_w, multi-frame), writing the list after each
//     frame into that frame's pinned host buffer.
// Results are those of frame-at-a-time calls, bit for bit.  Per batch the host enqueues
// nlevels + 1 launches and two event waits, independent of the batch size.
//
// Streams (the context's s_pyr / s_trk): P builds, T tracks.  submit(k) enqueues  P: wait(T done k-1); build batch k
//                                                   T: wait(P done k-1); track batch k-1
// so build(k) overlaps track(k-1).  With gklt_ctx_set_coarse_split the build is level 0 on P and the
// coarse levels on a second build stream P2 behind an event: level 0 of batch k+1 follows level 0 of
// batch k on P and overlaps the coarse levels of batch k; "P done" is then recorded on P2.  The ring holds 2*max_batch+1 pyramids: batch k overwrites the
// slots of frames that the tracker launch of submit(k-1) was the last to read.
#include <vector>

#include "gklt_internal.h"

using namespace gklt;

struct gklt_lookahead {
  gklt_ctx *ctx = nullptr;
  int W = 0, H = 0, max_batch = 0;
  std::vector<gklt_pyramid *> ring;
  gklt_flist *fl = nullptr;
  hipStream_t sp = nullptr, st = nullptr;
  hipStream_t sp2 = nullptr;      // coarse levels (gklt_ctx_set_coarse_split), the context's s_pyr2
  hipEvent_t e_level0 = nullptr;  // level 0 of the batch is enqueued up to here on sp
  hipStream_t s_built = nullptr;  // stream the last build ends on (sp or sp2)
  hipEvent_t e_built = nullptr, e_tracked = nullptr;
  bool has_built = false, has_tracked = false;
  // Host frames: staged H2D on a copy stream of their own, two sets of staging buffers taking
  // turns, so that the copies of batch k run while batch k-1 is being built (on stream P they
  // would sit between the two builds: 16 frames of 1080p are 0.9 ms of PCIe time, four builds).
  std::vector<uint8_t *> stage[2];  // per set and batch position
  hipStream_t sc = nullptr;
  hipEvent_t e_copied = nullptr, e_stage_free[2] = {nullptr, nullptr};
  bool stage_used[2] = {false, false};
  unsigned batch_no = 0;
  long built = -1;               // last frame whose pyramid has been enqueued
  long tracked = 0;              // last frame the list has been tracked to (enqueued)
  float *pending_mirror[GKLT_MAX_LOOKAHEAD] = {};  // device views of host_out of pending frames
  void *pending_host[GKLT_MAX_LOOKAHEAD] = {};     // unpinned host_out: copied after the launch
  bool started = false;
};

static gklt_pyramid *slot(gklt_lookahead *p, long frame) {
  return p->ring[(size_t)(frame % (long)p->ring.size())];
}

extern "C" int gklt_lookahead_create(gklt_ctx *ctx, int width, int height, int capacity,
                                     int max_batch, gklt_lookahead **out) {
  if (!ctx || !out || width < 1 || height < 1 || capacity < 1 || max_batch < 1 ||
      max_batch > GKLT_MAX_LOOKAHEAD)
    return GKLT_EINVAL;
  *out = nullptr;
  GKLT_HIP(hipSetDevice(ctx->device));
  gklt_lookahead *p = new gklt_lookahead();
  p->ctx = ctx;
  p->W = width;
  p->H = height;
  p->max_batch = max_batch;
  int rc = 0;
  p->ring.assign((size_t)2 * max_batch + 1, nullptr);
  for (size_t i = 0; i < p->ring.size() && !rc; ++i)
    rc = gklt_pyramid_create(ctx, width, height, &p->ring[i]);
  if (!rc) rc = gklt_flist_create(ctx, capacity, &p->fl);
  // The context's two streams: a process has GPU_MAX_HW_QUEUES (4) hardware queues, HIP streams
  // beyond that share them, and two streams on one queue run their kernels one after the other
  // (seen with rocprofv3 for gklt_multistream when it brought two more streams of its own).
  p->sp = ctx->s_pyr;
  p->st = ctx->s_trk;
  if (!rc && (hipEventCreateWithFlags(&p->e_built, hipEventDisableTiming) != hipSuccess ||
              hipEventCreateWithFlags(&p->e_tracked, hipEventDisableTiming) != hipSuccess)) {
    set_error("lookahead events could not be created");
    rc = GKLT_EHIP;
  }
  if (rc) {
    gklt_lookahead_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

extern "C" int gklt_lookahead_sync(gklt_lookahead *p) {
  if (!p) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(p->ctx->device));
  if (p->sc) GKLT_HIP(hipStreamSynchronize(p->sc));
  if (p->sp) GKLT_HIP(hipStreamSynchronize(p->sp));
  if (p->sp2) GKLT_HIP(hipStreamSynchronize(p->sp2));
  if (p->st) GKLT_HIP(hipStreamSynchronize(p->st));
  return 0;
}

extern "C" int gklt_lookahead_destroy(gklt_lookahead *p) {
  if (!p) return 0;
  (void)hipSetDevice(p->ctx->device);
  if (p->sp) (void)hipStreamSynchronize(p->sp);
  if (p->sp2) (void)hipStreamSynchronize(p->sp2);
  if (p->st) (void)hipStreamSynchronize(p->st);
  if (p->e_level0) (void)hipEventDestroy(p->e_level0);
  for (auto *py : p->ring) gklt_pyramid_destroy(py);
  gklt_flist_destroy(p->fl);
  if (p->sc) (void)hipStreamSynchronize(p->sc);
  for (auto &set : p->stage)
    for (auto *b : set)
      if (b) (void)hipFree(b);
  if (p->e_copied) (void)hipEventDestroy(p->e_copied);
  for (auto e : p->e_stage_free)
    if (e) (void)hipEventDestroy(e);
  if (p->sc) (void)hipStreamDestroy(p->sc);
  if (p->e_built) (void)hipEventDestroy(p->e_built);
  if (p->e_tracked) (void)hipEventDestroy(p->e_tracked);
  delete p;
  return 0;
}

extern "C" gklt_flist *gklt_lookahead_flist(gklt_lookahead *p) { return p ? p->fl : nullptr; }

// build of frames [first, first+n) on stream P
static int enqueue_batch_build(gklt_lookahead *p, long first, const uint8_t *const *frames, int n,
                               int stride, int on_device) {
  gklt_pyramid *py[GKLT_MAX_LOOKAHEAD];
  const uint8_t *src[GKLT_MAX_LOOKAHEAD];
  int src_stride = stride;
  const int set = (int)(p->batch_no++ & 1u);
  if (!on_device) {
    src_stride = p->W;
    if (!p->sc) {
      if (hipStreamCreateWithFlags(&p->sc, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&p->e_copied, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->e_stage_free[0], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->e_stage_free[1], hipEventDisableTiming) != hipSuccess) {
        set_error("copy stream / events of the lookahead pipeline could not be created");
        return GKLT_EHIP;
      }
    }
    auto &st = p->stage[set];
    if (st.size() < (size_t)p->max_batch) st.resize((size_t)p->max_batch, nullptr);
    // this set was last read by the build two batches ago
    if (p->stage_used[set]) GKLT_HIP(hipStreamWaitEvent(p->sc, p->e_stage_free[set], 0));
    for (int i = 0; i < n; ++i) {
      if (!st[i] && hipMalloc(&st[i], (size_t)p->W * p->H) != hipSuccess) {
        set_error("out of device memory for the frame staging buffer");
        return GKLT_ENOMEM;
      }
      GKLT_HIP(copy_plane_h2d(st[i], frames[i], stride, p->W, p->H, p->sc));
      src[i] = st[i];
    }
    GKLT_HIP(hipEventRecord(p->e_copied, p->sc));
    GKLT_HIP(hipStreamWaitEvent(p->sp, p->e_copied, 0));
  }
  for (int i = 0; i < n; ++i) {
    py[i] = slot(p, first + i);
    if (on_device) src[i] = frames[i];
  }
  // Coarse levels on a second stream when the context asks for it: level 0 of the NEXT batch, which
  // follows on sp, then runs beside them instead of behind them (they are short launches that fill
  // the chip less and less: 4080, 1088 and 320 workgroups per 16 C2 frames).
  p->s_built = p->sp;
  // (not with host frames: their copy stream is the process's fourth stream already, a fifth shares a
  // hardware queue with one of the others -- measured: PCIe-inclusive lookahead 69 -> 58 M at C2)
  if (p->ctx->coarse_split && n > 1 && on_device) {
    if (!p->sp2) {
      if (int rc = coarse_stream(p->ctx, &p->sp2)) return rc;
      if (hipEventCreateWithFlags(&p->e_level0, hipEventDisableTiming) != hipSuccess) {
        set_error("lookahead event could not be created");
        return GKLT_EHIP;
      }
    }
    // (the coarse levels overwrite planes the tracker launch before last read: they start behind
    // e_level0, i.e. behind level 0 on sp, which the caller made wait for that launch)
    if (int rc = enqueue_build_batch(p->ctx, p->sp, py, src, n, src_stride, p->sp2, p->e_level0)) return rc;
    p->s_built = p->sp2;
  } else {
    // (a one-stream batch after split ones touches other ring slots than their coarse levels, and
    // the tracker reaches it in stream order behind the launches that waited for those)
    if (int rc = enqueue_build_batch(p->ctx, p->sp, py, src, n, src_stride)) return rc;
  }
  if (!on_device) {
    GKLT_HIP(hipEventRecord(p->e_stage_free[set], p->sp));
    p->stage_used[set] = true;
  }
  for (int i = 0; i < n; ++i) py[i]->has_ready = true;
  return 0;
}

// the list is tracked from frame p->tracked to frame p->built on stream T (one launch)
static int enqueue_pending_track(gklt_lookahead *p) {
  const int nf = (int)(p->built - p->tracked);
  if (nf <= 0) return 0;
  if (p->has_built) GKLT_HIP(hipStreamWaitEvent(p->st, p->e_built, 0));
  const gklt_pyramid *py[GKLT_MAX_LOOKAHEAD + 1];
  for (int j = 0; j <= nf; ++j) py[j] = slot(p, p->tracked + j);
  gklt_flist *fl = p->fl;
  bool copies = false;
  for (int j = 0; j < nf; ++j) copies = copies || p->pending_host[j];
  if (!copies) {
    if (int rc = launch_track_frames(p->ctx, p->st, py, nf, fl->n, fl->x, fl->y, fl->val, nullptr,
                                     nullptr, nullptr, fl->stats, p->pending_mirror, fl->capacity))
      return rc;
  } else {
    // some host_out buffer is not pinned: frame-at-a-time launches with a copy after each
    for (int j = 0; j < nf; ++j) {
      if (int rc = launch_track_frames(p->ctx, p->st, py + j, 1, fl->n, fl->x, fl->y, fl->val,
                                       nullptr, nullptr, nullptr, fl->stats, p->pending_mirror + j,
                                       fl->capacity))
        return rc;
      if (p->pending_host[j])
        GKLT_HIP(hipMemcpyAsync(p->pending_host[j], fl->x, 3 * sizeof(float) * (size_t)fl->capacity,
                                hipMemcpyDeviceToHost, p->st));
    }
  }
  GKLT_HIP(hipEventRecord(p->e_tracked, p->st));
  p->has_tracked = true;
  p->tracked = p->built;
  for (int j = 0; j < GKLT_MAX_LOOKAHEAD; ++j) {
    p->pending_mirror[j] = nullptr;
    p->pending_host[j] = nullptr;
  }
  return 0;
}

extern "C" int gklt_lookahead_start(gklt_lookahead *p, const uint8_t *img0, int stride,
                                    int on_device) {
  if (!p || !img0 || stride < p->W) return GKLT_EINVAL;
  if (int rc = gklt_lookahead_sync(p)) return rc;
  p->has_built = p->has_tracked = false;
  p->built = -1;
  p->tracked = 0;
  if (int rc = enqueue_batch_build(p, 0, &img0, 1, stride, on_device)) return rc;
  if (int rc = gklt_lookahead_sync(p)) return rc;
  p->built = 0;
  p->started = true;
  return 0;
}

extern "C" int gklt_lookahead_submit(gklt_lookahead *p, const uint8_t *const *frames, int n,
                                     int stride, int on_device, void *const *host_out) {
  if (!p || !p->started || !frames || n < 1 || n > p->max_batch || stride < p->W) return GKLT_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!frames[i]) return GKLT_EINVAL;
  gklt_ctx *ctx = p->ctx;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (p->ring[0]->nlevels != ctx->prm.nPyramidLevels || p->ring[0]->ss != ctx->prm.subsampling) {
    set_error("context levels/subsampling changed after the lookahead pipeline was created");
    return GKLT_EINVAL;
  }
  // P: this batch's build.  Its ring slots were last read by the tracker launch of the PREVIOUS
  // call: e_tracked still holds that record, so the wait is enqueued here, before the launch below
  // re-records the event -- the build does not wait for that launch and overlaps it.
  if (p->has_tracked) GKLT_HIP(hipStreamWaitEvent(p->sp, p->e_tracked, 0));
  // T: track the batch submitted by the previous call (waits for e_built's previous record, that
  // batch's build; e_built is re-recorded only after this batch's build has been enqueued).  The
  // tracker launch goes out FIRST: its wavefronts (one per feature, ~64 VGPRs) then hold their four
  // slots per SIMD before the build's larger wavefronts arrive and take what is left; the other way
  // round the build can fill the register files and the tracker, whose serial chains set the
  // pace of the batch, starts late.
  if (int rc = enqueue_pending_track(p)) return rc;
  if (int rc = enqueue_batch_build(p, p->built + 1, frames, n, stride, on_device)) return rc;
  GKLT_HIP(hipEventRecord(p->e_built, p->s_built));
  p->has_built = true;
  for (int i = 0; i < n; ++i) {
    p->pending_mirror[i] = nullptr;
    p->pending_host[i] = nullptr;
    void *h = host_out ? host_out[i] : nullptr;
    if (!h) continue;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, h, 0) == hipSuccess && dp) {
      p->pending_mirror[i] = static_cast<float *>(dp);
    } else {
      (void)hipGetLastError();
      p->pending_host[i] = h;
    }
  }
  p->built += n;
  return 0;
}

extern "C" int gklt_lookahead_flush(gklt_lookahead *p) {
  if (!p || !p->started) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(p->ctx->device));
  if (int rc = enqueue_pending_track(p)) return rc;
  return gklt_lookahead_sync(p);
}

extern "C" int gklt_lookahead_upload_features(gklt_lookahead *p, int n, const float *x,
                                              const float *y, const int *val) {
  if (!p || n < 0 || (n > 0 && (!x || !y || !val))) return GKLT_EINVAL;
  if (n > p->fl->capacity) {
    set_error("%d features exceed the list capacity %d", n, p->fl->capacity);
    return GKLT_ECAPACITY;
  }
  GKLT_HIP(hipSetDevice(p->ctx->device));
  // stream T: after the last tracker launch, before the next one
  if (int rc = flist_upload_staged(p->fl, p->st, 0, n, x, y, val)) return rc;
  p->fl->n = n;
  return 0;
}

extern "C" long gklt_lookahead_frames_tracked(gklt_lookahead *p) { return p ? p->tracked : -1; }
