// gklt_pipeline: PyramidBuffer (System.h:30-56) plus the per-frame call pattern of System::Track
// (System.cpp:501-518) as one object -- a 3-deep ring of device pyramids and a device feature
// list; one step builds the pyramid of the NEXT frame while tracking previous -> current.
//
// Launch structure (sized from tools/hip_overheads.cpp on MI355X / ROCm 7.2: an eager kernel
// launch costs ~2.4 us of host time, a cross-stream event record+wait pair ~7 us, a forked HIP
// graph ~50 us per replay, a 48 KB D2H copy ~12 us on its stream): no graphs, no fork/join.
// Two streams alternate by frame parity.  Step t enqueues
//     stream  t   %2 :  T(t)     tracker previous -> current       (1 launch, enqueued first)
//     stream (t+1)%2 :  B(t+1)   build of the next frame's pyramid (one launch per level)
// so every stream runs ... B(k), T(k), B(k+2), T(k+2) ... in order.  T(t) follows B(t) on its own
// stream and waits (the only event pair of the step) for T(t-1) on the other stream, which it needs
// anyway for the feature list and which also covers B(t-1).  B(t+2) overwrites the ring slot of
// frame t-1; its readers T(t-1) and T(t) are both ordered before it (T(t) on the same stream,
// T(t-1) before T(t)).  The tracker writes the frame's list straight into the caller's pinned
// host buffer, so no copy is queued behind it.
#include <cstring>

#include "gklt_internal.h"

using namespace gklt;

struct gklt_pipeline {
  gklt_ctx *ctx = nullptr;
  int W = 0, H = 0;
  gklt_pyramid *ring[3] = {nullptr, nullptr, nullptr};
  gklt_flist *fl = nullptr;
  hipStream_t st[2] = {nullptr, nullptr};
  uint8_t *d_stage[2] = {nullptr, nullptr};  // per-stream staging of host frames
  hipEvent_t e_trk[2] = {nullptr, nullptr};  // tracker done, per stream
  bool has_trk[2] = {false, false};
  long t = 0;  // index of the "current" frame
  bool started = false;
};

static int slot_of(long frame) { return (int)(frame % 3); }

static int dev(gklt_pipeline *p) {
  GKLT_HIP(hipSetDevice(p->ctx->device));
  return 0;
}

// B(frame) on stream `s_idx`
static int enqueue_frame_build(gklt_pipeline *p, long frame, const uint8_t *img, int stride,
                               int on_device) {
  const int si = (int)(frame & 1);
  hipStream_t s = p->st[si];
  const uint8_t *src = img;
  int src_stride = stride;
  if (!on_device) {
    GKLT_HIP(copy_plane_h2d(p->d_stage[si], img, stride, p->W, p->H, s));
    src = p->d_stage[si];
    src_stride = p->W;
  }
  gklt_pyramid *py = p->ring[slot_of(frame)];
  if (int rc = enqueue_build(p->ctx, s, py, src, src_stride)) return rc;
  py->has_ready = true;
  return 0;
}

extern "C" int gklt_pipeline_create(gklt_ctx *ctx, int width, int height, int capacity,
                                    gklt_pipeline **out) {
  if (!ctx || !out || width < 1 || height < 1 || capacity < 1) return GKLT_EINVAL;
  *out = nullptr;
  GKLT_HIP(hipSetDevice(ctx->device));
  gklt_pipeline *p = new gklt_pipeline();
  p->ctx = ctx;
  p->W = width;
  p->H = height;
  int rc = 0;
  for (int i = 0; i < 3 && !rc; ++i) rc = gklt_pyramid_create(ctx, width, height, &p->ring[i]);
  if (!rc) rc = gklt_flist_create(ctx, capacity, &p->fl);
  for (int i = 0; i < 2 && !rc; ++i) {
    // the context's two streams (see gklt_lookahead_create: streams of its own could end up on
    // one hardware queue)
    p->st[i] = i ? ctx->s_trk : ctx->s_pyr;
    if (hipMalloc(&p->d_stage[i], (size_t)width * height) != hipSuccess) rc = GKLT_ENOMEM;
    else if (hipEventCreateWithFlags(&p->e_trk[i], hipEventDisableTiming) != hipSuccess)
      rc = GKLT_EHIP;
  }
  if (rc) {
    if (rc == GKLT_ENOMEM || rc == GKLT_EHIP) set_error("pipeline resources could not be created");
    gklt_pipeline_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

extern "C" int gklt_pipeline_sync(gklt_pipeline *p) {
  if (!p) return GKLT_EINVAL;
  if (int rc = dev(p)) return rc;
  for (auto s : p->st)
    if (s) GKLT_HIP(hipStreamSynchronize(s));
  return gklt_sync(p->ctx);
}

// Waits until the list of the last step is complete (on the device and, if the step had a
// host_out, in host memory); the build of the next frame keeps running.
extern "C" int gklt_pipeline_wait_tracked(gklt_pipeline *p) {
  if (!p) return GKLT_EINVAL;
  if (int rc = dev(p)) return rc;
  if (!p->started) return 0;
  // the last tracker ran on the stream of frame t-1 (t was advanced by the step) or, for the
  // final track-only step, of frame t: the newer of the two events covers both
  for (int i = 0; i < 2; ++i)
    if (p->has_trk[i]) GKLT_HIP(hipEventSynchronize(p->e_trk[i]));
  return 0;
}

extern "C" int gklt_pipeline_destroy(gklt_pipeline *p) {
  if (!p) return 0;
  (void)hipSetDevice(p->ctx->device);
  for (auto s : p->st)
    if (s) (void)hipStreamSynchronize(s);
  (void)gklt_sync(p->ctx);
  for (auto *py : p->ring) gklt_pyramid_destroy(py);
  gklt_flist_destroy(p->fl);
  for (int i = 0; i < 2; ++i) {
    if (p->d_stage[i]) (void)hipFree(p->d_stage[i]);
    if (p->e_trk[i]) (void)hipEventDestroy(p->e_trk[i]);
  }
  delete p;
  return 0;
}

extern "C" gklt_flist *gklt_pipeline_flist(gklt_pipeline *p) { return p ? p->fl : nullptr; }

extern "C" gklt_pyramid *gklt_pipeline_pyramid(gklt_pipeline *p, int which) {
  if (!p || !p->started || which < 0 || which > 1) return nullptr;
  return p->ring[slot_of(which == 0 ? p->t - 1 + 3 : p->t)];
}

extern "C" int gklt_pipeline_start(gklt_pipeline *p, const uint8_t *img0, const uint8_t *img1,
                                   int stride, int on_device) {
  if (!p || !img0 || !img1 || stride < p->W) return GKLT_EINVAL;
  if (int rc = gklt_pipeline_sync(p)) return rc;
  p->has_trk[0] = p->has_trk[1] = false;
  // System.cpp:367-373: pyramid(frame 0) now, pyramid(frame 1) preloaded
  if (int rc = enqueue_frame_build(p, 0, img0, stride, on_device)) return rc;
  if (int rc = enqueue_frame_build(p, 1, img1, stride, on_device)) return rc;
  if (int rc = gklt_pipeline_sync(p)) return rc;
  p->t = 1;
  p->started = true;
  return 0;
}

extern "C" int gklt_pipeline_step(gklt_pipeline *p, const uint8_t *img_next, int stride,
                                  int on_device, void *host_out) {
  if (!p || !p->started || (img_next && stride < p->W)) return GKLT_EINVAL;
  gklt_ctx *ctx = p->ctx;
  if (int rc = dev(p)) return rc;
  const long t = p->t;
  if (p->ring[0]->nlevels != ctx->prm.nPyramidLevels || p->ring[0]->ss != ctx->prm.subsampling) {
    set_error("context levels/subsampling changed after the pipeline was created");
    return GKLT_EINVAL;
  }
  // T(t): KLTTrackFeatures(pyramid t-1, pyramid t), System.cpp:515-518.  Enqueued BEFORE the
  // build of the next frame: the two are independent, and the 4 launches of the build cost
  // ~16 us of host time that would otherwise delay the frame's result.
  const int si = (int)(t & 1), so = si ^ 1;
  hipStream_t s = p->st[si];
  if (p->has_trk[so]) GKLT_HIP(hipStreamWaitEvent(s, p->e_trk[so], 0));
  float *mirror = nullptr;
  if (host_out) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, host_out, 0) == hipSuccess && dp) mirror = static_cast<float *>(dp);
    else (void)hipGetLastError();  // not pinned: fall back to a copy below
  }
  if (int rc = launch_track(ctx, s, p->ring[slot_of(t - 1 + 3)], p->ring[slot_of(t)], p->fl->n,
                            p->fl->x, p->fl->y, p->fl->val, nullptr, nullptr, nullptr, p->fl->stats,
                            mirror, p->fl->capacity))
    return rc;
  if (host_out && !mirror)
    GKLT_HIP(hipMemcpyAsync(host_out, p->fl->x, 3 * sizeof(float) * (size_t)p->fl->capacity,
                            hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipEventRecord(p->e_trk[si], s));
  p->has_trk[si] = true;
  // B(t+1): the preload of System.cpp:508-511, on the other stream
  if (img_next)
    if (int rc = enqueue_frame_build(p, t + 1, img_next, stride, on_device)) return rc;
  if (img_next) p->t = t + 1;  // PyramidBuffer::next(), System.h:47-55
  return 0;
}

// n consecutive gklt_pipeline_step calls (what a C++ caller's frame loop does), for harnesses whose
// foreign-function calls cost as much as a step's launches
extern "C" int gklt_pipeline_run(gklt_pipeline *p, const uint8_t *const *frames, int n, int stride,
                                 int on_device, void *const *host_out) {
  if (!p || !frames || n < 0) return GKLT_EINVAL;
  for (int i = 0; i < n; ++i)
    if (int rc = gklt_pipeline_step(p, frames[i], stride, on_device, host_out ? host_out[i] : nullptr))
      return rc;
  return 0;
}

// gklt_pipeline_step with the next frame taken from an ingested gklt_frame (gklt_ingest.hip)
extern "C" int gklt_pipeline_step_frame(gklt_pipeline *p, gklt_frame *next, void *host_out) {
  if (!p || !p->started || !next || frame_ctx(next) != p->ctx) return GKLT_EINVAL;
  if (!frame_matches(next, p->W, p->H)) {
    set_error("frame size differs from the pipeline's %dx%d", p->W, p->H);
    return GKLT_EINVAL;
  }
  if (int rc = dev(p)) return rc;
  hipStream_t sb = p->st[(int)((p->t + 1) & 1)];  // the stream B(t+1) is enqueued on
  const uint8_t *luma = nullptr;
  int stride = 0;
  if (int rc = frame_source(next, sb, &luma, &stride)) return rc;
  if (int rc = gklt_pipeline_step(p, luma, stride, 1, host_out)) return rc;
  return frame_consumed(next, sb);
}

extern "C" int gklt_pipeline_upload_features(gklt_pipeline *p, int n, const float *x,
                                             const float *y, const int *val) {
  if (!p || n < 0 || (n > 0 && (!x || !y || !val))) return GKLT_EINVAL;
  if (n > p->fl->capacity) {
    set_error("%d features exceed the list capacity %d", n, p->fl->capacity);
    return GKLT_ECAPACITY;
  }
  if (int rc = dev(p)) return rc;
  // ordered after the last tracker launch and before the next one: same stream as the next T
  const int si = (int)(p->t & 1), so = si ^ 1;
  hipStream_t s = p->st[si];
  if (p->started && p->has_trk[so]) GKLT_HIP(hipStreamWaitEvent(s, p->e_trk[so], 0));
  if (int rc = flist_upload_staged(p->fl, s, 0, n, x, y, val)) return rc;
  p->fl->n = n;
  return 0;
}
