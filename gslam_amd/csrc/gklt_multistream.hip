// gklt_multistream: one GPU's share of BASELINE config 4 -- several independent video streams,
// each with the state of a gklt_pipeline (PyramidBuffer ring, System.h:30-56, and a device feature
// list), advanced one frame per call in the pattern of System::Track (System.cpp:501-518).
//
// Frames of one stream are serially dependent, streams are not, so a step's work is
// batched ACROSS streams: the next frames' pyramids take one launch per level for up to
// GKLT_MAX_LOOKAHEAD streams (blockIdx.z = stream), and all lists are tracked by one launch of the
// wave-per-feature kernel in its multi-stream mode.  Compared with one gklt_pipeline per stream
// (5 launches + an event pair per frame, each tracker launch as long as its slowest feature) a
// step of S streams costs 4*ceil(S/16) + 1 launches and two event waits, and the tracker's
// wavefronts average over S*N features.
//
// Streams: P builds, T tracks (the context's s_pyr and s_trk).
//   step(t) enqueues   P: wait(T of step t-1); build frames t+1
//                      T: wait(P of step t-1); track t-1 -> t
// so build(t+1) overlaps track(t).  build(t+1) overwrites ring slot (t+1)%3 = (t-2)%3, last read
// by the tracker launch of step t-1.
#include <vector>

#include "gklt_internal.h"

using namespace gklt;

struct gklt_multistream {
  gklt_ctx *ctx = nullptr;
  int S = 0, W = 0, H = 0, cap = 0;
  std::vector<gklt_pyramid *> ring;  // [S][3]
  gklt_flist *fl = nullptr;          // capacity S*cap: stream s owns [s*cap, s*cap + n[s])
  int n[GKLT_MAX_STREAMS] = {};
  const Texel **d_tab = nullptr;     // device: [S][3][GKLT_MAX_LEVELS]
  hipStream_t sp = nullptr, st = nullptr;
  hipEvent_t e_built = nullptr, e_tracked = nullptr;
  bool has_built = false, has_tracked = false;
  std::vector<uint8_t *> stage;  // per stream: staging of host frames
  long t = 0;                    // index of the "current" frame
  bool started = false;
};

static gklt_pyramid *slot(gklt_multistream *m, int s, long frame) {
  return m->ring[(size_t)s * 3 + (size_t)(frame % 3)];
}

extern "C" int gklt_multistream_sync(gklt_multistream *m) {
  if (!m) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(m->ctx->device));
  if (m->sp) GKLT_HIP(hipStreamSynchronize(m->sp));
  if (m->st) GKLT_HIP(hipStreamSynchronize(m->st));
  return 0;
}

extern "C" int gklt_multistream_wait_tracked(gklt_multistream *m) {
  if (!m) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(m->ctx->device));
  if (m->has_tracked) GKLT_HIP(hipEventSynchronize(m->e_tracked));
  return 0;
}

extern "C" int gklt_multistream_destroy(gklt_multistream *m) {
  if (!m) return 0;
  (void)hipSetDevice(m->ctx->device);
  if (m->sp) (void)hipStreamSynchronize(m->sp);
  if (m->st) (void)hipStreamSynchronize(m->st);
  for (auto *py : m->ring) gklt_pyramid_destroy(py);
  gklt_flist_destroy(m->fl);
  for (auto *b : m->stage)
    if (b) (void)hipFree(b);
  if (m->d_tab) (void)hipFree(m->d_tab);
  if (m->e_built) (void)hipEventDestroy(m->e_built);
  if (m->e_tracked) (void)hipEventDestroy(m->e_tracked);
  delete m;
  return 0;
}

extern "C" int gklt_multistream_create(gklt_ctx *ctx, int nstreams, int width, int height,
                                       int capacity, gklt_multistream **out) {
  if (!ctx || !out || nstreams < 1 || nstreams > GKLT_MAX_STREAMS || width < 1 || height < 1 ||
      capacity < 1 || (long)nstreams * capacity > (1L << 30))
    return GKLT_EINVAL;
  *out = nullptr;
  GKLT_HIP(hipSetDevice(ctx->device));
  gklt_multistream *m = new gklt_multistream();
  m->ctx = ctx;
  m->S = nstreams;
  m->W = width;
  m->H = height;
  m->cap = capacity;
  int rc = 0;
  m->ring.assign((size_t)3 * nstreams, nullptr);
  for (size_t i = 0; i < m->ring.size() && !rc; ++i)
    rc = gklt_pyramid_create(ctx, width, height, &m->ring[i]);
  if (!rc) rc = gklt_flist_create(ctx, nstreams * capacity, &m->fl);
  if (!rc) {
    // the texel planes of the ring never move: the table is written once
    std::vector<const Texel *> tab((size_t)3 * nstreams * GKLT_MAX_LEVELS, nullptr);
    for (size_t i = 0; i < m->ring.size(); ++i)
      for (int l = 0; l < m->ring[i]->nlevels; ++l) tab[i * GKLT_MAX_LEVELS + l] = m->ring[i]->lv[l].tex;
    const size_t bytes = tab.size() * sizeof(tab[0]);
    if (hipMalloc(reinterpret_cast<void **>(&m->d_tab), bytes) != hipSuccess) {
      set_error("out of device memory for the multi-stream pyramid table");
      rc = GKLT_ENOMEM;
    } else if (hipMemcpy(m->d_tab, tab.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("multi-stream pyramid table could not be uploaded");
      rc = GKLT_EHIP;
    }
  }
  // The context's two streams, not streams of its own: a process has GPU_MAX_HW_QUEUES (4)
  // hardware queues, further HIP streams share them, and two streams that land on the same queue
  // run their kernels one after the other (seen with rocprofv3: builds and tracker of a step on
  // one queue, no overlap, when the object created a fifth and sixth stream).
  m->sp = ctx->s_pyr;
  m->st = ctx->s_trk;
  if (!rc && (hipEventCreateWithFlags(&m->e_built, hipEventDisableTiming) != hipSuccess ||
              hipEventCreateWithFlags(&m->e_tracked, hipEventDisableTiming) != hipSuccess)) {
    set_error("multi-stream events could not be created");
    rc = GKLT_EHIP;
  }
  if (rc) {
    gklt_multistream_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

// pyramids of frame `frame` of every stream, on stream P
static int enqueue_builds(gklt_multistream *m, long frame, const uint8_t *const *imgs, int stride,
                          int on_device) {
  int src_stride = stride;
  if (!on_device) {
    if (m->stage.empty()) m->stage.assign((size_t)m->S, nullptr);
    src_stride = m->W;
  }
  for (int s0 = 0; s0 < m->S; s0 += GKLT_MAX_LOOKAHEAD) {
    const int nb = m->S - s0 < GKLT_MAX_LOOKAHEAD ? m->S - s0 : GKLT_MAX_LOOKAHEAD;
    gklt_pyramid *py[GKLT_MAX_LOOKAHEAD];
    const uint8_t *src[GKLT_MAX_LOOKAHEAD];
    for (int i = 0; i < nb; ++i) {
      const int s = s0 + i;
      py[i] = slot(m, s, frame);
      if (on_device) {
        src[i] = imgs[s];
      } else {
        if (!m->stage[s] && hipMalloc(&m->stage[s], (size_t)m->W * m->H) != hipSuccess) {
          set_error("out of device memory for the frame staging buffers");
          return GKLT_ENOMEM;
        }
        GKLT_HIP(copy_plane_h2d(m->stage[s], imgs[s], stride, m->W, m->H, m->sp));
        src[i] = m->stage[s];
      }
    }
    if (int rc = enqueue_build_batch(m->ctx, m->sp, py, src, nb, src_stride)) return rc;
    for (int i = 0; i < nb; ++i) py[i]->has_ready = true;
  }
  return 0;
}

static int check_imgs(const gklt_multistream *m, const uint8_t *const *imgs, int stride) {
  if (!imgs || stride < m->W) return GKLT_EINVAL;
  for (int s = 0; s < m->S; ++s)
    if (!imgs[s]) return GKLT_EINVAL;
  return 0;
}

extern "C" int gklt_multistream_start(gklt_multistream *m, const uint8_t *const *imgs0,
                                      const uint8_t *const *imgs1, int stride, int on_device) {
  if (!m) return GKLT_EINVAL;
  if (int rc = check_imgs(m, imgs0, stride)) return rc;
  if (int rc = check_imgs(m, imgs1, stride)) return rc;
  if (int rc = gklt_multistream_sync(m)) return rc;
  m->has_built = m->has_tracked = false;
  // System.cpp:367-373: pyramid(frame 0) now, pyramid(frame 1) preloaded
  if (int rc = enqueue_builds(m, 0, imgs0, stride, on_device)) return rc;
  if (!on_device)
    if (int rc = gklt_multistream_sync(m)) return rc;  // the staging buffers are reused
  if (int rc = enqueue_builds(m, 1, imgs1, stride, on_device)) return rc;
  if (int rc = gklt_multistream_sync(m)) return rc;
  m->t = 1;
  m->started = true;
  return 0;
}

extern "C" int gklt_multistream_step(gklt_multistream *m, const uint8_t *const *imgs_next,
                                     int stride, int on_device, void *host_out) {
  if (!m || !m->started) return GKLT_EINVAL;
  if (imgs_next)
    if (int rc = check_imgs(m, imgs_next, stride)) return rc;
  gklt_ctx *ctx = m->ctx;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (m->ring[0]->nlevels != ctx->prm.nPyramidLevels || m->ring[0]->ss != ctx->prm.subsampling) {
    set_error("context levels/subsampling changed after the multi-stream pipeline was created");
    return GKLT_EINVAL;
  }
  const long t = m->t;
  // P waits for the tracker launch of the PREVIOUS step (the event still holds that record: the
  // wait is enqueued before the re-record below), so the build overlaps the launch enqueued next.
  if (imgs_next && m->has_tracked) GKLT_HIP(hipStreamWaitEvent(m->sp, m->e_tracked, 0));
  // T(t): KLTTrackFeatures(pyramid t-1, pyramid t) of every stream, System.cpp:515-518; enqueued
  // before the builds so that their host-side launch cost does not delay the lists
  if (m->has_built) GKLT_HIP(hipStreamWaitEvent(m->st, m->e_built, 0));
  float *mirror = nullptr;
  if (host_out) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, host_out, 0) == hipSuccess && dp) mirror = static_cast<float *>(dp);
    else (void)hipGetLastError();  // not pinned: a copy below
  }
  gklt_flist *fl = m->fl;
  if (int rc = launch_track_streams(ctx, m->st, m->d_tab, m->S, (int)((t + 2) % 3), (int)(t % 3),
                                    m->cap, m->n, m->ring[0], fl->x, fl->y, fl->val, fl->stats,
                                    mirror, fl->capacity))
    return rc;
  if (host_out && !mirror)
    GKLT_HIP(hipMemcpyAsync(host_out, fl->x, 3 * sizeof(float) * (size_t)fl->capacity,
                            hipMemcpyDeviceToHost, m->st));
  GKLT_HIP(hipEventRecord(m->e_tracked, m->st));
  m->has_tracked = true;
  if (imgs_next) {  // B(t+1): the preload of System.cpp:508-511
    if (int rc = enqueue_builds(m, t + 1, imgs_next, stride, on_device)) return rc;
    GKLT_HIP(hipEventRecord(m->e_built, m->sp));
    m->has_built = true;
    m->t = t + 1;  // PyramidBuffer::next(), System.h:47-55
  }
  return 0;
}

extern "C" int gklt_multistream_upload_features(gklt_multistream *m, int stream, int n,
                                                const float *x, const float *y, const int *val) {
  if (!m || stream < 0 || stream >= m->S || n < 0 || (n > 0 && (!x || !y || !val)))
    return GKLT_EINVAL;
  if (n > m->cap) {
    set_error("%d features exceed the list capacity %d", n, m->cap);
    return GKLT_ECAPACITY;
  }
  GKLT_HIP(hipSetDevice(m->ctx->device));
  // stream T: after the last tracker launch, before the next one
  const size_t o = (size_t)stream * m->cap;
  if (int rc = flist_upload_staged(m->fl, m->st, o, n, x, y, val)) return rc;
  m->n[stream] = n;
  return 0;
}

extern "C" int gklt_multistream_size(const gklt_multistream *m, int stream) {
  return (!m || stream < 0 || stream >= m->S) ? GKLT_EINVAL : m->n[stream];
}

extern "C" int gklt_multistream_download(gklt_multistream *m, int stream, float *x, float *y,
                                         int *val) {
  if (!m || stream < 0 || stream >= m->S) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(m->ctx->device));
  const size_t o = (size_t)stream * m->cap;
  const int n = m->n[stream];
  if (n > 0) {
    if (x) GKLT_HIP(hipMemcpyAsync(x, m->fl->x + o, sizeof(float) * n, hipMemcpyDeviceToHost, m->st));
    if (y) GKLT_HIP(hipMemcpyAsync(y, m->fl->y + o, sizeof(float) * n, hipMemcpyDeviceToHost, m->st));
    if (val) GKLT_HIP(hipMemcpyAsync(val, m->fl->val + o, sizeof(int) * n, hipMemcpyDeviceToHost, m->st));
  }
  GKLT_HIP(hipStreamSynchronize(m->st));
  return 0;
}

extern "C" int gklt_multistream_stats(gklt_multistream *m, uint64_t *tracked, uint64_t *iterations,
                                      int reset) {
  if (!m) return GKLT_EINVAL;
  if (int rc = gklt_multistream_sync(m)) return rc;
  return gklt_flist_stats(m->ctx, m->fl, tracked, iterations, reset);
}

extern "C" gklt_pyramid *gklt_multistream_pyramid(gklt_multistream *m, int stream, int which) {
  if (!m || !m->started || stream < 0 || stream >= m->S || which < 0 || which > 1) return nullptr;
  return slot(m, stream, which == 0 ? m->t + 2 : m->t);
}
