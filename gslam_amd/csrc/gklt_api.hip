// C ABI of libgklt.so: contexts, device-resident pyramids and feature lists, stream/event
// ordering between pyramid builds and tracker launches, and per-kernel event timing.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gklt_internal.h"

namespace gklt {

int fix_window(gklt_params *p);  // gklt_host.cpp

// ------------------------------------------------------------------ profiling brackets
// Set by entry points that must not allocate (profiling brackets may create events).
static thread_local bool t_no_profile = false;

ProfScope::ProfScope(gklt_ctx *c, hipStream_t st, int kid) : ctx(c), s(st), id(kid) {
  if (!ctx || !ctx->profiling || t_no_profile) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto take = [&]() {
    hipEvent_t e = nullptr;
    if (!ctx->free_events.empty()) {
      e = ctx->free_events.back();
      ctx->free_events.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
    }
    return e;
  };
  a = take();
  b = take();
  if (a && b) (void)hipEventRecord(a, s);
}

ProfScope::~ProfScope() {
  if (!a || !b) return;
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->spans.push_back({a, b, id});
}

static int fold_spans(gklt_ctx *ctx) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto &sp : ctx->spans) {
    float ms = 0.f;
    GKLT_HIP(hipEventSynchronize(sp.b));
    GKLT_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
    ctx->prof_ms[sp.id] += ms;
    ctx->prof_n[sp.id] += 1;
    ctx->free_events.push_back(sp.a);
    ctx->free_events.push_back(sp.b);
  }
  ctx->spans.clear();
  return 0;
}

static int use_device(gklt_ctx *ctx) {
  GKLT_HIP(hipSetDevice(ctx->device));
  return 0;
}

static int ensure_nv(gklt_ctx *ctx, int n) {
  if (n <= ctx->d_nv_capacity) return 0;
  if (ctx->d_norm) (void)hipFree(ctx->d_norm);
  if (ctx->d_vec) (void)hipFree(ctx->d_vec);
  ctx->d_norm = ctx->d_vec = nullptr;
  ctx->d_nv_capacity = 0;
  GKLT_HIP(hipMalloc(&ctx->d_norm, sizeof(double) * 3 * (size_t)n));
  GKLT_HIP(hipMalloc(&ctx->d_vec, sizeof(double) * 3 * (size_t)n));
  ctx->d_nv_capacity = n;
  return 0;
}

int ensure_bearing_buffers(gklt_ctx *ctx, int n) { return ensure_nv(ctx, n); }

int stage_reserve(HostStage *st, size_t bytes) {
  if (st->pending) {
    GKLT_HIP(hipEventSynchronize(st->done));
    st->pending = false;
  }
  if (st->bytes >= bytes) return 0;
  if (st->p) (void)hipHostFree(st->p);
  st->p = nullptr;
  st->bytes = 0;
  if (!st->done) GKLT_HIP(hipEventCreateWithFlags(&st->done, hipEventDisableTiming));
  const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 16);
  GKLT_HIP(hipHostMalloc(&st->p, want, hipHostMallocDefault));
  st->bytes = want;
  return 0;
}

int stage_mark(HostStage *st, hipStream_t s) {
  GKLT_HIP(hipEventRecord(st->done, s));
  st->pending = true;
  return 0;
}

void stage_free(HostStage *st) {
  if (st->p) (void)hipHostFree(st->p);
  if (st->done) (void)hipEventDestroy(st->done);
  *st = HostStage{};
}

bool host_is_pinned(const void *p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an unregistered pointer is an error on some runtimes: not ours
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

static int check_params(const gklt_params &p) {
  if (p.nPyramidLevels < 1 || p.nPyramidLevels > GKLT_MAX_LEVELS) {
    set_error("nPyramidLevels %d outside [1,%d]", p.nPyramidLevels, GKLT_MAX_LEVELS);
    return GKLT_EINVAL;
  }
  if (p.subsampling < 1) {
    set_error("subsampling %d < 1", p.subsampling);
    return GKLT_EINVAL;
  }
  if (p.window_width > 255 || p.window_height > 255) {
    set_error("window %dx%d too large", p.window_width, p.window_height);
    return GKLT_EINVAL;
  }
  return 0;
}

}  // namespace gklt

using namespace gklt;

// Scratch of gklt_track_noalloc: tracker counters (GKLT_STAT_SHARDS x {tracked, iterations}
// 64-bit words) followed by the staged list x[MAX] | y[MAX] | val[MAX].  One copy per GPU, held
// until the process ends.  In host code the name `scratch` is only a handle for
// hipGetSymbolAddress: kernels and copies get the per-device address kept in gklt_ctx.
namespace {
constexpr size_t kNoallocStatBytes = 2 * GKLT_STAT_SHARDS * sizeof(unsigned long long);
constexpr size_t kNoallocBytes = kNoallocStatBytes + 3 * sizeof(float) * (size_t)GKLT_NOALLOC_MAX_FEATURES;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kNoallocBytes];


// ====================================================================== context
extern "C" int gklt_ctx_create(int device, gklt_ctx **out) {
  if (!out) return GKLT_EINVAL;
  *out = nullptr;
  int ndev = 0;
  GKLT_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    set_error("device %d not present (%d visible): libgklt needs an AMD GPU", device, ndev);
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  GKLT_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libgklt is built for gfx950 only", device, prop.gcnArchName);
    return GKLT_EINVAL;
  }
  gklt_ctx *c = new gklt_ctx();
  c->device = device;
  gklt_default_params(&c->prm);
  const int rc = [&]() -> int {
    // GKLT_CU_SPLIT=k (experiment, round 3): the build stream on the first k compute units of the
    // mask, the tracker stream on the other 256 - k, instead of both streams on the whole chip
    const char *split = std::getenv("GKLT_CU_SPLIT");
    const int k = split ? std::atoi(split) : 0;
    if (k > 0 && k < 256) {
      uint32_t lo[8] = {0}, hi[8] = {0};
      for (int b = 0; b < 256; ++b) (b < k ? lo : hi)[b / 32] |= 1u << (b % 32);
      GKLT_HIP(hipExtStreamCreateWithCUMask(&c->s_pyr, 8, lo));
      GKLT_HIP(hipExtStreamCreateWithCUMask(&c->s_trk, 8, hi));
    } else {
      GKLT_HIP(hipStreamCreateWithFlags(&c->s_pyr, hipStreamNonBlocking));
      GKLT_HIP(hipStreamCreateWithFlags(&c->s_trk, hipStreamNonBlocking));
    }
    GKLT_HIP(hipMalloc(&c->d_invk, 9 * sizeof(double)));
    // this device's copy of the allocation-free entry point's scratch array
    GKLT_HIP(hipGetSymbolAddress(&c->noalloc_scratch, HIP_SYMBOL(scratch)));
    GKLT_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_chain_err), sizeof(unsigned), hipHostMallocDefault));
    *c->h_chain_err = 0;
    if (const char *e = std::getenv("GKLT_COARSE_SPLIT")) c->coarse_split = std::atoi(e) ? 1 : 0;  // A/B runs
    if (const char *e = std::getenv("GKLT_CHAIN")) c->chain_levels = std::min(std::max(std::atoi(e), 0), 2);  // A/B runs
    return 0;
  }();
  if (rc) {  // release what the failed set-up had created (gklt_last_error keeps its message)
    if (c->s_pyr) (void)hipStreamDestroy(c->s_pyr);
    if (c->s_trk) (void)hipStreamDestroy(c->s_trk);
    if (c->d_invk) (void)hipFree(c->d_invk);
    if (c->h_chain_err) (void)hipHostFree(c->h_chain_err);
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" int gklt_ctx_destroy(gklt_ctx *ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->s_pyr);
  if (ctx->s_pyr2) (void)hipStreamSynchronize(ctx->s_pyr2);
  (void)hipStreamSynchronize(ctx->s_trk);
  fold_spans(ctx);
  for (auto e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->scratch_list) gklt_flist_destroy(ctx->scratch_list);
  sel_free(ctx->sel);
  ransac_free(ctx->ransac);
  for (auto &st : ctx->stage_frame) stage_free(&st);
  stage_free(&ctx->stage_plane);
  stage_free(&ctx->stage_trk);
  (void)hipFree(ctx->d_trk);
  if (ctx->h_chain_err) (void)hipHostFree(ctx->h_chain_err);
  (void)hipFree(ctx->d_frame);
  (void)hipFree(ctx->d_invk);
  (void)hipFree(ctx->d_norm);
  (void)hipFree(ctx->d_vec);
  (void)hipFree(ctx->d_rot);
  (void)hipStreamDestroy(ctx->s_pyr);
  if (ctx->s_pyr2) (void)hipStreamDestroy(ctx->s_pyr2);
  (void)hipStreamDestroy(ctx->s_trk);
  delete ctx;
  return 0;
}

extern "C" int gklt_ctx_set_params(gklt_ctx *ctx, const gklt_params *p) {
  if (!ctx || !p) return GKLT_EINVAL;
  gklt_params q = *p;
  fix_window(&q);  // LK.hpp:982-1013 warn-and-fix (the shim prints the KLTWarning)
  if (int rc = check_params(q)) return rc;
  // GSLAM pushes its tracking context on every call, from the preload thread (computePyramid) and
  // from the main thread (KLTTrackFeatures) at the same time (System.cpp:326-344,508-518), with the
  // same values: an unchanged block is not written, so the calls of the other thread, which read
  // ctx->prm, see no write at all.  (Changing parameters while a call is in flight on another
  // thread is a race in the reference as well.)
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (std::memcmp(&ctx->prm, &q, sizeof q) != 0) ctx->prm = q;
  return 0;
}

extern "C" int gklt_ctx_get_params(gklt_ctx *ctx, gklt_params *p) {
  if (!ctx || !p) return GKLT_EINVAL;
  *p = ctx->prm;
  return 0;
}

extern "C" int gklt_sync(gklt_ctx *ctx) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipStreamSynchronize(ctx->s_pyr));
  if (ctx->s_pyr2) GKLT_HIP(hipStreamSynchronize(ctx->s_pyr2));
  GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  if (*ctx->h_chain_err) {
    *ctx->h_chain_err = 0;
    set_error("a coarse-level tile gave up waiting for the level above (k_level_chain): the pyramids "
              "built since the last gklt_sync are incomplete");
    return GKLT_EHIP;
  }
  return 0;
}

// ====================================================================== pyramid
extern "C" int gklt_pyramid_create(gklt_ctx *ctx, int width, int height, gklt_pyramid **out) {
  if (!ctx || !out || width < 1 || height < 1) return GKLT_EINVAL;
  *out = nullptr;
  if (int rc = use_device(ctx)) return rc;
  const gklt_params &P = ctx->prm;
  gklt_pyramid *py = new gklt_pyramid();
  py->ctx = ctx;
  py->nlevels = P.nPyramidLevels;
  py->ss = P.subsampling;
  int w = width, h = height;
  for (int l = 0; l < py->nlevels; ++l) {
    if (w < 1 || h < 1) {
      set_error("level %d of a %dx%d frame is empty", l, width, height);
      gklt_pyramid_destroy(py);
      return GKLT_EINVAL;
    }
    py->lv[l].w = w;
    py->lv[l].h = h;
    if (hipMalloc(&py->lv[l].img, sizeof(float) * (size_t)w * h) != hipSuccess ||
        hipMalloc(&py->lv[l].tex, sizeof(Texel) * (size_t)w * h) != hipSuccess) {
      set_error("hipMalloc of pyramid level %d (%dx%d) failed", l, w, h);
      gklt_pyramid_destroy(py);
      return GKLT_ENOMEM;
    }
    w /= py->ss;  // pyramid.cpp:64
    h /= py->ss;
  }
  if (hipMalloc(&py->tmp, sizeof(float) * (size_t)width * height) != hipSuccess) {
    set_error("hipMalloc of pyramid scratch failed");
    gklt_pyramid_destroy(py);
    return GKLT_ENOMEM;
  }
  {  // one flag per 64 x 32 tile of levels 1 .. L-1 (k_level_chain)
    int tiles = 0;
    for (int l = 1; l < py->nlevels; ++l) tiles += ((py->lv[l].w + 63) / 64) * ((py->lv[l].h + 31) / 32);
    py->chain_tiles = tiles;
    if (tiles > 0 && (hipMalloc(&py->chain_flags, sizeof(unsigned) * tiles) != hipSuccess ||
                      hipMemset(py->chain_flags, 0, sizeof(unsigned) * tiles) != hipSuccess)) {
      set_error("hipMalloc of the pyramid's tile flags failed");
      gklt_pyramid_destroy(py);
      return GKLT_ENOMEM;
    }
  }
  if (hipEventCreateWithFlags(&py->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&py->consumed, hipEventDisableTiming) != hipSuccess) {
    set_error("hipEventCreate for a pyramid failed");
    gklt_pyramid_destroy(py);
    return GKLT_EHIP;
  }
  *out = py;
  return 0;
}

extern "C" int gklt_pyramid_destroy(gklt_pyramid *py) {
  if (!py) return 0;
  (void)hipSetDevice(py->ctx->device);
  (void)hipStreamSynchronize(py->ctx->s_pyr);
  (void)hipStreamSynchronize(py->ctx->s_trk);
  for (int l = 0; l < GKLT_MAX_LEVELS; ++l) {
    if (py->lv[l].img) (void)hipFree(py->lv[l].img);
    if (py->lv[l].tex) (void)hipFree(py->lv[l].tex);
  }
  if (py->tmp) (void)hipFree(py->tmp);
  if (py->chain_flags) (void)hipFree(py->chain_flags);
  if (py->ready) (void)hipEventDestroy(py->ready);
  if (py->consumed) (void)hipEventDestroy(py->consumed);
  delete py;
  return 0;
}

extern "C" int gklt_pyramid_levels(const gklt_pyramid *py) { return py ? py->nlevels : GKLT_EINVAL; }

extern "C" int gklt_pyramid_level_size(const gklt_pyramid *py, int level, int *width, int *height) {
  if (!py || level < 0 || level >= py->nlevels) return GKLT_EINVAL;
  if (width) *width = py->lv[level].w;
  if (height) *height = py->lv[level].h;
  return 0;
}

extern "C" int gklt_pyramid_build(gklt_ctx *ctx, gklt_pyramid *py, const uint8_t *img, int stride,
                                  int img_on_device) {
  if (!ctx || !py || !img || py->ctx != ctx) return GKLT_EINVAL;
  const gklt_params &P = ctx->prm;
  const int W = py->lv[0].w, H = py->lv[0].h;
  if (stride < W) {
    set_error("stride %d < width %d", stride, W);
    return GKLT_EINVAL;
  }
  if (py->nlevels != P.nPyramidLevels || py->ss != P.subsampling) {
    set_error("pyramid was created for %d levels / subsampling %d, context now has %d / %d",
              py->nlevels, py->ss, P.nPyramidLevels, P.subsampling);
    return GKLT_EINVAL;
  }
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_pyr;

  // do not overwrite planes a tracker launch may still be reading
  if (py->has_consumed) GKLT_HIP(hipStreamWaitEvent(s, py->consumed, 0));

  const uint8_t *src = img;
  int src_stride = stride;
  if (!img_on_device) {
    const size_t bytes = (size_t)W * H;
    if (ctx->d_frame_bytes < bytes) {
      GKLT_HIP(hipStreamSynchronize(s));
      if (ctx->d_frame) (void)hipFree(ctx->d_frame);
      ctx->d_frame = nullptr;
      ctx->d_frame_bytes = 0;
      GKLT_HIP(hipMalloc(&ctx->d_frame, bytes));
      ctx->d_frame_bytes = bytes;
    }
    if (host_is_pinned(img)) {
      GKLT_HIP(copy_plane_h2d(ctx->d_frame, img, stride, W, H, s));
    } else {
      // pageable frame (a cv::Mat): through one of two page-locked blocks, so that the caller's
      // frame is free when the call returns and the H2D copy is one DMA transfer
      HostStage *st = &ctx->stage_frame[ctx->stage_frame_next];
      ctx->stage_frame_next ^= 1;
      if (int rc = stage_reserve(st, bytes)) return rc;
      uint8_t *hp = static_cast<uint8_t *>(st->p);
      // (copying in row bands so that a band's transfer overlaps the CPU copy of the next one was
      // measured in round 3 and changes nothing: 0.205 vs 0.208 ms per computePyramid at 1080p --
      // the call is bound by the copy, 0.09 ms, plus the GPU work behind the last byte)
      if (stride == W) {
        std::memcpy(hp, img, bytes);
      } else {
        for (int r = 0; r < H; ++r) std::memcpy(hp + (size_t)r * W, img + (size_t)r * stride, W);
      }
      GKLT_HIP(copy_plane_h2d(ctx->d_frame, hp, W, W, H, s));
      if (int rc = stage_mark(st, s)) return rc;
    }
    src = ctx->d_frame;
    src_stride = W;
  }
  if (int rc = enqueue_build(ctx, s, py, src, src_stride)) return rc;
  GKLT_HIP(hipEventRecord(py->ready, s));
  py->has_ready = true;
  py->built_on_pyr = true;
  return 0;
}

// The launches of one pyramid build, all on stream s; src is a device u8 frame.  No allocation,
// no synchronisation, no events (a cross-stream record+wait pair costs ~7 us of host time on this
// runtime, three kernel launches' worth: tools/hip_overheads.cpp).
int gklt::enqueue_build(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *py, const uint8_t *src,
                        int src_stride) {
  return enqueue_build_batch(ctx, s, &py, &src, 1, src_stride);
}

namespace {
struct BuildTaps {
  gklt::Taps smooth0, pyr, gradg, gradd;
};

int build_taps_uncached(const gklt_params &P, int nlevels, BuildTaps *t) {
  float g[GKLT_MAX_KERNEL_WIDTH], d[GKLT_MAX_KERNEL_WIDTH];
  int gw, dw;
  if (int rc = gklt_compute_kernels(gklt_compute_smooth_sigma(&P), g, &gw, d, &dw)) return rc;
  gklt::half_taps(g, gw, +1, &t->smooth0);
  if (nlevels > 1) {
    if (int rc = gklt_compute_kernels(P.subsampling * P.pyramid_sigma_fact, g, &gw, d, &dw))
      return rc;  // pyramid.cpp:56
    gklt::half_taps(g, gw, +1, &t->pyr);
  }
  if (int rc = gklt_compute_kernels(P.grad_sigma, g, &gw, d, &dw)) return rc;  // pyramid.cpp:69-71
  gklt::half_taps(g, gw, +1, &t->gradg);
  gklt::half_taps(d, dw, -1, &t->gradd);
  return 0;
}

// Kernels of a build.  The reference recomputes them on every call; computeKernels
// costs ~6 us of host time per sigma (71 exp() evaluations and two normalisations), three times
// what a kernel launch costs, so the taps of the last (sigma0, sigma_pyr, sigma_grad, levels)
// are kept per thread.
int build_taps(const gklt_params &P, int nlevels, BuildTaps *t) {
  struct Cache {
    bool valid = false;
    float s0 = 0, sp = 0, sg = 0;
    bool multi = false;
    BuildTaps taps;
  };
  static thread_local Cache cache;
  const float s0 = gklt_compute_smooth_sigma(&P), sp = P.subsampling * P.pyramid_sigma_fact;
  if (cache.valid && cache.s0 == s0 && cache.sp == sp && cache.sg == P.grad_sigma &&
      cache.multi == (nlevels > 1)) {
    *t = cache.taps;
    return 0;
  }
  cache.valid = false;
  if (int rc = build_taps_uncached(P, nlevels, t)) return rc;
  cache.taps = *t;
  cache.s0 = s0;
  cache.sp = sp;
  cache.sg = P.grad_sigma;
  cache.multi = nlevels > 1;
  cache.valid = true;
  return 0;
}

// separate kernels: smooth -> (smooth + decimate)* -> gradients of all levels
int enqueue_build_unfused(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *py, const uint8_t *src,
                          int src_stride, const BuildTaps &t) {
  using namespace gklt;
  const int W = py->lv[0].w, H = py->lv[0].h;
  // pyramid.cpp:59-61  u8 -> f32 -> smooth
  if (int rc = launch_smooth_u8(ctx, s, src, src_stride, W, H, t.smooth0, py->lv[0].img)) return rc;
  for (int l = 1; l < py->nlevels; ++l) {  // pyramid.cpp:62-66
    const Level &a = py->lv[l - 1], &b = py->lv[l];
    if (a.w == 2 * b.w && a.h == 2 * b.h) {
      if (int rc = launch_smooth_decimate2(ctx, s, a.img, a.w, a.h, t.pyr, b.img, b.w, b.h)) return rc;
    } else {
      if (int rc = launch_smooth_f32(ctx, s, a.img, a.w, a.h, t.pyr, py->tmp, GKLT_K_DOWN)) return rc;
      if (int rc = launch_resize_linear(ctx, s, py->tmp, a.w, a.h, b.img, b.w, b.h)) return rc;
    }
  }
  // pyramid.cpp:72-83: gradients of every level in one launch
  return launch_gradients_tex(ctx, s, py->lv, 0, py->nlevels, t.gradg, t.gradd);
}
}  // namespace

// s_coarse / e_level0 (optional, fused builds only): levels 1 .. L-1 go to s_coarse after an event
// recorded behind level 0 on s -- the caller's next batch can then start its level 0 on s while
// this batch's coarse levels, which fill the chip less and less, run beside it.
int gklt::enqueue_build_batch(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py,
                              const uint8_t *const *src, int n, int src_stride,
                              hipStream_t s_coarse, hipEvent_t e_level0) {
  if (n < 1 || n > GKLT_MAX_LOOKAHEAD) return GKLT_EINVAL;
  const gklt_pyramid *p0 = py[0];
  BuildTaps t;
  if (int rc = build_taps(ctx->prm, p0->nlevels, &t)) return rc;
  for (int i = 0; i < n; ++i) py[i]->build_gen = ++ctx->build_counter;

  // One fused launch per level (for the whole batch) when the radii are the instantiated ones
  // (the reference's defaults); level l+1 must be an exact 2x decimation for the fused
  // down-sampling, otherwise that step takes the generic smooth + bilinear resize.
  const bool fuse = ctx->pyr_impl != 1 &&
                    level_fusable(&t.smooth0, p0->nlevels > 1 ? &t.pyr : nullptr, t.gradg, t.gradd);
  if (!fuse) {
    for (int i = 0; i < n; ++i)
      if (int rc = enqueue_build_unfused(ctx, s, py[i], src[i], src_stride, t)) return rc;
    if (s_coarse && e_level0) {  // the caller takes s_coarse as the stream the build ends on
      GKLT_HIP(hipEventRecord(e_level0, s));
      GKLT_HIP(hipStreamWaitEvent(s_coarse, e_level0, 0));
    }
    return 0;
  }
  if (p0->nlevels == 1 && s_coarse && e_level0) {
    if (int rc = enqueue_build_batch(ctx, s, py, src, n, src_stride)) return rc;
    GKLT_HIP(hipEventRecord(e_level0, s));
    GKLT_HIP(hipStreamWaitEvent(s_coarse, e_level0, 0));
    return 0;
  }
  // single frame, default kernels, every level an exact 2x decimation: levels 1 .. L-1 in one launch
  bool chain = (n == 1 ? ctx->chain_levels >= 1 : ctx->chain_levels >= 2) && p0->nlevels >= 3 &&
               (ctx->pyr_impl == 0 || ctx->pyr_impl == 2) && p0->chain_flags;
  for (int l = 0; chain && l + 1 < p0->nlevels; ++l)
    chain = p0->lv[l].w == 2 * p0->lv[l + 1].w && p0->lv[l].h == 2 * p0->lv[l + 1].h;
  for (int l = 0; l < p0->nlevels; ++l) {
    if (l == 1 && s_coarse && e_level0) {
      GKLT_HIP(hipEventRecord(e_level0, s));
      GKLT_HIP(hipStreamWaitEvent(s_coarse, e_level0, 0));
      s = s_coarse;
    }
    if (chain && l == 1) return launch_level_chain(ctx, s, py, n, t.pyr, t.gradg, t.gradd);
    const Level &a = p0->lv[l];
    const bool last = l + 1 == p0->nlevels;
    const bool exact = !last && a.w == 2 * p0->lv[l + 1].w && a.h == 2 * p0->lv[l + 1].h;
    const bool resize = !last && !exact;
    if (l == 0 && resize && n > 1) {
      // the planar level-0 image has to be materialised for the generic resize: frame by frame
      for (int i = 0; i < n; ++i)
        if (int rc = launch_level(ctx, s, py + i, src + i, 1, src_stride, 0, t.smooth0, nullptr,
                                  t.gradg, t.gradd, py[i]->lv[0].img))
          return rc;
    } else {
      if (int rc = launch_level(ctx, s, py, src, n, src_stride, l, t.smooth0,
                                exact ? &t.pyr : nullptr, t.gradg, t.gradd,
                                (l == 0 && resize) ? py[0]->lv[0].img : nullptr))
        return rc;
    }
    if (resize) {
      for (int i = 0; i < n; ++i) {
        const Level &ai = py[i]->lv[l], &bi = py[i]->lv[l + 1];
        if (int rc = launch_smooth_f32(ctx, s, ai.img, ai.w, ai.h, t.pyr, py[i]->tmp, GKLT_K_DOWN))
          return rc;
        if (int rc = launch_resize_linear(ctx, s, py[i]->tmp, ai.w, ai.h, bi.img, bi.w, bi.h))
          return rc;
      }
    }
  }
  return 0;
}

extern "C" int gklt_pyramid_download(gklt_ctx *ctx, gklt_pyramid *py, int level, int plane,
                                     float *dst, int dst_stride) {
  if (!ctx || !py || !dst || level < 0 || level >= py->nlevels || plane < 0 || plane > 2)
    return GKLT_EINVAL;
  const Level &L = py->lv[level];
  if (dst_stride < L.w) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_pyr;
  // every plane comes out of the texels: the fused build never writes level 0's planar image.
  // A pyramid built by gklt_pyramid_build is complete in stream order (same stream); the pipeline
  // objects build on either stream, so their pyramids are waited for on the host.
  if (!py->built_on_pyr) GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  if (int rc = launch_tex_extract(ctx, s, L.tex, L.w * L.h, plane, py->tmp)) return rc;
  const size_t row = sizeof(float) * (size_t)L.w;
  if (host_is_pinned(dst)) {
    GKLT_HIP(hipMemcpy2DAsync(dst, sizeof(float) * dst_stride, py->tmp, row, row, L.h, hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipStreamSynchronize(s));
    return 0;
  }
  // pageable destination: one dense transfer into page-locked memory, rows copied on the host
  HostStage *st = &ctx->stage_plane;
  if (int rc = stage_reserve(st, row * L.h)) return rc;
  GKLT_HIP(hipMemcpyAsync(st->p, py->tmp, row * L.h, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  const char *hp = static_cast<const char *>(st->p);
  for (int r = 0; r < L.h; ++r) std::memcpy(dst + (size_t)r * dst_stride, hp + (size_t)r * row, row);
  return 0;
}

// pyramid[3*level].convertTo(ucharImage, CV_8UC1), System.cpp:403: what the keyframe path reads
extern "C" int gklt_pyramid_download_u8(gklt_ctx *ctx, gklt_pyramid *py, int level, uint8_t *dst,
                                        int dst_stride) {
  if (!ctx || !py || !dst || level < 0 || level >= py->nlevels) return GKLT_EINVAL;
  const Level &L = py->lv[level];
  if (dst_stride < L.w) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_pyr;
  GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  uint8_t *tmp = reinterpret_cast<uint8_t *>(py->tmp);
  if (int rc = launch_tex_to_u8(ctx, s, L.tex, L.w * L.h, tmp)) return rc;
  GKLT_HIP(hipMemcpy2DAsync(dst, dst_stride, tmp, L.w, L.w, L.h, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gklt_pyramid_upload(gklt_ctx *ctx, gklt_pyramid *py, int level, int plane,
                                   const float *src, int src_stride) {
  if (!ctx || !py || !src || level < 0 || level >= py->nlevels || plane < 0 || plane > 2)
    return GKLT_EINVAL;
  const Level &L = py->lv[level];
  if (src_stride < L.w) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_pyr;
  GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  float *stage = plane == 0 ? L.img : py->tmp;
  GKLT_HIP(hipMemcpy2DAsync(stage, sizeof(float) * L.w, src, sizeof(float) * src_stride,
                            sizeof(float) * L.w, L.h, hipMemcpyHostToDevice, s));
  if (int rc = launch_tex_insert(ctx, s, L.tex, L.w * L.h, plane, stage)) return rc;
  py->build_gen = ++ctx->build_counter;
  GKLT_HIP(hipEventRecord(py->ready, s));
  py->has_ready = true;
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;  // (built_on_pyr keeps its value: the other planes are as old as they were)
}

// standalone P1 / P2 on host images
namespace {
struct DevBuf {
  float *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" int gklt_smooth(gklt_ctx *ctx, const float *src, int width, int height, float sigma,
                           float *dst) {
  if (!ctx || !src || !dst || width < 1 || height < 1) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  float g[GKLT_MAX_KERNEL_WIDTH], d[GKLT_MAX_KERNEL_WIDTH];
  int gw, dw;
  if (int rc = gklt_compute_kernels(sigma, g, &gw, d, &dw)) return rc;  // convolve.cpp:50-52
  Taps tg;
  half_taps(g, gw, +1, &tg);
  const size_t bytes = sizeof(float) * (size_t)width * height;
  DevBuf a, b;
  GKLT_HIP(hipMalloc(&a.p, bytes));
  GKLT_HIP(hipMalloc(&b.p, bytes));
  hipStream_t s = ctx->s_pyr;
  GKLT_HIP(hipMemcpyAsync(a.p, src, bytes, hipMemcpyHostToDevice, s));
  if (int rc = launch_smooth_f32(ctx, s, a.p, width, height, tg, b.p, GKLT_K_MISC)) return rc;
  GKLT_HIP(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gklt_gradients(gklt_ctx *ctx, const float *src, int width, int height, float sigma,
                              float *gradx, float *grady) {
  if (!ctx || !src || !gradx || !grady || width < 1 || height < 1) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  float g[GKLT_MAX_KERNEL_WIDTH], d[GKLT_MAX_KERNEL_WIDTH];
  int gw, dw;
  if (int rc = gklt_compute_kernels(sigma, g, &gw, d, &dw)) return rc;  // convolve.cpp:26-28
  Taps tg, td;
  half_taps(g, gw, +1, &tg);
  half_taps(d, dw, -1, &td);
  const size_t bytes = sizeof(float) * (size_t)width * height;
  DevBuf a, bx, by;
  GKLT_HIP(hipMalloc(&a.p, bytes));
  GKLT_HIP(hipMalloc(&bx.p, bytes));
  GKLT_HIP(hipMalloc(&by.p, bytes));
  hipStream_t s = ctx->s_pyr;
  GKLT_HIP(hipMemcpyAsync(a.p, src, bytes, hipMemcpyHostToDevice, s));
  if (int rc = launch_gradients_planar(ctx, s, a.p, width, height, tg, td, bx.p, by.p)) return rc;
  GKLT_HIP(hipMemcpyAsync(gradx, bx.p, bytes, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipMemcpyAsync(grady, by.p, bytes, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

// ====================================================================== feature lists
extern "C" int gklt_flist_create(gklt_ctx *ctx, int capacity, gklt_flist **out) {
  if (!ctx || !out || capacity < 1) return GKLT_EINVAL;
  *out = nullptr;
  if (int rc = use_device(ctx)) return rc;
  gklt_flist *fl = new gklt_flist();
  fl->ctx = ctx;
  fl->capacity = capacity;
  // one allocation [x | y | val], each `capacity` long, so a frame's results leave in one copy
  if (hipMalloc(&fl->x, 3 * sizeof(float) * (size_t)capacity) != hipSuccess ||
      hipMalloc(&fl->stats, 2 * GKLT_STAT_SHARDS * sizeof(unsigned long long)) != hipSuccess) {
    set_error("hipMalloc of a %d-feature list failed", capacity);
    gklt_flist_destroy(fl);
    return GKLT_ENOMEM;
  }
  fl->y = fl->x + capacity;
  fl->val = reinterpret_cast<int *>(fl->x + 2 * (size_t)capacity);
  // Entries beyond the list's size are defined: x = y = 0, val = -1 (0xFF bytes = KLT_NOT_FOUND), so
  // whole-capacity downloads (gklt_flist_download_packed_async, pageable host_out) never carry
  // uninitialised device memory.
  if (hipMemsetAsync(fl->stats, 0, 2 * GKLT_STAT_SHARDS * sizeof(unsigned long long), ctx->s_trk) != hipSuccess ||
      hipMemsetAsync(fl->x, 0, 2 * sizeof(float) * (size_t)capacity, ctx->s_trk) != hipSuccess ||
      hipMemsetAsync(fl->val, 0xFF, sizeof(int) * (size_t)capacity, ctx->s_trk) != hipSuccess) {
    set_error("hipMemsetAsync of a new feature list failed");
    gklt_flist_destroy(fl);
    return GKLT_EHIP;
  }
  *out = fl;
  return 0;
}

extern "C" int gklt_flist_destroy(gklt_flist *fl) {
  if (!fl) return 0;
  (void)hipSetDevice(fl->ctx->device);
  (void)hipDeviceSynchronize();  // staged uploads may be queued on a pipeline's stream
  (void)hipFree(fl->x);
  (void)hipFree(fl->stats);
  if (fl->h_stage) (void)hipHostFree(fl->h_stage);
  if (fl->stage_done) (void)hipEventDestroy(fl->stage_done);
  delete fl;
  return 0;
}

int gklt::flist_upload_staged(gklt_flist *fl, hipStream_t s, size_t off, int n, const float *x,
                              const float *y, const int *val) {
  if (n <= 0) return 0;
  const size_t cap = (size_t)fl->capacity;
  if (!fl->h_stage) {
    GKLT_HIP(hipHostMalloc(reinterpret_cast<void **>(&fl->h_stage), 3 * sizeof(float) * cap, hipHostMallocDefault));
    GKLT_HIP(hipEventCreateWithFlags(&fl->stage_done, hipEventDisableTiming));
  }
  // the mirror has the list's layout, so uploads of disjoint segments (the streams of a
  // gklt_multistream) never wait for each other; a segment is rewritten only after the copies
  // that read it have finished
  const size_t a = off, b = off + (size_t)n;
  bool wait = fl->stage_pending.size() >= 256;
  for (auto &seg : fl->stage_pending) wait = wait || (a < seg.second && seg.first < b);
  if (wait) {
    GKLT_HIP(hipEventSynchronize(fl->stage_done));
    fl->stage_pending.clear();
  }
  float *hx = fl->h_stage + off, *hy = fl->h_stage + cap + off;
  int *hv = reinterpret_cast<int *>(fl->h_stage + 2 * cap) + off;
  std::memcpy(hx, x, sizeof(float) * n);
  std::memcpy(hy, y, sizeof(float) * n);
  std::memcpy(hv, val, sizeof(int) * n);
  GKLT_HIP(hipMemcpyAsync(fl->x + off, hx, sizeof(float) * n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(fl->y + off, hy, sizeof(float) * n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(fl->val + off, hv, sizeof(int) * n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipEventRecord(fl->stage_done, s));
  fl->stage_pending.emplace_back(a, b);
  return 0;
}

extern "C" int gklt_flist_size(const gklt_flist *fl) { return fl ? fl->n : GKLT_EINVAL; }

extern "C" int gklt_flist_upload(gklt_ctx *ctx, gklt_flist *fl, int n, const float *x,
                                 const float *y, const int *val) {
  if (!ctx || !fl || n < 0 || (n > 0 && (!x || !y || !val))) return GKLT_EINVAL;
  if (n > fl->capacity) {
    set_error("%d features exceed the list capacity %d", n, fl->capacity);
    return GKLT_ECAPACITY;
  }
  if (int rc = use_device(ctx)) return rc;
  if (int rc = flist_upload_staged(fl, ctx->s_trk, 0, n, x, y, val)) return rc;
  fl->n = n;
  return 0;
}

extern "C" int gklt_flist_download_async(gklt_ctx *ctx, gklt_flist *fl, float *x, float *y,
                                         int *val) {
  if (!ctx || !fl) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_trk;
  const int n = fl->n;
  if (n == 0) return 0;
  if (x) GKLT_HIP(hipMemcpyAsync(x, fl->x, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  if (y) GKLT_HIP(hipMemcpyAsync(y, fl->y, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  if (val) GKLT_HIP(hipMemcpyAsync(val, fl->val, sizeof(int) * n, hipMemcpyDeviceToHost, s));
  return 0;
}

extern "C" int gklt_flist_download_packed_async(gklt_ctx *ctx, gklt_flist *fl, void *dst) {
  if (!ctx || !fl || !dst) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipMemcpyAsync(dst, fl->x, 3 * sizeof(float) * (size_t)fl->capacity,
                          hipMemcpyDeviceToHost, ctx->s_trk));
  return 0;
}

extern "C" int gklt_flist_download(gklt_ctx *ctx, gklt_flist *fl, float *x, float *y, int *val) {
  if (int rc = gklt_flist_download_async(ctx, fl, x, y, val)) return rc;
  GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  return 0;
}

extern "C" int gklt_flist_stats(gklt_ctx *ctx, gklt_flist *fl, uint64_t *tracked,
                                uint64_t *iterations, int reset) {
  if (!ctx || !fl) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  unsigned long long h[2 * GKLT_STAT_SHARDS];
  GKLT_HIP(hipMemcpyAsync(h, fl->stats, sizeof h, hipMemcpyDeviceToHost, ctx->s_trk));
  if (reset) GKLT_HIP(hipMemsetAsync(fl->stats, 0, sizeof h, ctx->s_trk));
  GKLT_HIP(hipStreamSynchronize(ctx->s_trk));
  unsigned long long t = 0, it = 0;
  for (int i = 0; i < GKLT_STAT_SHARDS; ++i) { t += h[2 * i]; it += h[2 * i + 1]; }
  if (tracked) *tracked = t;
  if (iterations) *iterations = it;
  return 0;
}

// ====================================================================== tracker
static int track_common(gklt_ctx *ctx, gklt_pyramid *p1, gklt_pyramid *p2, int n, float *dx,
                        float *dy, int *dval, const double *d_invk, double *d_norm, double *d_vec,
                        unsigned long long *stats) {
  hipStream_t s = ctx->s_trk;
  if (!p1->has_ready || !p2->has_ready) {
    set_error("tracking on a pyramid that was never built");
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipStreamWaitEvent(s, p1->ready, 0));
  GKLT_HIP(hipStreamWaitEvent(s, p2->ready, 0));
  if (int rc = launch_track(ctx, s, p1, p2, n, dx, dy, dval, d_invk, d_norm, d_vec, stats))
    return rc;
  GKLT_HIP(hipEventRecord(p1->consumed, s));
  p1->has_consumed = true;
  if (p2 != p1) {
    GKLT_HIP(hipEventRecord(p2->consumed, s));
    p2->has_consumed = true;
  }
  return 0;
}

extern "C" int gklt_track_flist(gklt_ctx *ctx, gklt_pyramid *p1, gklt_pyramid *p2,
                                gklt_flist *fl) {
  if (!ctx || !p1 || !p2 || !fl || p1->ctx != ctx || p2->ctx != ctx || fl->ctx != ctx)
    return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  return track_common(ctx, p1, p2, fl->n, fl->x, fl->y, fl->val, nullptr, nullptr, nullptr,
                      fl->stats);
}

extern "C" int gklt_track(gklt_ctx *ctx, gklt_pyramid *p1, gklt_pyramid *p2, int n, float *x,
                          float *y, int *val, const double *invK, double *norm3, double *vec3) {
  if (!ctx || !p1 || !p2 || p1->ctx != ctx || p2->ctx != ctx || n < 0) return GKLT_EINVAL;
  if (n == 0) return 0;
  if (!x || !y || !val) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_trk;
  const bool epi = invK && norm3 && vec3;
  // One page-locked block and its device image, [x | y | val | invK | norm | vec]: the caller's
  // arrays (pageable as a rule: the shim's std::vectors) cost ONE H2D transfer of the first four
  // parts and ONE D2H transfer of the whole block, instead of ten pageable copies.
  const size_t nn = (size_t)n;
  const size_t o_k = (3 * sizeof(float) * nn + 7) / 8 * 8, o_norm = o_k + 9 * sizeof(double);
  const size_t o_vec = o_norm + 3 * sizeof(double) * nn, total = o_vec + 3 * sizeof(double) * nn;
  const size_t down = epi ? total : 3 * sizeof(float) * nn;
  HostStage *st = &ctx->stage_trk;
  if (int rc = stage_reserve(st, total)) return rc;
  if (ctx->d_trk_bytes < total) {
    GKLT_HIP(hipStreamSynchronize(s));
    if (ctx->d_trk) (void)hipFree(ctx->d_trk);
    ctx->d_trk = nullptr;
    ctx->d_trk_bytes = 0;
    const size_t want = total + total / 4;
    GKLT_HIP(hipMalloc(&ctx->d_trk, want));
    ctx->d_trk_bytes = want;
  }
  char *hp = static_cast<char *>(st->p), *dp = static_cast<char *>(ctx->d_trk);
  std::memcpy(hp, x, sizeof(float) * nn);
  std::memcpy(hp + sizeof(float) * nn, y, sizeof(float) * nn);
  std::memcpy(hp + 2 * sizeof(float) * nn, val, sizeof(int) * nn);
  if (epi) std::memcpy(hp + o_k, invK, 9 * sizeof(double));
  GKLT_HIP(hipMemcpyAsync(dp, hp, epi ? o_norm : 3 * sizeof(float) * nn, hipMemcpyHostToDevice, s));
  float *dx = reinterpret_cast<float *>(dp), *dy = dx + nn;
  int *dv = reinterpret_cast<int *>(dy + nn);
  double *d_k = reinterpret_cast<double *>(dp + o_k), *d_norm = reinterpret_cast<double *>(dp + o_norm);
  double *d_vec = reinterpret_cast<double *>(dp + o_vec);
  if (int rc = track_common(ctx, p1, p2, n, dx, dy, dv, epi ? d_k : nullptr, epi ? d_norm : nullptr,
                            epi ? d_vec : nullptr, nullptr))
    return rc;
  GKLT_HIP(hipMemcpyAsync(hp, dp, down, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  std::memcpy(x, hp, sizeof(float) * nn);
  std::memcpy(y, hp + sizeof(float) * nn, sizeof(float) * nn);
  std::memcpy(val, hp + 2 * sizeof(float) * nn, sizeof(int) * nn);
  if (epi) {  // the tracker writes norm / vec of tracked features only (LK.hpp:1019-1035): the
    const double *hn = reinterpret_cast<const double *>(hp + o_norm);  // others keep the caller's
    const double *hv = reinterpret_cast<const double *>(hp + o_vec);
    for (size_t i = 0; i < nn; ++i)
      if (val[i] == GKLT_TRACKED) {
        std::memcpy(norm3 + 3 * i, hn + 3 * i, 3 * sizeof(double));
        std::memcpy(vec3 + 3 * i, hv + 3 * i, 3 * sizeof(double));
      }
  }
  return 0;
}

// KLTTrackFeatures for callers that cannot allocate: the list is staged in the file-scope
// `scratch` array of this GPU instead of a hipMalloc'd list; no hipMalloc / hipFree / event
// creation on this path.
extern "C" int gklt_track_noalloc(gklt_ctx *ctx, gklt_pyramid *p1, gklt_pyramid *p2, int n,
                                  float *x, float *y, int *val, uint64_t *tracked) {
  if (!ctx || !p1 || !p2 || p1->ctx != ctx || p2->ctx != ctx || n < 0) return GKLT_EINVAL;
  if (tracked) *tracked = 0;
  if (n == 0) return 0;
  if (!x || !y || !val) return GKLT_EINVAL;
  if (n > GKLT_NOALLOC_MAX_FEATURES) {
    set_error("gklt_track_noalloc: %d features exceed GKLT_NOALLOC_MAX_FEATURES (%d); use gklt_track",
              n, GKLT_NOALLOC_MAX_FEATURES);
    return GKLT_ECAPACITY;
  }
  if (!ctx->noalloc_scratch) {
    set_error("gklt_track_noalloc: the context holds no scratch address");
    return GKLT_EINVAL;
  }
  if (int rc = use_device(ctx)) return rc;
  hipStream_t s = ctx->s_trk;
  unsigned char *base = static_cast<unsigned char *>(ctx->noalloc_scratch);
  unsigned long long *d_stats = reinterpret_cast<unsigned long long *>(base);
  float *dx = reinterpret_cast<float *>(base + kNoallocStatBytes);
  float *dy = dx + GKLT_NOALLOC_MAX_FEATURES;
  int *dv = reinterpret_cast<int *>(dy + GKLT_NOALLOC_MAX_FEATURES);
  // the array keeps the last call's contents: the counters must start at zero
  GKLT_HIP(hipMemsetAsync(d_stats, 0, kNoallocStatBytes, s));
  GKLT_HIP(hipMemcpyAsync(dx, x, sizeof(float) * n, hipMemcpyDefault, s));
  GKLT_HIP(hipMemcpyAsync(dy, y, sizeof(float) * n, hipMemcpyDefault, s));
  GKLT_HIP(hipMemcpyAsync(dv, val, sizeof(int) * n, hipMemcpyDefault, s));
  t_no_profile = true;
  const int rc = track_common(ctx, p1, p2, n, dx, dy, dv, nullptr, nullptr, nullptr, d_stats);
  t_no_profile = false;
  if (rc) return rc;
  unsigned long long h[2 * GKLT_STAT_SHARDS];
  GKLT_HIP(hipMemcpyAsync(x, dx, sizeof(float) * n, hipMemcpyDefault, s));
  GKLT_HIP(hipMemcpyAsync(y, dy, sizeof(float) * n, hipMemcpyDefault, s));
  GKLT_HIP(hipMemcpyAsync(val, dv, sizeof(int) * n, hipMemcpyDefault, s));
  GKLT_HIP(hipMemcpyAsync(h, d_stats, sizeof h, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  if (tracked) {
    unsigned long long t = 0;
    for (int i = 0; i < GKLT_STAT_SHARDS; ++i) t += h[2 * i];
    *tracked = t;
  }
  return 0;
}

// ====================================================================== measurement
extern "C" int gklt_profile_enable(gklt_ctx *ctx, int on) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = gklt_sync(ctx)) return rc;
  ctx->profiling = on != 0;
  return 0;
}

extern "C" int gklt_profile_get(gklt_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches) {
  if (!ctx || kernel_id < 0 || kernel_id >= GKLT_K_COUNT) return GKLT_EINVAL;
  if (int rc = gklt_sync(ctx)) return rc;
  if (int rc = fold_spans(ctx)) return rc;
  if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
  if (launches) *launches = ctx->prof_n[kernel_id];
  return 0;
}

extern "C" int gklt_profile_reset(gklt_ctx *ctx) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = gklt_sync(ctx)) return rc;
  if (int rc = fold_spans(ctx)) return rc;
  for (int i = 0; i < GKLT_K_COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
  return 0;
}

// ====================================================================== memory helpers
extern "C" int gklt_device_malloc(gklt_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out || bytes == 0) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipMalloc(out, bytes));
  return 0;
}
extern "C" int gklt_device_free(gklt_ctx *ctx, void *ptr) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipFree(ptr));
  return 0;
}
extern "C" int gklt_host_malloc_pinned(gklt_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out || bytes == 0) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return 0;
}
extern "C" int gklt_host_free_pinned(gklt_ctx *ctx, void *ptr) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipHostFree(ptr));
  return 0;
}
extern "C" int gklt_copy_to_device(gklt_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || !dst || !src) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
extern "C" int gklt_device_mem_info(gklt_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}

extern "C" int gklt_device_synchronize(gklt_ctx *ctx) {
  if (!ctx) return GKLT_EINVAL;
  if (int rc = use_device(ctx)) return rc;
  GKLT_HIP(hipDeviceSynchronize());
  return 0;
}

// 0 = one feature per wavefront, LDS footprint staging chosen by window size (default);
// 1 = one feature per 8-lane slice; 2 / 3 = one feature per wavefront with four gathers per
// element / with LDS staging and the element-per-lane mapping (0 picks footprint-in-registers,
// gathers or staging by window size, and the row-run mapping where it stages)
extern "C" int gklt_ctx_set_track_impl(gklt_ctx *ctx, int impl) {
  if (!ctx || impl < 0 || impl > 3) return GKLT_EINVAL;
  ctx->track_impl = impl;
  return 0;
}

// Lookahead pipeline: 1 = the coarse levels of a batch run on a second build stream, so that level 0
// of the next batch overlaps them; 0 = one build stream.
extern "C" int gklt_ctx_set_coarse_split(gklt_ctx *ctx, int on) {
  if (!ctx) return GKLT_EINVAL;
  ctx->coarse_split = on ? 1 : 0;
  return 0;
}

// the second build stream, created on first use (a process has few hardware queues: contexts that
// never split their builds do not take one)
int gklt::coarse_stream(gklt_ctx *ctx, hipStream_t *out) {
  if (!ctx->s_pyr2) GKLT_HIP(hipStreamCreateWithFlags(&ctx->s_pyr2, hipStreamNonBlocking));
  *out = ctx->s_pyr2;
  return 0;
}

// wavefronts of a wave-per-feature tracker launch (0 = default cap of 4096)
extern "C" int gklt_ctx_set_track_waves(gklt_ctx *ctx, int waves) {
  if (!ctx || waves < 0) return GKLT_EINVAL;
  ctx->track_waves = waves;
  return 0;
}

// 0 = one fused launch per level when the radii allow it (default); 1 = separate kernels; 3 = the
// streaming form of the fused kernel; 4 = the fused kernel with mul+add contracted into FMAs (NOT
// bit-identical: inside north_star's 1e-4 relative tolerance; never the default)
extern "C" int gklt_ctx_set_chain(gklt_ctx *ctx, int on) {
  if (!ctx) return GKLT_EINVAL;
  ctx->chain_levels = on < 0 ? 0 : (on > 2 ? 2 : on);
  return 0;
}

extern "C" int gklt_ctx_set_pyramid_impl(gklt_ctx *ctx, int impl) {
  if (!ctx || impl < 0 || impl > 4) return GKLT_EINVAL;
  ctx->pyr_impl = impl;
  return 0;
}
