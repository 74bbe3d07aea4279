// Internal declarations shared by the translation units of libgklt.so.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gklt.h"
#include "../../include/gklt_order.h"

namespace gklt {

void set_error(const char *fmt, ...);
#define GKLT_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      gklt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,     \
                      __LINE__);                                                           \
      return GKLT_EHIP;                                                                    \
    }                                                                                      \
  } while (0)

// Half of a symmetric (sym=+1) or antisymmetric (sym=-1) convolution kernel:
// k[0] = centre tap, k[j] = tap at offset +j (tap at -j is sym*k[j]).
constexpr int kMaxRadius = GKLT_MAX_KERNEL_WIDTH / 2;  // 35
#define GKLT_STAT_SHARDS 256  // tracker counters are sharded over this many (tracked, iters) pairs
struct Taps {
  int r;
  int sym;
  float k[kMaxRadius + 1];
};

// One pixel of a pyramid level as the tracker reads it: (I, gx, gy), 12 bytes, so that one
// bilinear tap is one global_load_dwordx3 that brings all three planes.  (A 16-byte texel with a
// zero pad costs a quarter more HBM write traffic in the level kernels and L2 space in the tracker.)
struct Texel {
  float x, y, z;
};

struct Level {
  int w = 0, h = 0;
  float *img = nullptr;  // planar I, w*h
  Texel *tex = nullptr;  // interleaved (I, gx, gy), w*h
};

}  // namespace gklt

namespace gklt {
struct SelScratch;  // selector scratch (gklt_select.hip)
void sel_free(SelScratch *s);
struct RansacScratch;  // outlier-rejection scratch (gklt_ransac.hip)
void ransac_free(RansacScratch *r);
}  // namespace gklt

namespace gklt {
// A page-locked staging block that ONE host thread / stream uses: host arrays of unknown provenance
// (pageable cv::Mat frames, std::vector lists) are copied through it, because the runtime moves
// pageable memory in small synchronous pieces (a 2-D copy of a pageable 1080p frame: 10 ms, row by
// row; the same bytes from page-locked memory: 55 us).  `done` marks the last copy that read or
// wrote the block.
struct HostStage {
  void *p = nullptr;
  size_t bytes = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
};
int stage_reserve(HostStage *st, size_t bytes);  // waits for the pending copy, grows the block
int stage_mark(HostStage *st, hipStream_t s);    // records `done` behind the copies just enqueued
void stage_free(HostStage *st);
bool host_is_pinned(const void *p);              // page-locked / registered host memory?
}  // namespace gklt

struct gklt_ctx {
  int device = 0;
  gklt::SelScratch *sel = nullptr;
  gklt::RansacScratch *ransac = nullptr;
  gklt_params prm{};
  hipStream_t s_pyr = nullptr;  // pyramid builds
  hipStream_t s_pyr2 = nullptr; // coarse levels of batched builds (lookahead pipeline), created on first use
  int coarse_split = 1;         // gklt_ctx_set_coarse_split
  hipStream_t s_trk = nullptr;  // tracker, feature list copies
  // staging for host frames / host feature lists (grown on demand)
  uint8_t *d_frame = nullptr;
  size_t d_frame_bytes = 0;
  gklt::HostStage stage_frame[2];  // pageable frames on their way to d_frame (builds, s_pyr's thread)
  int stage_frame_next = 0;
  gklt::HostStage stage_plane;     // plane downloads (s_pyr's thread)
  gklt::HostStage stage_trk;       // gklt_track's list, invK and bearing vectors (s_trk's thread)
  std::atomic<unsigned long long> build_counter{0};  // source of gklt_pyramid::build_gen (unique per context)
  unsigned *h_chain_err = nullptr;  // page-locked word a k_level_chain tile raises when it gives up waiting
  int chain_levels = 1;             // gklt_ctx_set_chain: 1 = coarse levels of single-frame builds in one launch, 2 = of batched builds too
  void *d_trk = nullptr;           // device image of stage_trk
  size_t d_trk_bytes = 0;
  gklt_flist *scratch_list = nullptr;
  double *d_invk = nullptr;  // 9 doubles
  double *d_norm = nullptr, *d_vec = nullptr;
  int d_nv_capacity = 0;
  double *d_rot = nullptr;  // row rotations + invK of the rotate-and-normalise epilogue
  int d_rot_capacity = 0;   // in doubles
  void *noalloc_scratch = nullptr;  // this GPU's address of gklt_api.hip's `scratch` array
  // profiling
  int pyr_impl = 0;    // gklt_ctx_set_pyramid_impl: 0 default, 1 separate kernels, 2 tile kernels, 3 streaming kernels, 4 contracted (FMA) tile kernels
  int track_waves = 0;  // wavefronts per tracker launch (0 = default cap)
  int sel_impl = 0;         // suppression rounds: 0 = box tops (dense 32-bit keys where the radius rules them out), 1 = dense 64-bit keys, 2 = dense 32-bit keys
  int sel_local_iters = 3;  // selector: fixpoint iterations per tile and launch (dense rounds)
  int sel_scores = 0;       // gklt_ctx_set_select_scores: 0 = integral images (the reference's inputs), 1 = direct window sums
  int sel_tops_iters = 12;  // the same for the rounds on box tops, whose iterations cost less
  int track_impl = 0;  // 0 = one feature per wavefront, 1 = one feature per 8-lane slice
  bool profiling = false;
  struct Span {
    hipEvent_t a, b;
    int id;
  };
  std::mutex mu;
  std::vector<Span> spans;
  std::vector<hipEvent_t> free_events;
  double prof_ms[GKLT_K_COUNT] = {};
  int64_t prof_n[GKLT_K_COUNT] = {};
};

struct gklt_pyramid {
  gklt_ctx *ctx = nullptr;
  int nlevels = 0, ss = 2;
  gklt::Level lv[GKLT_MAX_LEVELS];
  float *tmp = nullptr;      // level-0-sized scratch plane (generic resize path, plane download)
  hipEvent_t ready = nullptr;     // recorded on s_pyr after a build
  hipEvent_t consumed = nullptr;  // recorded on s_trk after a tracker launch that read it
  bool has_ready = false, has_consumed = false;
  bool built_on_pyr = false;  // last written by gklt_pyramid_build / _upload, i.e. on s_pyr
  // k_level_chain (levels 1 .. L-1 in one launch): one flag per tile of those levels, holding the
  // epoch of the last launch that finished the tile
  unsigned long long build_gen = 0;  // a value no other build on the context has, set by every build /
                                     // plane upload (gklt_select_prepare's tag)
  unsigned *chain_flags = nullptr;
  int chain_tiles = 0;
  unsigned chain_epoch = 0;
};

struct gklt_flist {
  gklt_ctx *ctx = nullptr;
  int capacity = 0, n = 0;
  float *x = nullptr, *y = nullptr;
  int *val = nullptr;
  unsigned long long *stats = nullptr;  // GKLT_STAT_SHARDS x {tracked, iterations}
  // page-locked mirror of the list layout through which host lists are uploaded (flist_upload_staged)
  float *h_stage = nullptr;
  hipEvent_t stage_done = nullptr;
  std::vector<std::pair<size_t, size_t>> stage_pending;  // [first, last) entries still being copied
};

namespace gklt {

// RAII-less profiling bracket: call begin() before and end() after the launch.
struct ProfScope {
  gklt_ctx *ctx;
  hipStream_t s;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(gklt_ctx *c, hipStream_t st, int kid);
  ~ProfScope();
};

int half_taps(const float *full, int width, int sym, Taps *out);
int fix_window(gklt_params *p);

// pyramid kernels (gklt_pyramid.hip)
int launch_smooth_u8(gklt_ctx *ctx, hipStream_t s, const uint8_t *src, int stride, int w, int h,
                     const Taps &g, float *dst);
int launch_smooth_f32(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, const Taps &g,
                      float *dst, int prof_id);
int launch_smooth_decimate2(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, float *dst, int w2, int h2);
int launch_resize_linear(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, float *dst,
                         int w2, int h2);
// gradients + texel packing of levels [first, first+count) in ONE launch
int launch_gradients_tex(gklt_ctx *ctx, hipStream_t s, const Level *levels, int first, int count,
                         const Taps &g, const Taps &d);
int launch_gradients_planar(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, const Taps &d, float *gx, float *gy);
// fused level kernels (one launch per level): see gklt_pyramid.hip
bool level_fusable(const Taps *s0, const Taps *pyr, const Taps &g, const Taps &d);
int launch_level(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, const uint8_t *const *src,
                 int n, int stride, int level, const Taps &s0, const Taps *pyr, const Taps &g,
                 const Taps &d, float *img0_out);
// the same kernels built with mul+add contraction (gklt_pyramid.hip compiled -DGKLT_PYR_FMA)
int launch_level_fma(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, const uint8_t *const *src,
                 int n, int stride, int level, const Taps &s0, const Taps *pyr, const Taps &g,
                 const Taps &d, float *img0_out);
// levels 1 .. L-1 of one pyramid in one launch (exact 2x decimations only)
int launch_level_chain(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, int n, const Taps &pyr,
                       const Taps &g, const Taps &d);
// streaming form of the fused level kernel (gklt_pyramid_stream.hip)
bool level_streamable(int w, int h, int level, const Taps &s0);
int launch_level_stream(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, const uint8_t *const *src, int n,
                        int stride, int level, const Taps &s0, const Taps *pyr, const Taps &g, const Taps &d);
int launch_tex_extract(gklt_ctx *ctx, hipStream_t s, const Texel *tex, int n, int comp, float *dst);
int launch_tex_to_u8(gklt_ctx *ctx, hipStream_t s, const Texel *tex, int n, uint8_t *dst);
int launch_tex_insert(gklt_ctx *ctx, hipStream_t s, Texel *tex, int n, int comp, const float *src);

// tracker (gklt_track.hip)
int launch_track(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *p1, const gklt_pyramid *p2,
                 int n, float *x, float *y, int *val, const double *d_invk, double *d_norm,
                 double *d_vec, unsigned long long *stats, float *mirror = nullptr,
                 int mirror_stride = 0);
int launch_track_frames(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *const *py, int nframes,
                        int n, float *x, float *y, int *val, const double *d_invk, double *d_norm,
                        double *d_vec, unsigned long long *stats, float *const *mirrors,
                        int mirror_stride);
int launch_track_streams(gklt_ctx *ctx, hipStream_t s, const Texel *const *tab, int nstreams,
                         int slot1, int slot2, int cap, const int *n, const gklt_pyramid *py0,
                         float *x, float *y, int *val, unsigned long long *stats, float *mirror,
                         int mirror_stride);
// H2D copy of a w x h u8 plane with row stride `stride` into a dense device plane.  Always the 2-D
// copy, also for dense sources: from page-locked memory it moves 37 GB/s on a stream that also
// runs kernels, while the 1-D hipMemcpyAsync of the same 2 MB took 0.6 ms per frame when the copies
// alternate between two streams (gklt_pipeline_step), 3 GB/s (measured, round 2).
inline hipError_t copy_plane_h2d(void *dst, const void *src, int stride, int w, int h, hipStream_t s) {
  return hipMemcpy2DAsync(dst, w, src, stride, w, h, hipMemcpyHostToDevice, s);
}
// Replaces entries [off, off + n) of the list from host arrays on stream s.  The arrays may be
// pageable and may be reused as soon as the call returns: they are copied into the list's
// page-locked mirror first, and the H2D copies read from there (gklt_api.hip).
int flist_upload_staged(gklt_flist *fl, hipStream_t s, size_t off, int n, const float *x,
                        const float *y, const int *val);
// ingested frames (gklt_ingest.hip): device luma plane of the last ingest once `consumer` may read
// it; the consumer records when it is done with the plane
int frame_source(gklt_frame *f, hipStream_t consumer, const uint8_t **luma, int *stride);
int frame_consumed(gklt_frame *f, hipStream_t consumer);
gklt_ctx *frame_ctx(gklt_frame *f);
bool frame_matches(gklt_frame *f, int w, int h);
// bearing vectors (gklt_bearing.hip); grows ctx->d_norm / d_vec to n entries (gklt_api.hip)
int ensure_bearing_buffers(gklt_ctx *ctx, int n);
// all launches of one pyramid build on stream s (gklt_api.hip)
int enqueue_build(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *py, const uint8_t *src,
                  int src_stride);
// the same for n same-sized pyramids (n <= GKLT_MAX_LOOKAHEAD): fused levels take one launch per
// level for the whole batch
int enqueue_build_batch(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py,
                        const uint8_t *const *src, int n, int src_stride,
                        hipStream_t s_coarse = nullptr, hipEvent_t e_level0 = nullptr);

int coarse_stream(gklt_ctx *ctx, hipStream_t *out);

}  // namespace gklt
