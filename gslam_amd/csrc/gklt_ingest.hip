// Frame ingest: what main.cpp:77-107 does on the host for every decoded frame,
// on the GPU -- the NV12 pixel buffer (full-resolution Y plane, half-resolution interleaved chroma
// plane) goes H2D once (1.5 bytes per pixel); the Y plane feeds computePyramid where it lies, and
// one kernel forms the half-resolution colour image System keeps for point colours
// (System.cpp:310-315,436,705: colorImage->at<Vec3b>(round(y/2), round(x/2))):
//   yuv[0] = cv::resize(Y, W/2 x H/2)          INTER_LINEAR at exactly 1/2 = the 2x2 area mean,
//                                              8 bit: (a + b + c + d + 2) >> 2
//   yuv[1], yuv[2] = cv::split(uv)             the two bytes of every chroma sample, in memory order
//   rgb = cvtColor(merge(yuv), CV_YCrCb2RGB)   channel 1 is read as Cr, channel 2 as Cb -- whatever
//                                              the pixel format put there -- 8-bit fixed point:
//     R = Y + ((Cr-128)*22987 + 8192 >> 14)
//     G = Y + ((Cb-128)*-5636 + (Cr-128)*-11698 + 8192 >> 14)
//     B = Y + ((Cb-128)*29049 + 8192 >> 14)     each saturated to [0, 255]
// The arithmetic is OpenCV's (un-vendored, version unpinned: SURVEY F4); the constants are those of
// its 8-bit YCrCb2RGB path.  oracle/oracle.py::ingest_nv12 restates the same integers.
#include "gklt_internal.h"

struct gklt_frame {
  gklt_ctx *ctx = nullptr;
  int W = 0, H = 0;
  uint8_t *d_y = nullptr, *d_uv = nullptr;   // own copies of host planes
  uint8_t *d_rgb = nullptr;                  // (H/2) x (W/2) x 3
  const uint8_t *luma = nullptr, *chroma = nullptr;  // the planes of the last ingest (own or borrowed)
  int luma_stride = 0, chroma_stride = 0;
  bool has_color = false;
  hipEvent_t ready = nullptr, consumed = nullptr;
  bool has_ready = false, has_consumed = false;
};

namespace gklt {
namespace {

__device__ __forceinline__ int sat8(int v) { return min(max(v, 0), 255); }

__device__ __forceinline__ void ycrcb_to_rgb(int Y, int c1, int c2, uint8_t *rgb) {
  const int cr = c1 - 128, cb = c2 - 128;
  rgb[0] = (uint8_t)sat8(Y + ((cr * 22987 + 8192) >> 14));
  rgb[1] = (uint8_t)sat8(Y + ((cb * -5636 + cr * -11698 + 8192) >> 14));
  rgb[2] = (uint8_t)sat8(Y + ((cb * 29049 + 8192) >> 14));
}

// one thread = 4 adjacent half-resolution pixels: two 8-byte loads of Y rows, one 8-byte load of
// chroma, one 12-byte store (rows and pointers 8-byte aligned, W/2 a multiple of 4)
__global__ void k_nv12_color4(const uint8_t *__restrict__ y, int ys, const uint8_t *__restrict__ uv,
                              int uvs, int w2, int h2, uint8_t *__restrict__ rgb) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (4 * q >= w2) return;
  const uint2 a = *reinterpret_cast<const uint2 *>(y + (size_t)(2 * j) * ys + 8 * q);
  const uint2 b = *reinterpret_cast<const uint2 *>(y + (size_t)(2 * j + 1) * ys + 8 * q);
  const uint2 c = *reinterpret_cast<const uint2 *>(uv + (size_t)j * uvs + 8 * q);
  const unsigned ya[2] = {a.x, a.y}, yb[2] = {b.x, b.y}, cc[2] = {c.x, c.y};
  uint8_t out[12];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned ra = ya[k >> 1] >> (16 * (k & 1)), rb = yb[k >> 1] >> (16 * (k & 1));
    const int Y = (int)((ra & 0xff) + ((ra >> 8) & 0xff) + (rb & 0xff) + ((rb >> 8) & 0xff) + 2) >> 2;
    const unsigned ch = cc[k >> 1] >> (16 * (k & 1));
    ycrcb_to_rgb(Y, (int)(ch & 0xff), (int)((ch >> 8) & 0xff), out + 3 * k);
  }
  unsigned *dst = reinterpret_cast<unsigned *>(rgb + ((size_t)j * w2 + 4 * q) * 3);
  dst[0] = out[0] | out[1] << 8 | out[2] << 16 | (unsigned)out[3] << 24;
  dst[1] = out[4] | out[5] << 8 | out[6] << 16 | (unsigned)out[7] << 24;
  dst[2] = out[8] | out[9] << 8 | out[10] << 16 | (unsigned)out[11] << 24;
}

// any alignment / width: one thread = one half-resolution pixel
__global__ void k_nv12_color1(const uint8_t *__restrict__ y, int ys, const uint8_t *__restrict__ uv,
                              int uvs, int w2, int h2, uint8_t *__restrict__ rgb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= w2) return;
  const uint8_t *r0 = y + (size_t)(2 * j) * ys + 2 * i, *r1 = r0 + ys;
  const int Y = (r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2;
  const uint8_t *c = uv + (size_t)j * uvs + 2 * i;
  ycrcb_to_rgb(Y, c[0], c[1], rgb + ((size_t)j * w2 + i) * 3);
}

// colour of every live feature: colorImage->at<Vec3b>(round(y / 2), round(x / 2))
// (System.cpp:310-315; round = half away from zero).  The reference indexes without a bounds
// test; indices are clamped here.  Dead features (val < 0) get (0, 0, 0).
__global__ void k_feature_colors(const float *__restrict__ x, const float *__restrict__ y,
                                 const int *__restrict__ val, int n, const uint8_t *__restrict__ rgb,
                                 int w2, int h2, uint8_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t c[3] = {0, 0, 0};
  if (val[i] >= 0) {
    const int r = min(max((int)round((double)(y[i] / 2)), 0), h2 - 1);
    const int q = min(max((int)round((double)(x[i] / 2)), 0), w2 - 1);
    const uint8_t *p = rgb + ((size_t)r * w2 + q) * 3;
    c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
  }
  out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
}

int ingest(gklt_ctx *ctx, gklt_frame *f, const uint8_t *y, int ys, const uint8_t *uv, int uvs, int on_device) {
  if (!ctx || !f || f->ctx != ctx || !y) return GKLT_EINVAL;
  if (ys < f->W || (uv && uvs < f->W)) {
    set_error("ingest: stride smaller than the %d-pixel row", f->W);
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->s_pyr;
  // the planes of the previous ingest may still feed a pyramid build
  if (f->has_consumed) GKLT_HIP(hipStreamWaitEvent(s, f->consumed, 0));
  const int W = f->W, H = f->H;
  if (on_device) {  // borrowed: the caller keeps the planes valid until they are consumed
    f->luma = y; f->luma_stride = ys;
    f->chroma = uv; f->chroma_stride = uvs;
  } else {
    GKLT_HIP(copy_plane_h2d(f->d_y, y, ys, W, H, s));
    f->luma = f->d_y; f->luma_stride = W;
    if (uv) {
      GKLT_HIP(copy_plane_h2d(f->d_uv, uv, uvs, W, H / 2, s));
      f->chroma = f->d_uv; f->chroma_stride = W;
    }
  }
  if (uv) {
    const int w2 = W / 2, h2 = H / 2;
    ProfScope ps(ctx, s, GKLT_K_MISC);
    const bool fast = w2 % 4 == 0 && f->luma_stride % 8 == 0 && f->chroma_stride % 8 == 0 &&
                      (reinterpret_cast<size_t>(f->luma) | reinterpret_cast<size_t>(f->chroma)) % 8 == 0;
    if (fast)
      hipLaunchKernelGGL(k_nv12_color4, dim3((w2 / 4 + 255) / 256, h2), dim3(256), 0, s, f->luma, f->luma_stride,
                         f->chroma, f->chroma_stride, w2, h2, f->d_rgb);
    else
      hipLaunchKernelGGL(k_nv12_color1, dim3((w2 + 255) / 256, h2), dim3(256), 0, s, f->luma, f->luma_stride,
                         f->chroma, f->chroma_stride, w2, h2, f->d_rgb);
    GKLT_HIP(hipGetLastError());
    f->has_color = true;
  }
  GKLT_HIP(hipEventRecord(f->ready, s));
  f->has_ready = true;
  return 0;
}

}  // namespace

// used by the entry points that consume an ingested frame (gklt_pipeline.hip)
int frame_source(gklt_frame *f, hipStream_t consumer, const uint8_t **luma, int *stride) {
  if (!f->has_ready) {
    set_error("frame has not been ingested");
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipStreamWaitEvent(consumer, f->ready, 0));
  *luma = f->luma;
  *stride = f->luma_stride;
  return 0;
}
int frame_consumed(gklt_frame *f, hipStream_t consumer) {
  GKLT_HIP(hipEventRecord(f->consumed, consumer));
  f->has_consumed = true;
  return 0;
}
gklt_ctx *frame_ctx(gklt_frame *f) { return f->ctx; }
bool frame_matches(gklt_frame *f, int w, int h) { return f->W == w && f->H == h; }

}  // namespace gklt

using namespace gklt;

extern "C" int gklt_frame_create(gklt_ctx *ctx, int width, int height, gklt_frame **out) {
  if (!ctx || !out || width < 2 || height < 2) return GKLT_EINVAL;
  *out = nullptr;
  if ((width | height) & 1) {
    set_error("NV12 frames have even sizes (%dx%d)", width, height);
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipSetDevice(ctx->device));
  gklt_frame *f = new gklt_frame();
  f->ctx = ctx;
  f->W = width;
  f->H = height;
  const size_t px = (size_t)width * height;
  if (hipMalloc(&f->d_y, px) != hipSuccess || hipMalloc(&f->d_uv, px / 2) != hipSuccess ||
      hipMalloc(&f->d_rgb, px / 4 * 3) != hipSuccess ||
      hipEventCreateWithFlags(&f->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&f->consumed, hipEventDisableTiming) != hipSuccess) {
    set_error("resources of a %dx%d frame could not be created", width, height);
    gklt_frame_destroy(f);
    return GKLT_ENOMEM;
  }
  *out = f;
  return 0;
}

extern "C" int gklt_frame_destroy(gklt_frame *f) {
  if (!f) return 0;
  (void)hipSetDevice(f->ctx->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(f->d_y);
  (void)hipFree(f->d_uv);
  (void)hipFree(f->d_rgb);
  if (f->ready) (void)hipEventDestroy(f->ready);
  if (f->consumed) (void)hipEventDestroy(f->consumed);
  delete f;
  return 0;
}

extern "C" int gklt_frame_ingest_nv12(gklt_ctx *ctx, gklt_frame *f, const uint8_t *y, int y_stride,
                                      const uint8_t *uv, int uv_stride, int on_device) {
  if (!uv) return GKLT_EINVAL;
  return ingest(ctx, f, y, y_stride, uv, uv_stride, on_device);
}

extern "C" int gklt_frame_ingest_gray(gklt_ctx *ctx, gklt_frame *f, const uint8_t *y, int y_stride,
                                      int on_device) {
  return ingest(ctx, f, y, y_stride, nullptr, 0, on_device);
}

extern "C" int gklt_frame_luma(gklt_ctx *ctx, gklt_frame *f, const uint8_t **device_ptr, int *stride) {
  if (!ctx || !f || f->ctx != ctx || !device_ptr || !stride) return GKLT_EINVAL;
  if (!f->has_ready) {
    set_error("frame has not been ingested");
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipSetDevice(ctx->device));
  GKLT_HIP(hipEventSynchronize(f->ready));
  *device_ptr = f->luma;
  *stride = f->luma_stride;
  return 0;
}

extern "C" int gklt_pyramid_build_frame(gklt_ctx *ctx, gklt_pyramid *py, gklt_frame *f) {
  if (!ctx || !py || !f || f->ctx != ctx || py->ctx != ctx) return GKLT_EINVAL;
  if (!f->has_ready) {
    set_error("frame has not been ingested");
    return GKLT_EINVAL;
  }
  // ingest and build share the context's pyramid stream: ordered without an event
  if (int rc = gklt_pyramid_build(ctx, py, f->luma, f->luma_stride, 1)) return rc;
  return frame_consumed(f, ctx->s_pyr);
}

extern "C" int gklt_frame_download_color(gklt_ctx *ctx, gklt_frame *f, uint8_t *rgb, int stride) {
  if (!ctx || !f || f->ctx != ctx || !rgb || stride < f->W / 2 * 3) return GKLT_EINVAL;
  if (!f->has_color) {
    set_error("frame holds no colour image (no NV12 ingest yet)");
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->s_pyr;
  GKLT_HIP(hipMemcpy2DAsync(rgb, stride, f->d_rgb, (size_t)f->W / 2 * 3, (size_t)f->W / 2 * 3, f->H / 2,
                            hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gklt_flist_colors(gklt_ctx *ctx, gklt_frame *f, gklt_flist *fl, uint8_t *rgb_out) {
  if (!ctx || !f || !fl || f->ctx != ctx || fl->ctx != ctx || !rgb_out) return GKLT_EINVAL;
  if (!f->has_color) {
    set_error("frame holds no colour image (no NV12 ingest yet)");
    return GKLT_EINVAL;
  }
  const int n = fl->n;
  if (n == 0) return 0;
  GKLT_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->s_trk;  // the list's stream
  GKLT_HIP(hipStreamWaitEvent(s, f->ready, 0));
  if (int rc = ensure_bearing_buffers(ctx, (n + 7) / 8 + 1)) return rc;  // 24 bytes per entry: room for 3 n bytes
  uint8_t *d_out = reinterpret_cast<uint8_t *>(ctx->d_norm);
  hipLaunchKernelGGL(k_feature_colors, dim3((n + 255) / 256), dim3(256), 0, s, fl->x, fl->y, fl->val, n, f->d_rgb,
                     f->W / 2, f->H / 2, d_out);
  GKLT_HIP(hipGetLastError());
  GKLT_HIP(hipMemcpyAsync(rgb_out, d_out, 3 * (size_t)n, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}
