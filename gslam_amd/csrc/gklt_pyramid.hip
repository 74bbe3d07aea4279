// Pyramid kernels for gfx950: separable Gaussian smoothing, smoothing + 2x decimation, and
// Gaussian-derivative gradients, each as ONE launch per level that stages a haloed tile in LDS,
// runs the row pass LDS -> LDS and the column pass LDS -> HBM.
//
// Replaces cv::sepFilter2D / cv::resize as called from pyramid.cpp:45-84 and convolve.cpp:19-56.
// Arithmetic is fp32 with every product and sum rounded separately (the translation unit is
// built with -ffp-contract=off) in the order
//     symmetric taps      k0*s0 + sum_{j=1..r} kj*(s[+j] + s[-j])
//     antisymmetric taps        sum_{j=1..r} kj*(s[+j] - s[-j])
// with BORDER_REFLECT_101 on the source index, which is the order the test oracle uses, so the
// planes can be compared bit for bit.
//
// HBM layout written here: per level a planar fp32 image `img` (input of the
// next level, downloads) and an interleaved texel array (I, gx, gy, 0) that the tracker gathers
// with one 16-byte load per bilinear tap.
#include "gklt_internal.h"

namespace gklt {

namespace {

constexpr int TW = 64;  // output tile width  (one wavefront spans a tile row)
constexpr int TH = 16;  // output tile height
constexpr int NT = 256; // threads per block

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// 1-D correlation centred on s[0]; `st` is the element stride between neighbouring samples.
__device__ __forceinline__ float corr(const float *s, int st, const Taps &t) {
  float acc;
  if (t.sym > 0) {
    acc = t.k[0] * s[0];
    for (int j = 1; j <= t.r; ++j) acc = acc + t.k[j] * (s[j * st] + s[-j * st]);
  } else {
    acc = 0.0f;
    for (int j = 1; j <= t.r; ++j) acc = acc + t.k[j] * (s[j * st] - s[-j * st]);
  }
  return acc;
}

// Stage the (TH+2ry) x (TW+2rx) source window of the tile at (x0,y0) into LDS, reflecting at
// the image border.  Rows of consecutive lanes read consecutive source addresses.
template <typename SrcT>
__device__ __forceinline__ void stage_tile(const SrcT *src, int stride, int w, int h, int x0,
                                           int y0, int rx, int ry, float *in, int inW) {
  const int inH = TH + 2 * ry;
  for (int i = threadIdx.x; i < inW * inH; i += NT) {
    const int ty = i / inW, tx = i - ty * inW;
    const int sy = reflect101(y0 - ry + ty, h), sx = reflect101(x0 - rx + tx, w);
    in[i] = static_cast<float>(src[(size_t)sy * stride + sx]);
  }
}

// ---------------------------------------------------------------- smooth (P1) --------------
// MODE 0: dst = smoothed plane.  MODE 1: dst = 2x2 mean of the smoothed plane (w2 x h2).
template <typename SrcT, int MODE>
__global__ __launch_bounds__(NT) void k_smooth(const SrcT *__restrict__ src, int stride, int w,
                                               int h, Taps g, float *__restrict__ dst, int w2,
                                               int h2) {
  extern __shared__ __align__(16) float lds[];
  const int r = g.r;
  const int inW = TW + 2 * r, inH = TH + 2 * r;
  float *in = lds;             // inH x inW
  float *row = lds + inW * inH; // inH x TW
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;

  stage_tile(src, stride, w, h, x0, y0, r, r, in, inW);
  __syncthreads();
  for (int i = threadIdx.x; i < inH * TW; i += NT) {
    const int ty = i / TW, tx = i % TW;
    row[i] = corr(in + ty * inW + tx + r, 1, g);
  }
  __syncthreads();
  if (MODE == 0) {
    for (int i = threadIdx.x; i < TH * TW; i += NT) {
      const int ty = i / TW, tx = i % TW;
      const int x = x0 + tx, y = y0 + ty;
      const float v = corr(row + (ty + r) * TW + tx, TW, g);
      if (x < w && y < h) dst[(size_t)y * w + x] = v;
    }
  } else {
    // smoothed tile back into `in` (free after the row pass), then the 2x2 means
    float *sm = in;
    for (int i = threadIdx.x; i < TH * TW; i += NT) {
      const int ty = i / TW, tx = i % TW;
      sm[i] = corr(row + (ty + r) * TW + tx, TW, g);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (TH / 2) * (TW / 2); i += NT) {
      const int ty = i / (TW / 2), tx = i % (TW / 2);
      const int x = x0 / 2 + tx, y = y0 / 2 + ty;
      const float *p = sm + (2 * ty) * TW + 2 * tx;
      const float v = (((p[0] + p[1]) + p[TW]) + p[TW + 1]) * 0.25f;
      if (x < w2 && y < h2) dst[(size_t)y * w2 + x] = v;
    }
  }
}

// ---------------------------------------------------------------- gradients (P2) -----------
// gx = rows G', cols G ; gy = rows G, cols G'  (convolve.cpp:36-37, pyramid.cpp:77-78).
// One read of I produces both.  TEX: write (I,gx,gy,0) texels, else two planar outputs.
template <bool TEX>
__global__ __launch_bounds__(NT) void k_grad(const float *__restrict__ src, int w, int h, Taps g,
                                             Taps d, Texel *__restrict__ tex,
                                             float *__restrict__ gxo, float *__restrict__ gyo) {
  extern __shared__ __align__(16) float lds[];
  const int rx = max(g.r, d.r), ry = rx;
  const int inW = TW + 2 * rx, inH = TH + 2 * ry;
  float *in = lds;                 // inH x inW
  float *rowd = in + inW * inH;    // inH x TW : rows filtered with G'
  float *rowg = rowd + inH * TW;   // inH x TW : rows filtered with G
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;

  stage_tile(src, w, w, h, x0, y0, rx, ry, in, inW);
  __syncthreads();
  for (int i = threadIdx.x; i < inH * TW; i += NT) {
    const int ty = i / TW, tx = i % TW;
    const float *c = in + ty * inW + tx + rx;
    rowd[i] = corr(c, 1, d);
    rowg[i] = corr(c, 1, g);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * TW; i += NT) {
    const int ty = i / TW, tx = i % TW;
    const int x = x0 + tx, y = y0 + ty;
    const float gx = corr(rowd + (ty + ry) * TW + tx, TW, g);
    const float gy = corr(rowg + (ty + ry) * TW + tx, TW, d);
    if (x < w && y < h) {
      if (TEX) {
        tex[(size_t)y * w + x] = make_float4(in[(ty + ry) * inW + tx + rx], gx, gy, 0.0f);
      } else {
        gxo[(size_t)y * w + x] = gx;
        gyo[(size_t)y * w + x] = gy;
      }
    }
  }
}

// ---------------------------------------------------------------- generic resize -----------
// cv::resize(INTER_LINEAR) for scales other than exactly 2 (pyramid.cpp:65 with subsampling 4/8
// or odd sizes): sx = (dx+0.5)*(W/W2)-0.5, source index clamped, horizontal then vertical blend.
__global__ void k_resize_linear(const float *__restrict__ src, int w, int h,
                                float *__restrict__ dst, int w2, int h2, double sx_scale,
                                double sy_scale) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w2 || y >= h2) return;
  float fy = (float)((y + 0.5) * sy_scale - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  if (sy < 0) { sy = 0; fy = 0.f; }
  if (sy >= h - 1) { sy = h - 1; fy = 0.f; }
  const int sy1 = min(sy + 1, h - 1);
  float fx = (float)((x + 0.5) * sx_scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) { sx = 0; fx = 0.f; }
  if (sx >= w - 1) { sx = w - 1; fx = 0.f; }
  const int sx1 = min(sx + 1, w - 1);
  const float top = src[(size_t)sy * w + sx] * (1.f - fx) + src[(size_t)sy * w + sx1] * fx;
  const float bot = src[(size_t)sy1 * w + sx] * (1.f - fx) + src[(size_t)sy1 * w + sx1] * fx;
  dst[(size_t)y * w2 + x] = top * (1.f - fy) + bot * fy;
}

__global__ void k_tex_extract(const Texel *__restrict__ tex, int n, int comp,
                              float *__restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Texel t = tex[i];
  dst[i] = comp == 0 ? t.x : (comp == 1 ? t.y : t.z);
}

__global__ void k_tex_insert(Texel *__restrict__ tex, int n, int comp,
                             const float *__restrict__ src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Texel t = tex[i];
  const float v = src[i];
  if (comp == 0) t.x = v; else if (comp == 1) t.y = v; else t.z = v;
  t.w = 0.0f;
  tex[i] = t;
}

inline dim3 tile_grid(int w, int h) { return dim3((w + TW - 1) / TW, (h + TH - 1) / TH); }
inline size_t smooth_lds(int r) {
  return sizeof(float) * ((size_t)(TW + 2 * r) * (TH + 2 * r) + (size_t)(TH + 2 * r) * TW);
}
inline size_t grad_lds(int r) {
  return sizeof(float) * ((size_t)(TW + 2 * r) * (TH + 2 * r) + 2 * (size_t)(TH + 2 * r) * TW);
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) {
    set_error("filter radius needs %zu bytes of LDS (> 160 KiB)", bytes);
    return GKLT_EINVAL;
  }
  if (bytes > 48 * 1024)
    GKLT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

}  // namespace

int launch_smooth_u8(gklt_ctx *ctx, hipStream_t s, const uint8_t *src, int stride, int w, int h,
                     const Taps &g, float *dst) {
  const size_t lds = smooth_lds(g.r);
  if (int rc = allow_lds(k_smooth<uint8_t, 0>, lds)) return rc;
  ProfScope ps(ctx, s, GKLT_K_SMOOTH0);
  hipLaunchKernelGGL((k_smooth<uint8_t, 0>), tile_grid(w, h), dim3(NT), lds, s, src, stride, w, h,
                     g, dst, 0, 0);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_smooth_f32(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, const Taps &g,
                      float *dst, int prof_id) {
  const size_t lds = smooth_lds(g.r);
  if (int rc = allow_lds(k_smooth<float, 0>, lds)) return rc;
  ProfScope ps(ctx, s, prof_id);
  hipLaunchKernelGGL((k_smooth<float, 0>), tile_grid(w, h), dim3(NT), lds, s, src, w, w, h, g, dst,
                     0, 0);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_smooth_decimate2(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, float *dst, int w2, int h2) {
  if (w != 2 * w2 || h != 2 * h2) {
    set_error("smooth_decimate2 needs an exact 2x size ratio");
    return GKLT_EINVAL;
  }
  const size_t lds = smooth_lds(g.r);
  if (int rc = allow_lds(k_smooth<float, 1>, lds)) return rc;
  ProfScope ps(ctx, s, GKLT_K_DOWN);
  hipLaunchKernelGGL((k_smooth<float, 1>), tile_grid(w, h), dim3(NT), lds, s, src, w, w, h, g, dst,
                     w2, h2);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_resize_linear(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, float *dst,
                         int w2, int h2) {
  ProfScope ps(ctx, s, GKLT_K_DOWN);
  hipLaunchKernelGGL(k_resize_linear, dim3((w2 + 255) / 256, h2), dim3(256), 0, s, src, w, h, dst,
                     w2, h2, (double)w / w2, (double)h / h2);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_gradients_tex(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                         const Taps &g, const Taps &d, Texel *tex) {
  const size_t lds = grad_lds(std::max(g.r, d.r));
  if (int rc = allow_lds(k_grad<true>, lds)) return rc;
  ProfScope ps(ctx, s, GKLT_K_GRAD);
  hipLaunchKernelGGL((k_grad<true>), tile_grid(w, h), dim3(NT), lds, s, src, w, h, g, d, tex,
                     (float *)nullptr, (float *)nullptr);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_gradients_planar(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, const Taps &d, float *gx, float *gy) {
  const size_t lds = grad_lds(std::max(g.r, d.r));
  if (int rc = allow_lds(k_grad<false>, lds)) return rc;
  ProfScope ps(ctx, s, GKLT_K_GRAD);
  hipLaunchKernelGGL((k_grad<false>), tile_grid(w, h), dim3(NT), lds, s, src, w, h, g, d,
                     (Texel *)nullptr, gx, gy);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_tex_extract(gklt_ctx *ctx, hipStream_t s, const Texel *tex, int n, int comp,
                       float *dst) {
  ProfScope ps(ctx, s, GKLT_K_MISC);
  hipLaunchKernelGGL(k_tex_extract, dim3((n + 255) / 256), dim3(256), 0, s, tex, n, comp, dst);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_tex_insert(gklt_ctx *ctx, hipStream_t s, Texel *tex, int n, int comp,
                      const float *src) {
  ProfScope ps(ctx, s, GKLT_K_MISC);
  hipLaunchKernelGGL(k_tex_insert, dim3((n + 255) / 256), dim3(256), 0, s, tex, n, comp, src);
  GKLT_HIP(hipGetLastError());
  return 0;
}

}  // namespace gklt
