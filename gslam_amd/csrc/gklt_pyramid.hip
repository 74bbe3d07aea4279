// Pyramid kernels for gfx950.  Two families in this file:
//  * k_level<SrcT, R0, RP, RG, DOWN, NTHREADS> (the default path): ONE launch per pyramid level
//    that fuses smoothing, both gradient passes, texel packing and the 2x decimation for the next
//    level; 64 x 32 output tile per workgroup of 4 or 8 wavefronts, register-blocked passes
//    through LDS, blockIdx.z = frame (lookahead) or stream (multi-stream).  DESIGN.md 4.1.
//    (gklt_pyramid_stream.hip holds a second, barrier-free form of the same kernel: one
//    wavefront per column strip, selectable with gklt_ctx_set_pyramid_impl(ctx, 3).)
//  * k_smooth_fast / k_grad_fast / k_smooth / k_grad (pyramid impl 1, the standalone gklt_smooth /
//    gklt_gradients entry points, and every configuration the fused kernel is not instantiated
//    for): separable Gaussian with optional 2x decimation, and gradients + texel packing, each
//    staging a haloed tile in LDS, row pass LDS -> LDS, column pass LDS -> HBM.
//
// Replaces cv::sepFilter2D / cv::resize as called from pyramid.cpp:45-84 and convolve.cpp:19-56.
// Arithmetic is fp32 with every product and sum rounded separately (the translation unit is
// built with -ffp-contract=off) in the order
//     symmetric taps      k0*s0 + sum_{j=1..r} kj*(s[+j] + s[-j])
//     antisymmetric taps        sum_{j=1..r} kj*(s[+j] - s[-j])
// with BORDER_REFLECT_101 on the source index, which is the order the test oracle uses, so the
// planes can be compared bit for bit.
//
// Separate kernels: tile = 64 x 16 outputs per 256-thread block (TW, TH below); a wavefront owns
// whole tile rows (lane = column), so LDS rows are read and written conflict-free and HBM rows are
// 256-byte (fp32) or 768-byte (12-byte texel) contiguous stores.  Kernel radii are template
// parameters for the radii the reference's sigmas produce (taps live in SGPRs, loops fully
// unrolled); other radii take the R = 0 instantiation, which keeps the taps in LDS.
//
// HBM layout written here: per level a planar fp32 image `img` (input of the
// next level, downloads) and an interleaved texel array (I, gx, gy) that the tracker gathers
// with one 12-byte load per bilinear tap.
#include "gklt_internal.h"

namespace gklt {

namespace {

constexpr int TW = 64;   // output tile width  (one wavefront spans a tile row)
constexpr int TH = 16;   // output tile height
constexpr int NT = 256;  // threads per block
constexpr int NW = NT / 64;

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// Tap storage: compile-time radius -> registers/SGPRs, run-time radius -> LDS copy.
template <int R>
struct TapRef {
  const Taps &t;
  __device__ __forceinline__ int radius() const { return R; }
  __device__ __forceinline__ float k(int j) const { return t.k[j]; }
};
template <>
struct TapRef<0> {
  const float *lds;
  int r;
  __device__ __forceinline__ int radius() const { return r; }
  __device__ __forceinline__ float k(int j) const { return lds[j]; }
};

// 1-D correlation centred on s[0]; `st` is the element stride between neighbouring samples.
template <int R, int SYM>
__device__ __forceinline__ float corr(const float *s, int st, const TapRef<R> &t) {
  float acc;
  if (SYM > 0) {
    acc = t.k(0) * s[0];
    if constexpr (R > 0) {
#pragma unroll
      for (int j = 1; j <= R; ++j) acc = acc + t.k(j) * (s[j * st] + s[-j * st]);
    } else {
      for (int j = 1; j <= t.radius(); ++j) acc = acc + t.k(j) * (s[j * st] + s[-j * st]);
    }
  } else {
    acc = 0.0f;
    if constexpr (R > 0) {
#pragma unroll
      for (int j = 1; j <= R; ++j) acc = acc + t.k(j) * (s[j * st] - s[-j * st]);
    } else {
      for (int j = 1; j <= t.radius(); ++j) acc = acc + t.k(j) * (s[j * st] - s[-j * st]);
    }
  }
  return acc;
}

// Stage the (TH+2r) x (TW+2r) source window of the tile at (x0,y0) into LDS, reflecting at the
// image border.  Wave w takes rows w, w+4, ...; lanes walk the row, so loads are row-contiguous.
// With a compile-time radius the loops are fully unrolled and every load of the wave is issued
// before the first LDS store (one HBM round trip per block instead of one per row).
template <int R, typename SrcT>
__device__ __forceinline__ void stage_tile(const SrcT *__restrict__ src, int stride, int w, int h,
                                           int x0, int y0, int r, float *in, int inW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (R > 0) {
    constexpr int IH = TH + 2 * R, IW = TW + 2 * R;
    constexpr int ROWS = (IH + NW - 1) / NW;
    static_assert(IW <= 128, "two lane chunks per row");
    const int sx0 = reflect101(x0 - R + lane, w);
    const int sx1 = reflect101(x0 - R + 64 + lane, w);
    SrcT v0[ROWS], v1[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int ty = wave + i * NW;
      const int sy = reflect101(y0 - R + min(ty, IH - 1), h);
      const SrcT *row = src + (size_t)sy * stride;
      v0[i] = row[sx0];
      v1[i] = row[sx1];
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int ty = wave + i * NW;
      if (ty < IH) {
        in[ty * IW + lane] = static_cast<float>(v0[i]);
        if (lane < IW - 64) in[ty * IW + 64 + lane] = static_cast<float>(v1[i]);
      }
    }
  } else {
    const int inH = TH + 2 * r;
#pragma unroll 2
    for (int ty = wave; ty < inH; ty += NW) {
      const int sy = reflect101(y0 - r + ty, h);
      const SrcT *row = src + (size_t)sy * stride;
      for (int tx = lane; tx < inW; tx += 64)
        in[ty * inW + tx] = static_cast<float>(row[reflect101(x0 - r + tx, w)]);
    }
  }
}

template <int R>
__device__ __forceinline__ TapRef<R> make_taps(const Taps &t, float *lds_slot) {
  if constexpr (R > 0) {
    return TapRef<R>{t};
  } else {
    if ((int)threadIdx.x <= t.r) lds_slot[threadIdx.x] = t.k[threadIdx.x];
    return TapRef<0>{lds_slot, t.r};
  }
}

// ---------------------------------------------------------------- smooth (P1) --------------
// MODE 0: dst = smoothed plane.  MODE 1: dst = 2x2 mean of the smoothed plane (w2 x h2).
template <typename SrcT, int MODE, int R>
__global__ __launch_bounds__(NT) void k_smooth(const SrcT *__restrict__ src, int stride, int w,
                                               int h, Taps g, float *__restrict__ dst, int w2,
                                               int h2) {
  extern __shared__ __align__(16) float lds[];
  const int r = R > 0 ? R : g.r;
  const int inW = TW + 2 * r, inH = TH + 2 * r;
  float *in = lds;                // inH x inW
  float *row = in + inW * inH;    // inH x TW
  float *tapmem = row + inH * TW; // kMaxRadius+1 floats (R == 0 only)
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const TapRef<R> tg = make_taps<R>(g, tapmem);

  stage_tile<R>(src, stride, w, h, x0, y0, r, in, inW);
  __syncthreads();
  for (int ty = wave; ty < inH; ty += NW)
    row[ty * TW + lane] = corr<R, +1>(in + ty * inW + lane + r, 1, tg);
  __syncthreads();
  if (MODE == 0) {
    const int x = x0 + lane;
#pragma unroll
    for (int ty = wave; ty < TH; ty += NW) {
      const int y = y0 + ty;
      const float v = corr<R, +1>(row + (ty + r) * TW + lane, TW, tg);
      if (x < w && y < h) dst[(size_t)y * w + x] = v;
    }
  } else {
    // smoothed tile back into `in` (free after the row pass), then the 2x2 means
    float *sm = in;
#pragma unroll
    for (int ty = wave; ty < TH; ty += NW)
      sm[ty * TW + lane] = corr<R, +1>(row + (ty + r) * TW + lane, TW, tg);
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 outputs, one per thread
    const int x = x0 / 2 + tx, y = y0 / 2 + ty;
    const float *p = sm + (2 * ty) * TW + 2 * tx;
    const float v = GKLT_MEAN2X2(p[0], p[1], p[TW], p[TW + 1]);
    if (x < w2 && y < h2) dst[(size_t)y * w2 + x] = v;
  }
}

// ---------------------------------------------------------------- gradients (P2) -----------
// gx = rows G', cols G ; gy = rows G, cols G'  (convolve.cpp:36-37, pyramid.cpp:77-78).
// One read of I produces both.  Several levels can share one launch: block -> level by tile id.
struct GradJob {
  const float *src;
  Texel *tex;
  float *gx, *gy;  // planar outputs (tex == nullptr)
  int w, h;
  int tiles_x;
  int first_tile;  // prefix sum of tiles over the jobs of the launch
};
struct GradBatch {
  int n;
  GradJob job[GKLT_MAX_LEVELS];
};

// RG / RD: radii of G and G' (0,0 = run time)
template <int RG, int RD>
__global__ __launch_bounds__(NT) void k_grad(GradBatch batch, Taps g, Taps d) {
  extern __shared__ __align__(16) float lds[];
  constexpr bool kStatic = RG > 0;
  const int r = kStatic ? (RG > RD ? RG : RD) : max(g.r, d.r);
  const int inW = TW + 2 * r, inH = TH + 2 * r;
  float *in = lds;                // inH x inW
  float *rowd = in + inW * inH;   // inH x TW : rows filtered with G'
  float *rowg = rowd + inH * TW;  // inH x TW : rows filtered with G
  float *tapmem = rowg + inH * TW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  int ji = 0;
  for (int i = 1; i < batch.n; ++i)
    if ((int)blockIdx.x >= batch.job[i].first_tile) ji = i;
  const GradJob J = batch.job[ji];
  const int tile = blockIdx.x - J.first_tile;
  const int x0 = (tile % J.tiles_x) * TW, y0 = (tile / J.tiles_x) * TH;
  const int w = J.w, h = J.h;

  const TapRef<RG> tg = make_taps<RG>(g, tapmem);
  const TapRef<RD> td = make_taps<RD>(d, tapmem + kMaxRadius + 1);

  stage_tile<(RG > RD ? RG : RD)>(J.src, w, w, h, x0, y0, r, in, inW);
  __syncthreads();
  for (int ty = wave; ty < inH; ty += NW) {
    const float *c = in + ty * inW + lane + r;
    rowd[ty * TW + lane] = corr<RD, -1>(c, 1, td);
    rowg[ty * TW + lane] = corr<RG, +1>(c, 1, tg);
  }
  __syncthreads();
  const int x = x0 + lane;
#pragma unroll
  for (int ty = wave; ty < TH; ty += NW) {
    const int y = y0 + ty;
    const float gx = corr<RG, +1>(rowd + (ty + r) * TW + lane, TW, tg);
    const float gy = corr<RD, -1>(rowg + (ty + r) * TW + lane, TW, td);
    if (x < w && y < h) {
      if (J.tex) {
        J.tex[(size_t)y * w + x] = Texel{in[(ty + r) * inW + lane + r], gx, gy};
      } else {
        J.gx[(size_t)y * w + x] = gx;
        J.gy[(size_t)y * w + x] = gy;
      }
    }
  }
}

// ---------------------------------------------------------------- register-blocked fast path
// The simple kernels above read LDS once per tap (44 ds_read_b32 per output pair of passes at
// R = 5) and are LDS-bandwidth bound.  For compile-time radii the passes are register blocked:
//  * row pass: a thread produces 4 adjacent outputs from one window of NG aligned ds_read_b128
//    (the halo is padded to a multiple of 4 columns so that every window is 16-byte aligned);
//  * column pass: a wave owns 4 consecutive rows, a lane one column: 4+2R ds_read_b32 feed a
//    sliding window that produces 4 outputs.
// Each output is still the same left-to-right fp32 expression (bit-identical results).
template <int R>
struct FastGeom {
  static constexpr int HP = (R + 3) / 4 * 4;     // halo padded to whole float4 groups
  static constexpr int IWP = TW + 2 * HP;        // staged row length (multiple of 4)
  static constexpr int IH = TH + 2 * R;
  static constexpr int O = HP - R;               // first needed float inside the first group
  static constexpr int NG = (O + 3 + 2 * R) / 4 + 1;  // float4 groups per 4-output window
  static constexpr int ROW_TASKS = IH * (TW / 4);
  static_assert(IWP <= 128 && IWP % 4 == 0, "two lane chunks per staged row");
  static_assert(IH % 2 == 0, "row_of_task16 hands out rows in pairs");
};

template <int R, int SYM, int N>
__device__ __forceinline__ float corr_win(const float (&v)[N], int c, const Taps &t) {
  float acc;
  if (SYM > 0) {
    acc = t.k[0] * v[c];
#pragma unroll
    for (int j = 1; j <= R; ++j) acc = acc + t.k[j] * (v[c + j] + v[c - j]);
  } else {
    acc = 0.0f;
#pragma unroll
    for (int j = 1; j <= R; ++j) acc = acc + t.k[j] * (v[c + j] - v[c - j]);
  }
  return acc;
}

// The same expression on two columns at once (x = one column, y = its neighbour): every operation
// is one v_pk_*_f32, each half rounded exactly as the scalar form rounds it.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int R, int N>
__device__ __forceinline__ v2f corr_win2(const v2f (&v)[N], int c, const Taps &t) {
  v2f acc = v2f{t.k[0], t.k[0]} * v[c];
#pragma unroll
  for (int j = 1; j <= R; ++j) acc = acc + v2f{t.k[j], t.k[j]} * (v[c + j] + v[c - j]);
  return acc;
}
// Both gradient column filters of one pixel at once: v = (rows filtered with G', rows filtered
// with G); x = symmetric taps g on v.x (gx), y = antisymmetric taps d on v.y (gy); K[j] = (g[j], d[j]).
// The y half starts from 0.0f and its pair differences come from the negate-source modifier of the
// packed add (a + (-b) is a - b bit for bit): the bits of corr_win<R, -1>.
__device__ __forceinline__ v2f pk_add_sub(v2f a, v2f b) {  // (a.x + b.x, a.y - b.y)
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int R, int N>
__device__ __forceinline__ v2f corr_win_gxgy(const v2f (&v)[N], int c, const v2f (&K)[R + 1]) {
  v2f acc = v2f{K[0].x * v[c].x, 0.0f};
#pragma unroll
  for (int j = 1; j <= R; ++j) acc = acc + K[j] * pk_add_sub(v[c + j], v[c - j]);
  return acc;
}

#ifdef GKLT_PYR_FMA
// taps k[1 .. R + 1] of the polyphase pyramid filter (see level_tile), k[0] unused
template <int R>
struct PolyTaps {
  float k[R + 2];
};
// sum_{k = 1 .. R + 1} h[k] * (v[c + k] + v[c + 1 - k]): the decimated pixel whose 2 x 2 block starts at c
template <int R, int N>
__device__ __forceinline__ float poly_win(const float (&v)[N], int c, const PolyTaps<R> &t) {
  float acc = t.k[1] * (v[c + 1] + v[c]);
#pragma unroll
  for (int k = 2; k <= R + 1; ++k) acc = acc + t.k[k] * (v[c + k] + v[c + 1 - k]);
  return acc;
}
#endif

template <int R, typename SrcT>
__device__ __forceinline__ void stage_fast(const SrcT *__restrict__ src, int stride, int w, int h,
                                           int x0, int y0, float *in) {
  using G = FastGeom<R>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS = (G::IH + NW - 1) / NW;
  const int sx0 = reflect101(x0 - G::HP + lane, w);
  const int sx1 = reflect101(x0 - G::HP + 64 + min(lane, G::IWP - 65), w);
  SrcT v0[ROWS], v1[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int ty = wave + i * NW;
    const int sy = reflect101(y0 - R + min(ty, G::IH - 1), h);
    const SrcT *row = src + (size_t)sy * stride;
    v0[i] = row[sx0];
    v1[i] = row[sx1];
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int ty = wave + i * NW;
    if (ty < G::IH) {
      in[ty * G::IWP + lane] = static_cast<float>(v0[i]);
      if (lane < G::IWP - 64) in[ty * G::IWP + 64 + lane] = static_cast<float>(v1[i]);
    }
  }
}

// NG float4 groups of an LDS row into registers, each as ONE ds_read_b128.  The loads are volatile on
// purpose: a window uses only part of its first and last group, and the compiler otherwise narrows
// those (and, with them, the rest) to ds_read2_b32 / ds_read_b64 -- whose lanes, 16 bytes apart, fall
// on a quarter of the banks: measured 24 LDS cycles per group read instead of 4, a third of the
// level-0 kernel's LDS time (profiles/r03_level0_lds_*.txt).
template <int NG>
__device__ __forceinline__ void load_groups(const float *p, float (&v)[4 * NG]) {
  typedef const volatile v4f __attribute__((address_space(3))) * lds_v4f_ptr;  // LDS, not a flat address
  lds_v4f_ptr q4 = (lds_v4f_ptr)p;
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const v4f f = q4[q];
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
}

// row of task t of a row pass over rows of 16 float4 groups: 32 consecutive tasks are two rows, and
// the row is chosen by the parity of the lane's quad index, so that each of ds_read_b128's lane
// groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) reads 16 contiguous groups of ONE row --
// all 64 banks once, whatever the row stride -- and the 8-lane groups of ds_write_b128 stay
// conflict-free (tools/lds_model.py).  Needs an even number of rows.
__device__ __forceinline__ int row_of_task16(int task) {
  return 2 * (task >> 5) + (__builtin_popcount((task >> 2) & 7) & 1);
}

// loads the 4-output window of row task (ty, g) into registers
template <int R>
__device__ __forceinline__ void load_window(const float *in, int ty, int g,
                                            float (&v)[4 * FastGeom<R>::NG]) {
  using G = FastGeom<R>;
  load_groups<G::NG>(in + ty * G::IWP + 4 * g, v);
}

template <typename SrcT, int MODE, int R>
__global__ __launch_bounds__(NT) void k_smooth_fast(const SrcT *__restrict__ src, int stride, int w,
                                                    int h, Taps g, float *__restrict__ dst, int w2,
                                                    int h2) {
  using G = FastGeom<R>;
  extern __shared__ __align__(16) float lds[];
  float *in = lds;                    // IH x IWP
  float *row = in + G::IH * G::IWP;   // IH x TW
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  stage_fast<R>(src, stride, w, h, x0, y0, in);
  __syncthreads();
#pragma unroll
  for (int task = threadIdx.x; task < G::ROW_TASKS; task += NT) {
    const int ty = row_of_task16(task), gq = task & 15;
    float v[4 * G::NG];
    load_window<R>(in, ty, gq, v);
    float4 o;
    o.x = corr_win<R, +1>(v, G::O + R + 0, g);
    o.y = corr_win<R, +1>(v, G::O + R + 1, g);
    o.z = corr_win<R, +1>(v, G::O + R + 2, g);
    o.w = corr_win<R, +1>(v, G::O + R + 3, g);
    *reinterpret_cast<float4 *>(row + ty * TW + 4 * gq) = o;
  }
  __syncthreads();
  float v[4 + 2 * R], o[4];
#pragma unroll
  for (int i = 0; i < 4 + 2 * R; ++i) v[i] = row[(4 * wave + i) * TW + lane];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = corr_win<R, +1>(v, R + j, g);
  if (MODE == 0) {
    const int x = x0 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + 4 * wave + j;
      if (x < w && y < h) dst[(size_t)y * w + x] = o[j];
    }
  } else {
    // 2x2 mean in the order of include/gklt_order.h: the odd neighbour comes from lane ^ 1
    const int x = x0 / 2 + (lane >> 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const float a = o[2 * jj], c = o[2 * jj + 1];
      const float b = __shfl_xor(a, 1), d = __shfl_xor(c, 1);
      const float m = GKLT_MEAN2X2(a, b, c, d);
      const int y = y0 / 2 + 2 * wave + jj;
      if (!(lane & 1) && x < w2 && y < h2) dst[(size_t)y * w2 + x] = m;
    }
  }
}

template <int R>
__global__ __launch_bounds__(NT) void k_grad_fast(GradBatch batch, Taps g, Taps d) {
  using G = FastGeom<R>;
  extern __shared__ __align__(16) float lds[];
  float *in = lds;                      // IH x IWP
  float *rowd = in + G::IH * G::IWP;    // IH x TW : rows filtered with G'
  float *rowg = rowd + G::IH * TW;      // IH x TW : rows filtered with G
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  int ji = 0;
  for (int i = 1; i < batch.n; ++i)
    if ((int)blockIdx.x >= batch.job[i].first_tile) ji = i;
  const GradJob J = batch.job[ji];
  const int tile = blockIdx.x - J.first_tile;
  const int x0 = (tile % J.tiles_x) * TW, y0 = (tile / J.tiles_x) * TH;
  const int w = J.w, h = J.h;

  stage_fast<R>(J.src, w, w, h, x0, y0, in);
  __syncthreads();
#pragma unroll
  for (int task = threadIdx.x; task < G::ROW_TASKS; task += NT) {
    const int ty = row_of_task16(task), gq = task & 15;
    float v[4 * G::NG];
    load_window<R>(in, ty, gq, v);
    float4 od, og;
    od.x = corr_win<R, -1>(v, G::O + R + 0, d); og.x = corr_win<R, +1>(v, G::O + R + 0, g);
    od.y = corr_win<R, -1>(v, G::O + R + 1, d); og.y = corr_win<R, +1>(v, G::O + R + 1, g);
    od.z = corr_win<R, -1>(v, G::O + R + 2, d); og.z = corr_win<R, +1>(v, G::O + R + 2, g);
    od.w = corr_win<R, -1>(v, G::O + R + 3, d); og.w = corr_win<R, +1>(v, G::O + R + 3, g);
    *reinterpret_cast<float4 *>(rowd + ty * TW + 4 * gq) = od;
    *reinterpret_cast<float4 *>(rowg + ty * TW + 4 * gq) = og;
  }
  __syncthreads();
  float vd[4 + 2 * R], vg[4 + 2 * R];
#pragma unroll
  for (int i = 0; i < 4 + 2 * R; ++i) {
    vd[i] = rowd[(4 * wave + i) * TW + lane];
    vg[i] = rowg[(4 * wave + i) * TW + lane];
  }
  const int x = x0 + lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int y = y0 + 4 * wave + j;
    const float gx = corr_win<R, +1>(vd, R + j, g);
    const float gy = corr_win<R, -1>(vg, R + j, d);
    if (x < w && y < h) {
      if (J.tex) {
        J.tex[(size_t)y * w + x] = Texel{in[(4 * wave + j + R) * G::IWP + G::HP + lane], gx, gy};
      } else {
        J.gx[(size_t)y * w + x] = gx;
        J.gy[(size_t)y * w + x] = gy;
      }
    }
  }
}

template <int R>
constexpr size_t fast_smooth_lds() {
  return sizeof(float) * (size_t)(FastGeom<R>::IH * FastGeom<R>::IWP + FastGeom<R>::IH * TW);
}
template <int R>
constexpr size_t fast_grad_lds() {
  return sizeof(float) * (size_t)(FastGeom<R>::IH * FastGeom<R>::IWP + 2 * FastGeom<R>::IH * TW);
}

// ---------------------------------------------------------------- fused level kernel -------
// One launch per pyramid level does everything that level needs:
//   level 0 : u8 tile -> smooth(sigma0) -> I (in LDS only)        (pyramid.cpp:59-61)
//   level l : planar I_l tile -> I (in LDS)
//   then    : gradients of I -> texels (I, gx, gy)                (pyramid.cpp:72-83)
//             smooth(sigma_pyr) of I -> 2x2 mean -> planar I_{l+1} (pyramid.cpp:62-66)
// so a level's image is read from HBM once, level 0's fp32 plane is never written, and a build is
// nlevels launches.  I is produced with a halo of HI = max(RP, RG) pixels around the tile; at
// level 0 the halo pixels are smoothed from the reflected u8 window, which gives the same bits
// as reflecting the smoothed plane because the taps are symmetric and each pair sum
// (s[+j] + s[-j]) is commutative.  All passes are the register-blocked ones described above.

// The fused level kernel works on 64 x 32 tiles with 8 waves (4 output rows per wave): compared
// with 64 x 16 the halo rows (2 * (HI + R0) = 14 at C2) are shared by twice as many outputs, which
// takes ~17 % of the kernel's VALU work away.
// Workgroups of 8 waves (4 output rows per wave) for single-frame launches, whose dependent chain
// of four launches wants short workgroups, and of 4 waves (8 rows per wave) for batched launches:
// those run next to a tracker launch that holds 16 of a CU's 32 wave slots, and three 4-wave
// workgroups fit beside it where only two 8-wave ones do (LDS allows three either way): level 0
// 244 -> 219 us per 16-frame launch while overlapped, C2 191 -> 201 M.
constexpr int LTH = 32;

template <int R0, int HI, int LNT>
struct LevelGeom {
  static constexpr int HIP = (HI + 3) / 4 * 4;    // halo of I padded to whole float4 groups
  static constexpr int AW = TW + 2 * HIP;         // row length of A and I
  static constexpr int IHh = LTH + 2 * HI;        // rows of I
  static constexpr int R0P = (R0 + 3) / 4 * 4;
  static constexpr int SW = AW + 2 * R0P;         // staged source row length (R0 > 0)
  static constexpr int SH = IHh + 2 * R0;         // staged source rows       (R0 > 0)
  static constexpr int OA = R0P - R0;             // first needed float of a level-0 row window
  static constexpr int NGA = (OA + 3 + 2 * R0) / 4 + 1;
  static constexpr int OI = HIP - HI;             // first needed float of an I row window
  static constexpr int NGI = (OI + 3 + 2 * HI) / 4 + 1;
  static constexpr int STRIPS = LNT / (AW / 2);    // column pass of level 0: strips per column pair
  static constexpr int SR = (IHh + STRIPS - 1) / STRIPS;
  static_assert(SW <= 128 && AW % 4 == 0, "two lane chunks per staged row");
  static_assert(IHh % 2 == 0, "row_of_task16 hands out rows in pairs");
  static_assert(4 * NGA - 4 <= 2 * R0P && 60 + 4 * NGI <= AW, "row windows stay inside a row");
};

__device__ __forceinline__ int reflect101_near(int i, int n, bool wide) {
  if (wide) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
  }
  return reflect101(i, n);
}

// out[ROWS][WIDTH] <- src[reflect(ys + ty)][reflect(xs + tx)], every load issued before the
// first LDS store.
template <int ROWS, int WIDTH, int LNT, typename SrcT>
__device__ __forceinline__ void stage_region(const SrcT *__restrict__ src, int stride, int w, int h,
                                             int xs, int ys, float *out) {
  // the wave index as a scalar: row indices, reflections and row pointers then live in SGPRs
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LNW = LNT / 64;
  constexpr int PER = (ROWS + LNW - 1) / LNW;
  // Images of at least 192 px per side (uniform test): every staged index is less than one
  // image size outside [0, n), so one reflection (two selects) suffices; smaller images take
  // the general loop.
  const bool wide = w >= 192 && h >= 192;
  const int sx0 = reflect101_near(xs + lane, w, wide);
  const int sx1 = reflect101_near(xs + 64 + min(lane, WIDTH - 65), w, wide);
  SrcT v0[PER], v1[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ty = wave + i * LNW;
    const int sy = reflect101_near(ys + min(ty, ROWS - 1), h, wide);
    const SrcT *row = src + (size_t)sy * stride;
    v0[i] = row[sx0];
    v1[i] = row[sx1];
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ty = wave + i * LNW;
    if (ty < ROWS) {
      out[ty * WIDTH + lane] = static_cast<float>(v0[i]);
      if (lane < WIDTH - 64) out[ty * WIDTH + 64 + lane] = static_cast<float>(v1[i]);
    }
  }
}

// The same for a u8 source whose staged columns all lie inside the image (no reflection in x) and
// whose rows are 4-byte aligned: one dword (4 pixels) per load and one ds_write_b128 per store
// instead of four byte loads, four conversions' worth of addressing and four ds_write_b32.
template <int ROWS, int WIDTH, int LNT>
__device__ __forceinline__ void stage_region_u8x4(const uint8_t *__restrict__ src, int stride, int h,
                                                  int xs, int ys, float *out) {
  static_assert(WIDTH % 4 == 0, "whole dwords per staged row");
  constexpr int QPR = WIDTH / 4, TASKS = ROWS * QPR, PER = (TASKS + LNT - 1) / LNT;
  unsigned v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int t = min((int)threadIdx.x + i * LNT, TASKS - 1);
    const int row = t / QPR, q = t - row * QPR;
    const int sy = reflect101_near(ys + row, h, h >= 192);
    v[i] = *reinterpret_cast<const unsigned *>(src + (size_t)sy * stride + xs + 4 * q);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int t = (int)threadIdx.x + i * LNT;
    if (t < TASKS) {
      float4 f;
      f.x = (float)(v[i] & 0xffu);
      f.y = (float)((v[i] >> 8) & 0xffu);
      f.z = (float)((v[i] >> 16) & 0xffu);
      f.w = (float)(v[i] >> 24);
      *reinterpret_cast<float4 *>(out + 4 * t) = f;  // out[row * WIDTH + 4 * q], WIDTH = 4 * QPR
    }
  }
}

// blockIdx.z = frame of the batch (lookahead builds put several frames into one launch)
struct LevelIO {
  const void *src[GKLT_MAX_LOOKAHEAD];
  Texel *tex[GKLT_MAX_LOOKAHEAD];
  float *down[GKLT_MAX_LOOKAHEAD];
};

// one 64 x 32 tile of one level (origin x0, y0): the body of k_level and of k_level_chain
// THROUGH: the decimated image is stored write-through at agent scope (k_level_chain: tiles of the
// same launch on other XCDs read it)
template <typename SrcT, int R0, int RP, int RG, bool DOWN, int LNT, bool THROUGH = false>
__device__ __forceinline__ void level_tile(const SrcT *__restrict__ src, Texel *__restrict__ tex,
                                           float *__restrict__ down, int stride, int w, int h,
                                           const Taps &s0, const Taps &pyr, const Taps &g, const Taps &d,
                                           float *__restrict__ img_out, int w2, int h2, int x0, int y0) {
  constexpr int HI = DOWN ? (RP > RG ? RP : RG) : RG;
  using G = LevelGeom<R0, HI, LNT>;
  constexpr int RPW = LTH / (LNT / 64);  // output rows per wave
  constexpr int GH = LTH + 2 * RG;  // rows of the gradient row pass
  constexpr int GOFF = HI - RG;    // its first row inside I
  extern __shared__ __align__(16) float lds[];
  float *I = lds;                        // IHh x AW
  float *scratch = I + G::IHh * G::AW;   // level-0 phase: S, A; afterwards: rowd, rowg, rowp
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef GKLT_PYR_FMA
  // The contracted build folds the 2 x 2 mean into the pyramid filter: the decimated pixel is
  // sum_m h[m] I(2x' + m), m = -RP .. RP + 1, h[m] = (g[m] + g[m-1]) / 2 = h[1 - m] -- RP + 1 distinct
  // taps, evaluated at the decimated sites only (a quarter of the sites, half of them per pass)
  [[maybe_unused]] PolyTaps<RP> hp;
  if constexpr (DOWN) {
#pragma unroll
    for (int k = 1; k <= RP; ++k) hp.k[k] = 0.5f * (pyr.k[k] + pyr.k[k - 1]);
    hp.k[RP + 1] = 0.5f * pyr.k[RP];
  }
#endif

  if constexpr (R0 > 0) {
    float *S = scratch;               // SH x SW source window
    float *A = S + G::SH * G::SW;     // SH x AW rows smoothed
    const int xs = x0 - G::HIP - G::R0P;
    if constexpr (sizeof(SrcT) == 1) {
      // uniform per block: interior columns and dword-aligned rows take the 4-pixel loads
      if (xs >= 0 && xs + G::SW <= w && ((stride | (int)(reinterpret_cast<size_t>(src) & 3)) & 3) == 0)
        stage_region_u8x4<G::SH, G::SW, LNT>(reinterpret_cast<const uint8_t *>(src), stride, h, xs,
                                        y0 - HI - R0, S);
      else
        stage_region<G::SH, G::SW, LNT>(src, stride, w, h, xs, y0 - HI - R0, S);
    } else {
      stage_region<G::SH, G::SW, LNT>(src, stride, w, h, xs, y0 - HI - R0, S);
    }
    __syncthreads();
    constexpr int GPR = G::AW / 4;
#pragma unroll
    for (int task = threadIdx.x; task < G::SH * GPR; task += LNT) {
      const int ty = task / GPR, gq = task - ty * GPR;
      float v[4 * G::NGA];
      load_groups<G::NGA>(S + ty * G::SW + 4 * gq, v);
      float4 o;
      o.x = corr_win<R0, +1>(v, G::OA + R0 + 0, s0);
      o.y = corr_win<R0, +1>(v, G::OA + R0 + 1, s0);
      o.z = corr_win<R0, +1>(v, G::OA + R0 + 2, s0);
      o.w = corr_win<R0, +1>(v, G::OA + R0 + 3, s0);
      *reinterpret_cast<float4 *>(A + ty * G::AW + 4 * gq) = o;
    }
    __syncthreads();
    if ((int)threadIdx.x < G::STRIPS * (G::AW / 2)) {
      // a lane owns a PAIR of adjacent columns (one ds_read_b64 per row), so that every add and
      // multiply of the pass is one packed-fp32 instruction on the pair
      const int strip = threadIdx.x / (G::AW / 2), cp = threadIdx.x - strip * (G::AW / 2);
      const v2f *A2 = reinterpret_cast<const v2f *>(A) + cp;
      v2f v[G::SR + 2 * R0];
#pragma unroll
      for (int i = 0; i < G::SR + 2 * R0; ++i) v[i] = A2[min(strip * G::SR + i, G::SH - 1) * (G::AW / 2)];
#pragma unroll
      for (int j = 0; j < G::SR; ++j) {
        const int rr = strip * G::SR + j;
        const v2f o = corr_win2<R0>(v, R0 + j, s0);
        if (rr < G::IHh) reinterpret_cast<v2f *>(I)[rr * (G::AW / 2) + cp] = o;
      }
    }
    __syncthreads();
  } else {
    stage_region<G::IHh, G::AW, LNT>(src, stride, w, h, x0 - G::HIP, y0 - HI, I);
    __syncthreads();
  }

  // rows filtered with G' (even rows of rowdg) and with G (odd rows), interleaved row by row: the
  // column pass reads the two values of a pixel with one ds_read2st64_b32 into one register pair
  float *rowdg = scratch;               // 2 GH x TW
  float *rowp = rowdg + 2 * GH * TW;    // IHh x TW : rows filtered with the pyramid kernel (DOWN)
#pragma unroll
  for (int task = threadIdx.x; task < G::IHh * (TW / 4); task += LNT) {
    const int ty = row_of_task16(task), gq = task & 15;
    float v[4 * G::NGI];
    load_groups<G::NGI>(I + ty * G::AW + 4 * gq, v);
    constexpr int C = G::OI + HI;
    if constexpr (DOWN) {
#ifdef GKLT_PYR_FMA
      // polyphase: the filter is evaluated at the two decimated columns of the group only
      float2 op;
      op.x = poly_win<RP>(v, C + 0, hp);
      op.y = poly_win<RP>(v, C + 2, hp);
      *reinterpret_cast<float2 *>(rowp + ty * (TW / 2) + 2 * gq) = op;
#else
      float4 op;
      op.x = corr_win<RP, +1>(v, C + 0, pyr);
      op.y = corr_win<RP, +1>(v, C + 1, pyr);
      op.z = corr_win<RP, +1>(v, C + 2, pyr);
      op.w = corr_win<RP, +1>(v, C + 3, pyr);
      *reinterpret_cast<float4 *>(rowp + ty * TW + 4 * gq) = op;
#endif
    }
    const int tg = ty - GOFF;
    if (tg >= 0 && tg < GH) {
      float4 od, og;
      od.x = corr_win<RG, -1>(v, C + 0, d); og.x = corr_win<RG, +1>(v, C + 0, g);
      od.y = corr_win<RG, -1>(v, C + 1, d); og.y = corr_win<RG, +1>(v, C + 1, g);
      od.z = corr_win<RG, -1>(v, C + 2, d); og.z = corr_win<RG, +1>(v, C + 2, g);
      od.w = corr_win<RG, -1>(v, C + 3, d); og.w = corr_win<RG, +1>(v, C + 3, g);
      *reinterpret_cast<float4 *>(rowdg + 2 * tg * TW + 4 * gq) = od;
      *reinterpret_cast<float4 *>(rowdg + (2 * tg + 1) * TW + 4 * gq) = og;
    }
  }
  __syncthreads();

  {  // gradient column pass -> texels
    // (G'-filtered, G-filtered) of a row as one register pair: both column filters run as packed fp32
    v2f vdg[RPW + 2 * RG], K[RG + 1];
#pragma unroll
    for (int j = 0; j <= RG; ++j) {
      float kg = g.k[j], kd = d.k[j];
      asm("" : "+v"(kg), "+v"(kd));  // taps in VGPRs: the pair is then a register sequence
      K[j] = v2f{kg, kd};
    }
#pragma unroll
    for (int i = 0; i < RPW + 2 * RG; ++i)
      vdg[i] = v2f{rowdg[2 * (RPW * wave + i) * TW + lane], rowdg[(2 * (RPW * wave + i) + 1) * TW + lane]};
    const int x = x0 + lane;
    // row pointers are wave-uniform (SGPRs); the lane's part of the address is one 32-bit offset
    const unsigned xoff = (unsigned)lane * (unsigned)sizeof(Texel);
    if (x < w) {
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int y = y0 + RPW * wave + j;
        const v2f gxy = corr_win_gxgy<RG>(vdg, RG + j, K);
        const float ic = I[(RPW * wave + j + HI) * G::AW + G::HIP + lane];
        if (y < h) {
          // streaming store: the texels are next read by a tracker launch at least one build
          // later, and must not evict the window footprints a concurrent tracker gathers from L2
          char *trow = reinterpret_cast<char *>(tex + (size_t)y * w + x0);
          asm("" : "+s"(trow));  // keeps the row pointer in SGPRs: store address = SGPR base + lane offset
          typedef float __attribute__((address_space(1))) * global_float_ptr;  // global, not a flat address
          global_float_ptr t = (global_float_ptr)(trow + xoff);  // the three stores merge into one dwordx3
          __builtin_nontemporal_store(ic, t);
          __builtin_nontemporal_store(gxy.x, t + 1);
          __builtin_nontemporal_store(gxy.y, t + 2);
          if (img_out) img_out[(size_t)y * w + x] = ic;
        }
      }
    }
  }
#ifdef GKLT_PYR_FMA
  if constexpr (DOWN) {  // polyphase column pass at the decimated rows: lane = (row half, column)
    constexpr int DPL = RPW / 4;  // decimated rows per lane: a wave owns RPW / 2 of them, 32 columns wide
    static_assert(RPW % 4 == 0, "two lane halves share a wave's decimated rows");
    const int xq = lane & 31, half = lane >> 5;
    float v[2 * DPL + 2 * RP];
    // decimated row j of this lane = tile row 2 * (RPW/2 * wave + DPL * half + j): taps at I rows
    // r0 - RP .. r0 + RP + 1 with r0 = that tile row + HI
    const int rbase = RPW * wave + 2 * DPL * half + HI - RP;
#pragma unroll
    for (int i = 0; i < 2 * DPL + 2 * RP; ++i) v[i] = rowp[(rbase + i) * (TW / 2) + xq];
    const int x = x0 / 2 + xq;
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
      const float m = poly_win<RP>(v, RP + 2 * j, hp);
      const int y = y0 / 2 + RPW / 2 * wave + DPL * half + j;
      if (x < w2 && y < h2) down[(size_t)y * w2 + x] = m;
    }
  }
#else
  if constexpr (DOWN) {  // pyramid column pass -> 2x2 mean -> next level
    // a lane owns a pair of adjacent columns and half of its wave's rows: packed fp32 throughout,
    // and the four pixels of a 2 x 2 block are in one lane (no cross-lane traffic for the mean)
    constexpr int RL = RPW / 2;  // rows per lane
    static_assert(RL % 2 == 0, "whole 2 x 2 blocks per lane");
    const int xq = lane & 31, half = lane >> 5;
    const v2f *P2 = reinterpret_cast<const v2f *>(rowp) + (RPW * wave + RL * half + HI - RP) * (TW / 2) + xq;
    v2f v[RL + 2 * RP], o[RL];
#pragma unroll
    for (int i = 0; i < RL + 2 * RP; ++i) v[i] = P2[i * (TW / 2)];
#pragma unroll
    for (int j = 0; j < RL; ++j) o[j] = corr_win2<RP>(v, RP + j, pyr);
    const int x = x0 / 2 + xq;
#pragma unroll
    for (int jj = 0; jj < RL / 2; ++jj) {
      const float m = GKLT_MEAN2X2(o[2 * jj].x, o[2 * jj].y, o[2 * jj + 1].x, o[2 * jj + 1].y);
      const int y = y0 / 2 + RPW / 2 * wave + RL / 2 * half + jj;
      if (x < w2 && y < h2) {
        if constexpr (THROUGH)
          __hip_atomic_store(reinterpret_cast<unsigned *>(down + (size_t)y * w2 + x), __float_as_uint(m),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          down[(size_t)y * w2 + x] = m;
      }
    }
  }
#endif
}

template <typename SrcT, int R0, int RP, int RG, bool DOWN, int LNT>
__global__ __launch_bounds__(LNT) void k_level(LevelIO io, int stride, int w, int h, Taps s0,
                                              Taps pyr, Taps g, Taps d,
                                              float *__restrict__ img_out, int w2, int h2) {
  level_tile<SrcT, R0, RP, RG, DOWN, LNT>(static_cast<const SrcT *>(io.src[blockIdx.z]), io.tex[blockIdx.z],
                                          io.down[blockIdx.z], stride, w, h, s0, pyr, g, d, img_out, w2, h2,
                                          blockIdx.x * TW, blockIdx.y * LTH);
}

// ---- levels 1 .. L-1 of one pyramid in ONE launch.
// A frame-at-a-time build is a chain of dependent launches (level l+1 reads the image level l
// wrote), and at 1080p the three coarse levels are 5-6 us each, mostly launch latency and tail.
// Here every tile of every coarse level is a workgroup of one launch, ordered level by level; a
// tile of level l >= 2 waits until the (at most 4 x 4) tiles of level l-1 that produce its haloed
// input have published their part of the decimated image:
//   producer: write-through (agent-scope) stores of the decimated image -> wait for them ->
//             workgroup barrier -> one lane stores the launch's epoch into the tile's flag.  (A
//             plain store + agent-scope release fence is the textbook form, but that fence writes
//             back the whole L2 -- with level 0's texels in it -- in every wavefront: the
//             frame-at-a-time step went from 44 to 106 us with it);
//   consumer: up to 16 lanes poll one flag each (agent-scope loads) -> barrier -> agent-scope
//             acquire fence (invalidates this XCD's stale lines of the image) -> staging loads.
// Workgroups are dispatched in index order and a tile only waits for tiles of lower index, which
// wait for nothing later than themselves: every wait ends.  The wait is bounded all the same: a
// tile that gives up raises *err (reported by the next synchronising call) instead of hanging.
struct ChainArgs {
  int nlev, nframes;                 // levels in the launch (pyramid levels 1 .. nlev), frames of the batch
  // per launch level: first workgroup of the level (all frames), tiles of ONE frame, first flag of
  // the level inside a pyramid's flag array, tiles per row / column, size
  int first_block[GKLT_MAX_LEVELS], tiles[GKLT_MAX_LEVELS], first_flag[GKLT_MAX_LEVELS];
  int tiles_x[GKLT_MAX_LEVELS], tiles_y[GKLT_MAX_LEVELS], w[GKLT_MAX_LEVELS], h[GKLT_MAX_LEVELS];
  const float *img[GKLT_MAX_LOOKAHEAD][GKLT_MAX_LEVELS];  // img[f][i + 1] is where level i's decimated image goes
  Texel *tex[GKLT_MAX_LOOKAHEAD][GKLT_MAX_LEVELS];
  unsigned *flags[GKLT_MAX_LOOKAHEAD];  // one word per tile of the pyramid's levels 1 .. L-1
  unsigned epoch[GKLT_MAX_LOOKAHEAD];
  unsigned *err;
};

// Workgroups are ordered level by level (all frames of level 1, then all frames of level 2, ...):
// by the time a tile of level l is dispatched its producers were dispatched a whole level ago, and
// in a batch the tiles of the next level fill the chip while the stragglers of this one finish.
template <int RP, int RG, int LNT>
__global__ __launch_bounds__(LNT) void k_level_chain(ChainArgs a, Taps pyr, Taps g, Taps d) {
  int li = 0;
  for (int i = 1; i < a.nlev; ++i)
    if ((int)blockIdx.x >= a.first_block[i]) li = i;
  const int local = blockIdx.x - a.first_block[li];
  const int f = local / a.tiles[li], tile = local - f * a.tiles[li];
  const int tx = tile % a.tiles_x[li], ty = tile / a.tiles_x[li];
  const int x0 = tx * TW, y0 = ty * LTH, w = a.w[li], h = a.h[li];
  unsigned *const flags = a.flags[f];
  const unsigned epoch = a.epoch[f];
  if (li > 0) {
    // producers: tiles of the level above whose 32 x 16 outputs meet this tile's haloed input
    // (columns x0 - 8 .. x0 + 71, rows y0 - 5 .. y0 + 36, reflected into the image: inside it)
    constexpr int HIP = (((RP > RG ? RP : RG) + 3) / 4) * 4, HI = RP > RG ? RP : RG;
    const int px0 = max(x0 - HIP, 0) / (TW / 2), px1 = min(min(x0 + TW + HIP, w) - 1, w - 1) / (TW / 2);
    const int py0 = max(y0 - HI, 0) / (LTH / 2), py1 = min(min(y0 + LTH + HI, h) - 1, h - 1) / (LTH / 2);
    const int npx = px1 - px0 + 1, np = npx * (py1 - py0 + 1);
    if ((int)threadIdx.x < np) {
      const int pxx = min(px0 + (int)threadIdx.x % npx, a.tiles_x[li - 1] - 1);
      const int pyy = min(py0 + (int)threadIdx.x / npx, a.tiles_y[li - 1] - 1);
      const unsigned *flag = flags + a.first_flag[li - 1] + pyy * a.tiles_x[li - 1] + pxx;
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        if (++spins > (1 << 20)) {  // ~ a second: something is badly wrong; do not hang the GPU
          *a.err = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const bool down = li + 1 < a.nlev;
  if (down)
    level_tile<float, 0, RP, RG, true, LNT, true>(a.img[f][li], a.tex[f][li], const_cast<float *>(a.img[f][li + 1]), w,
                                                  w, h, g, pyr, g, d, nullptr, a.w[li + 1], a.h[li + 1], x0, y0);
  else
    level_tile<float, 0, RP, RG, false, LNT>(a.img[f][li], a.tex[f][li], nullptr, w, w, h, g, g, g, d, nullptr, 0, 0, x0, y0);
  if (down) {
    __builtin_amdgcn_s_waitcnt(0);  // this wavefront's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(flags + a.first_flag[li] + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int R0, int RP, int RG, bool DOWN>
constexpr size_t level_lds() {
  constexpr int HI = DOWN ? (RP > RG ? RP : RG) : RG;
  using G = LevelGeom<R0, HI, 512>;  // the sizes used here do not depend on the workgroup size
  constexpr size_t p1 = R0 > 0 ? (size_t)G::SH * (G::SW + G::AW) : 0;
  constexpr size_t p2 = 2 * (size_t)(LTH + 2 * RG) * TW + (DOWN ? (size_t)G::IHh * TW : 0);
  return sizeof(float) * ((size_t)G::IHh * G::AW + (p1 > p2 ? p1 : p2));
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

// cv::Mat::convertTo(CV_8UC1) of the image plane (System.cpp:403): saturate_cast<uchar>(cvRound(v)),
// cvRound = round half to even
__global__ void k_tex_to_u8(const Texel *__restrict__ tex, int n, uint8_t *__restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = __float2int_rn(tex[i].x);
  dst[i] = (uint8_t)min(max(v, 0), 255);
}

__global__ void k_tex_insert(Texel *__restrict__ tex, int n, int comp,
                             const float *__restrict__ src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Texel t = tex[i];
  const float v = src[i];
  if (comp == 0) t.x = v; else if (comp == 1) t.y = v; else t.z = v;
  tex[i] = t;
}

inline dim3 tile_grid(int w, int h) { return dim3((w + TW - 1) / TW, (h + TH - 1) / TH); }
inline size_t smooth_lds(int r) {
  return sizeof(float) * ((size_t)(TW + 2 * r) * (TH + 2 * r) + (size_t)(TH + 2 * r) * TW +
                          (kMaxRadius + 1));
}
inline size_t grad_lds(int r) {
  return sizeof(float) * ((size_t)(TW + 2 * r) * (TH + 2 * r) + 2 * (size_t)(TH + 2 * r) * TW +
                          2 * (kMaxRadius + 1));
}

int allow_lds(const void *kernel, size_t bytes) {
  if (bytes > 160 * 1024) {
    set_error("filter radius needs %zu bytes of LDS (> 160 KiB)", bytes);
    return GKLT_EINVAL;
  }
  if (bytes > 48 * 1024)
    GKLT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

template <typename SrcT, int MODE>
int launch_smooth_any(gklt_ctx *ctx, hipStream_t s, int prof_id, const SrcT *src, int stride,
                      int w, int h, const Taps &g, float *dst, int w2, int h2) {
  const dim3 grid = tile_grid(w, h), block(NT);
#define GKLT_SMOOTH_CASE(RR)                                                                  \
  case RR: {                                                                                  \
    ProfScope ps(ctx, s, prof_id);                                                            \
    hipLaunchKernelGGL((k_smooth_fast<SrcT, MODE, RR>), grid, block, fast_smooth_lds<RR>(), s, \
                       src, stride, w, h, g, dst, w2, h2);                                    \
  } break;
  switch (g.r <= 6 ? g.r : 0) {
    GKLT_SMOOTH_CASE(1)
    GKLT_SMOOTH_CASE(2)
    GKLT_SMOOTH_CASE(3)
    GKLT_SMOOTH_CASE(4)
    GKLT_SMOOTH_CASE(5)
    GKLT_SMOOTH_CASE(6)
    default: {
      const size_t lds = smooth_lds(g.r);
      auto kern = k_smooth<SrcT, MODE, 0>;
      if (int rc = allow_lds(reinterpret_cast<const void *>(kern), lds)) return rc;
      ProfScope ps(ctx, s, prof_id);
      hipLaunchKernelGGL(kern, grid, block, lds, s, src, stride, w, h, g, dst, w2, h2);
    }
  }
#undef GKLT_SMOOTH_CASE
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_grad_batch(gklt_ctx *ctx, hipStream_t s, GradBatch &b, const Taps &g, const Taps &d) {
  int tiles = 0;
  for (int i = 0; i < b.n; ++i) {
    GradJob &J = b.job[i];
    J.tiles_x = (J.w + TW - 1) / TW;
    J.first_tile = tiles;
    tiles += J.tiles_x * ((J.h + TH - 1) / TH);
  }
  if (tiles == 0) return 0;
  ProfScope ps(ctx, s, GKLT_K_GRAD);
  if (g.r == 3 && d.r == 3) {  // grad_sigma = 1.0, the only value the reference ever uses
    hipLaunchKernelGGL((k_grad_fast<3>), dim3(tiles), dim3(NT), fast_grad_lds<3>(), s, b, g, d);
  } else if (g.r == 2 && d.r == 2) {
    hipLaunchKernelGGL((k_grad_fast<2>), dim3(tiles), dim3(NT), fast_grad_lds<2>(), s, b, g, d);
  } else {
    const size_t lds = grad_lds(std::max(g.r, d.r));
    auto kern = k_grad<0, 0>;
    if (int rc = allow_lds(reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT), lds, s, b, g, d);
  }
  GKLT_HIP(hipGetLastError());
  return 0;
}

template <typename SrcT, int R0, bool DOWN>
int launch_level_inst(gklt_ctx *ctx, hipStream_t s, int prof_id, const LevelIO &io, int nframes,
                      int stride, int w, int h, const Taps &s0, const Taps &pyr, const Taps &g,
                      const Taps &d, float *img_out, int w2, int h2) {
  constexpr size_t lds = level_lds<R0, 5, 3, DOWN>();
  static_assert(lds <= 64 * 1024, "fused level kernel fits the LDS a launch gets without opt-in");
  dim3 grid((w + TW - 1) / TW, (h + LTH - 1) / LTH, nframes);
  ProfScope ps(ctx, s, prof_id);
  if (nframes > 1)  // batched: 4-wave workgroups (see LTH)
    hipLaunchKernelGGL((k_level<SrcT, R0, 5, 3, DOWN, 256>), grid, dim3(256), lds, s, io, stride, w, h,
                       s0, pyr, g, d, img_out, w2, h2);
  else
    hipLaunchKernelGGL((k_level<SrcT, R0, 5, 3, DOWN, 512>), grid, dim3(512), lds, s, io, stride, w, h,
                       s0, pyr, g, d, img_out, w2, h2);
  GKLT_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// Radii the fused level kernel is instantiated for: grad_sigma = 1.0 (R 3), pyramid sigma 1.8
// (subsampling 2 * 0.9, R 5), level-0 smoothing radius 1..6 (windows up to 29 px; the as-shipped 21x21 has radius 6).
#ifndef GKLT_PYR_FMA
bool level_fusable(const Taps *s0, const Taps *pyr, const Taps &g, const Taps &d) {
  if (g.r != 3 || d.r != 3) return false;
  if (pyr && pyr->r != 5) return false;
  if (s0 && (s0->r < 1 || s0->r > 6)) return false;
  return true;
}
#else
// This translation unit is compiled a second time with -DGKLT_PYR_FMA -ffp-contract=fast
// (gslam_amd/build.py): the same fused level kernels with mul+add contracted into FMAs, exported as
// launch_level_fma and selected by gklt_ctx_set_pyramid_impl(ctx, 4).  Results are NOT the
// oracle's bits any more; they stay inside north_star's tolerance (1e-4 relative on every plane:
// tests/test_gpu_pyramid.py::test_contracted_level_kernels_within_tolerance).  Never the default.
#define launch_level launch_level_fma
#endif

// Level `level` of `n` pyramids in one launch (n <= GKLT_MAX_LOOKAHEAD).  Level 0 reads the u8
// frames src[i] (smoothing s0); level l > 0 reads py[i]->lv[l].img.  pyr == nullptr: no
// down-sampling (last level, or the next level is not an exact 2x decimation).  img0_out (single
// frame only) additionally materialises the planar level-0 image.
int launch_level(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, const uint8_t *const *src,
                 int n, int stride, int level, const Taps &s0, const Taps *pyr, const Taps &g,
                 const Taps &d, float *img0_out) {
  if (n < 1 || n > GKLT_MAX_LOOKAHEAD || (img0_out && n != 1)) return GKLT_EINVAL;
  // pyr_impl 3: the streaming form of the kernel (gklt_pyramid_stream.hip; same bits).  Alone it is
  // faster than the tile kernel on batched level-0 launches (131 vs 161 us per 16 C2 frames), but
  // next to the tracker launch a batch overlaps with it does not shorten the batch,
  // so the tile kernel stays the default.
#ifndef GKLT_PYR_FMA
  if (!img0_out && ctx->pyr_impl == 3 && level_streamable(py[0]->lv[level].w, py[0]->lv[level].h, level, s0))
    return launch_level_stream(ctx, s, py, src, n, stride, level, s0, pyr, g, d);
  if (ctx->pyr_impl == 4)  // contracted arithmetic (tolerance variant, see the #else branch above)
    return launch_level_fma(ctx, s, py, src, n, stride, level, s0, pyr, g, d, img0_out);
#endif
  LevelIO io{};
  const Level &L = py[0]->lv[level];
  int w2 = 0, h2 = 0;
  for (int i = 0; i < n; ++i) {
    const Level &Li = py[i]->lv[level];
    if (Li.w != L.w || Li.h != L.h) return GKLT_EINVAL;
    io.src[i] = level == 0 ? static_cast<const void *>(src[i]) : static_cast<const void *>(Li.img);
    io.tex[i] = Li.tex;
    io.down[i] = pyr ? py[i]->lv[level + 1].img : nullptr;
  }
  if (pyr) { w2 = py[0]->lv[level + 1].w; h2 = py[0]->lv[level + 1].h; }
  const Taps &p = pyr ? *pyr : g;
  if (level > 0) {
    return pyr ? launch_level_inst<float, 0, true>(ctx, s, GKLT_K_LEVELN, io, n, L.w, L.w, L.h, g, p,
                                                   g, d, nullptr, w2, h2)
               : launch_level_inst<float, 0, false>(ctx, s, GKLT_K_LEVELN, io, n, L.w, L.w, L.h, g,
                                                    p, g, d, nullptr, 0, 0);
  }
#define GKLT_LEVEL_CASE(RR)                                                                        \
  case RR:                                                                                         \
    return pyr ? launch_level_inst<uint8_t, RR, true>(ctx, s, GKLT_K_LEVEL0, io, n, stride, L.w,   \
                                                      L.h, s0, p, g, d, img0_out, w2, h2)          \
               : launch_level_inst<uint8_t, RR, false>(ctx, s, GKLT_K_LEVEL0, io, n, stride, L.w,  \
                                                       L.h, s0, p, g, d, img0_out, 0, 0);
  switch (s0.r) {
    GKLT_LEVEL_CASE(1)
    GKLT_LEVEL_CASE(2)
    GKLT_LEVEL_CASE(3)
    GKLT_LEVEL_CASE(4)
    GKLT_LEVEL_CASE(5)
    GKLT_LEVEL_CASE(6)
  }
#undef GKLT_LEVEL_CASE
  set_error("fused level kernel: unsupported level-0 radius %d", s0.r);
  return GKLT_EINVAL;
}

#ifndef GKLT_PYR_FMA
// Levels 1 .. L-1 of n same-sized pyramids in one launch (k_level_chain); the caller has checked
// that every level is an exact 2x decimation of the one above and that the radii are the fused kernels'.
int launch_level_chain(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, int n, const Taps &pyr,
                       const Taps &g, const Taps &d) {
  if (n < 1 || n > GKLT_MAX_LOOKAHEAD) return GKLT_EINVAL;
  ChainArgs a{};
  const gklt_pyramid *p0 = py[0];
  a.nlev = p0->nlevels - 1;
  a.nframes = n;
  if (a.nlev < 1) return GKLT_EINVAL;
  int blocks = 0, nflags = 0;
  for (int i = 0; i < a.nlev; ++i) {
    const Level &L = p0->lv[i + 1];
    a.tiles_x[i] = (L.w + TW - 1) / TW;
    a.tiles_y[i] = (L.h + LTH - 1) / LTH;
    a.tiles[i] = a.tiles_x[i] * a.tiles_y[i];
    a.first_block[i] = blocks;
    a.first_flag[i] = nflags;
    a.w[i] = L.w;
    a.h[i] = L.h;
    blocks += n * a.tiles[i];
    nflags += a.tiles[i];
  }
  for (int f = 0; f < n; ++f) {
    gklt_pyramid *q = py[f];
    if (!q->chain_flags || nflags > q->chain_tiles || q->nlevels != p0->nlevels) return GKLT_EINVAL;
    for (int i = 0; i < a.nlev; ++i) {
      if (q->lv[i + 1].w != a.w[i] || q->lv[i + 1].h != a.h[i]) return GKLT_EINVAL;
      a.img[f][i] = q->lv[i + 1].img;
      a.tex[f][i] = q->lv[i + 1].tex;
    }
    a.flags[f] = q->chain_flags;
    a.epoch[f] = ++q->chain_epoch;
  }
  a.err = ctx->h_chain_err;
  constexpr size_t lds = level_lds<0, 5, 3, true>();
  ProfScope ps(ctx, s, GKLT_K_LEVELN);
  if (n > 1)  // batched: 4-wave workgroups, as the per-level launches (see LTH)
    hipLaunchKernelGGL((k_level_chain<5, 3, 256>), dim3(blocks), dim3(256), lds, s, a, pyr, g, d);
  else
    hipLaunchKernelGGL((k_level_chain<5, 3, 512>), dim3(blocks), dim3(512), lds, s, a, pyr, g, d);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_smooth_u8(gklt_ctx *ctx, hipStream_t s, const uint8_t *src, int stride, int w, int h,
                     const Taps &g, float *dst) {
  return launch_smooth_any<uint8_t, 0>(ctx, s, GKLT_K_SMOOTH0, src, stride, w, h, g, dst, 0, 0);
}

int launch_smooth_f32(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, const Taps &g,
                      float *dst, int prof_id) {
  return launch_smooth_any<float, 0>(ctx, s, prof_id, src, w, w, h, g, dst, 0, 0);
}

int launch_smooth_decimate2(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, float *dst, int w2, int h2) {
  if (w != 2 * w2 || h != 2 * h2) {
    set_error("smooth_decimate2 needs an exact 2x size ratio");
    return GKLT_EINVAL;
  }
  return launch_smooth_any<float, 1>(ctx, s, GKLT_K_DOWN, src, w, w, h, g, dst, w2, h2);
}

int launch_resize_linear(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h, float *dst,
                         int w2, int h2) {
  ProfScope ps(ctx, s, GKLT_K_DOWN);
  hipLaunchKernelGGL(k_resize_linear, dim3((w2 + 255) / 256, h2), dim3(256), 0, s, src, w, h, dst,
                     w2, h2, (double)w / w2, (double)h / h2);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_gradients_tex(gklt_ctx *ctx, hipStream_t s, const Level *levels, int first, int count,
                         const Taps &g, const Taps &d) {
  GradBatch b{};
  for (int i = 0; i < count; ++i) {
    const Level &L = levels[first + i];
    b.job[b.n++] = GradJob{L.img, L.tex, nullptr, nullptr, L.w, L.h, 0, 0};
  }
  return launch_grad_batch(ctx, s, b, g, d);
}

int launch_gradients_planar(gklt_ctx *ctx, hipStream_t s, const float *src, int w, int h,
                            const Taps &g, const Taps &d, float *gx, float *gy) {
  GradBatch b{};
  b.job[b.n++] = GradJob{src, nullptr, gx, gy, w, h, 0, 0};
  return launch_grad_batch(ctx, s, b, g, d);
}

int launch_tex_extract(gklt_ctx *ctx, hipStream_t s, const Texel *tex, int n, int comp,
                       float *dst) {
  ProfScope ps(ctx, s, GKLT_K_MISC);
  hipLaunchKernelGGL(k_tex_extract, dim3((n + 255) / 256), dim3(256), 0, s, tex, n, comp, dst);
  GKLT_HIP(hipGetLastError());
  return 0;
}

int launch_tex_to_u8(gklt_ctx *ctx, hipStream_t s, const Texel *tex, int n, uint8_t *dst) {
  ProfScope ps(ctx, s, GKLT_K_MISC);
  hipLaunchKernelGGL(k_tex_to_u8, dim3((n + 255) / 256), dim3(256), 0, s, tex, n, dst);
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

#endif  // !GKLT_PYR_FMA

}  // namespace gklt
