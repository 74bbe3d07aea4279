// Streaming form of the fused level kernel (P1 + P2 + P3 of one pyramid level, pyramid.cpp:45-84):
// the same arithmetic as k_level in gklt_pyramid.hip -- every output is the same left-to-right fp32
// expression, so the planes are bit-identical -- organised for VALU issue, which is what bounds the
// tile kernel (rocprofv3: 5.9e7 VALU wave-instructions per 16-frame launch at C2):
//
//  * A WAVEFRONT owns a strip of 128 image columns (112 outputs + 8 halo columns on each side) and
//    marches down a segment of rows.  The column passes keep their windows (5 + 11 + 7 + 7 rows) in
//    registers, so a row of the image enters the kernel once and no halo ROW is ever recomputed
//    (the 64 x 32 tiles spend 44 % / 31 % of their level-0 / row-pass work on halo rows); only the
//    warm-up rows at the top of a segment are extra (14 of ~120).
//  * A lane holds two columns, l and l + 64 of the strip, as one float2: every filter operation is
//    the same operation on both halves, i.e. one packed instruction (v_pk_mul_f32 / v_pk_add_f32)
//    with naturally aligned operands.  (Packing ADJACENT outputs, as the compiler tries in the
//    tile kernel, leaves half of the operand pairs misaligned.)
//  * Only the two row passes go through LDS (a lane needs its x neighbours): one row of u8 source
//    as float2 pairs and one row of the smoothed image, ~1.2 KB per wavefront, written and read by
//    the same wavefront in program order -- no barriers, no workgroup-level synchronisation.
//  * The unrolled row loop (11 or 13 rows) makes every window index a compile-time constant:
//    the "rings" below are plain SSA values, nothing is shifted or indexed dynamically.
//
// Algorithmic bytes per pixel are those of k_level (1 + 12 + 1); LDS traffic drops from ~119 to
// ~80 bytes per pixel and VALU instructions from ~113 to ~55 per pixel.
#include <cstdlib>

#include "gklt_internal.h"

namespace gklt {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// Strip geometry.  A wavefront loads exactly 128 source columns per row (lane l: columns l and
// l + 64).  The level-0 smoothing (radius R0) is valid on strip columns [R0, 128 - R0), the row
// passes of radius HI = 5 on [R0 + HI, 128 - R0 - HI): the outputs are taken from there, starting
// at an even strip column (2x2 blocks of the decimation then sit in lane pairs l, l ^ 1) and a
// multiple of 16 wide: a row of 16 texels is 192 bytes, so every strip begins on a 64-byte
// boundary of the texel plane.  (Strips whose ends fall inside a 64-byte segment -- 114 or 116
// columns were tried -- are written at 3.2 TB/s instead of 4.8: tools/store_pattern.cpp.)
template <int R0>
struct StripGeom {
  static constexpr int HIG = 5;
  static constexpr int FIRST = (R0 + HIG + 1) & ~1;                  // strip column of the first output
  static constexpr int OUTW = ((128 - R0 - HIG) - FIRST) & ~15;      // 112 for R0 = 0, 2, 3; 96 for R0 = 6
};

// The row loop is unrolled U times, U >= the longest column window (2 R0 + 1 or 2 HI + 1 rows), so
// that window slot indices are compile-time constants; the source prefetch ring has D slots.
template <int R0, int HI>
struct StreamUnroll {
  static constexpr int W = (2 * R0 + 1 > 2 * HI + 1) ? 2 * R0 + 1 : 2 * HI + 1;
  static constexpr int U = (W + 1) & ~1;                 // 12 (windows up to 11 rows) or 14 (13 rows)
  static constexpr int D = U % 3 == 0 ? U / 3 : U / 2;   // 4 or 7
};

struct StreamIO {
  const void *src[GKLT_MAX_LOOKAHEAD];
  Texel *tex[GKLT_MAX_LOOKAHEAD];
  float *down[GKLT_MAX_LOOKAHEAD];
};

__device__ __forceinline__ int refl1(int i, int n) {  // one reflection: |overshoot| < n
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// k0 v[c] + sum_j k_j (v[c+j] + v[c-j]) resp. 0 + sum_j k_j (v[c+j] - v[c-j]), left to right: the
// expressions of corr_win in gklt_pyramid.hip, on both halves of a float2.  `at(m)` returns the
// window element m rows / columns from the centre.
template <int R, int SYM, typename At>
__device__ __forceinline__ v2f corr2(const Taps &t, At at) {
  v2f acc;
  if (SYM > 0) {
    acc = t.k[0] * at(0);
#pragma unroll
    for (int j = 1; j <= R; ++j) acc = acc + t.k[j] * (at(j) + at(-j));
  } else {
    acc = v2f{0.0f, 0.0f};
#pragma unroll
    for (int j = 1; j <= R; ++j) acc = acc + t.k[j] * (at(j) - at(-j));
  }
  return acc;
}

// LDS rows are written by all lanes and read by their neighbours inside ONE wavefront: the hardware
// executes a wavefront's LDS instructions in order, so no s_barrier is needed -- but the compiler
// must not move a lane's reads above (or its next writes below) this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float quad_swap1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
template <typename SrcT, int R0, int RP, int RG, bool DOWN>
__global__ __launch_bounds__(64) void k_level_stream(StreamIO io, int stride, int w, int h, Taps s0, Taps pyr,
                                                     Taps g, Taps d, int w2, int h2, int seg_rows) {
  constexpr int HI = DOWN ? (RP > RG ? RP : RG) : RG;  // halo of the smoothed image I
  constexpr int U = StreamUnroll<R0, HI>::U;  // ring length = unroll factor
  constexpr int D = StreamUnroll<R0, HI>::D;  // source rows are loaded D rows ahead (D divides U)
  using G = StripGeom<R0>;
  static_assert(HI <= G::HIG && G::FIRST >= R0 + HI && G::FIRST + G::OUTW + HI + R0 <= 128, "strip halo");
  __shared__ v2f S2[64 + 2 * R0 + 1];  // source row:   S2[i] = (S[i - R0], S[i - R0 + 64])
  __shared__ v2f I2[64 + 2 * HI];      // smoothed row: I2[i] = (I[i - HI], I[i - HI + 64])

  const SrcT *__restrict__ src = static_cast<const SrcT *>(io.src[blockIdx.z]);
  Texel *__restrict__ tex = io.tex[blockIdx.z];
  float *__restrict__ down = io.down[blockIdx.z];
  const int lane = threadIdx.x;
  const int x0 = blockIdx.x * G::OUTW;                   // first output column of the strip
  const int ya = blockIdx.y * seg_rows;                  // output rows [ya, yb)
  const int yb = min(ya + seg_rows, h);
  const int xa = x0 - G::FIRST + lane, xb = xa + 64;     // image columns of the lane's two halves
  const int sxa = refl1(xa, w), sxb = refl1(xb, w);
  const bool outA = lane >= G::FIRST && xa < w;                     // strip columns [FIRST, 64)
  const bool outB = 64 + lane < G::FIRST + G::OUTW && xb < w;       // strip columns [64, FIRST + OUTW)
  const bool evn = (lane & 1) == 0;                                 // the lane holds even image columns
  const bool dnA = DOWN && evn && outA && (xa >> 1) < w2;
  const bool dnB = DOWN && evn && outB && (xb >> 1) < w2;

  v2f aw[U], rp[U], rd[U], rg[U], ic[U];
#pragma unroll
  for (int k = 0; k < U; ++k) aw[k] = rp[k] = rd[k] = rg[k] = ic[k] = v2f{0.0f, 0.0f};
  v2f o_prev = v2f{0.0f, 0.0f};

  // rows: iteration t handles source/A row ra = ya - HI - R0 + t, completes I row ri = ra - R0, the
  // gradient output row yg = ri - RG and the pyramid output row yp = ri - RP
  const int rows = (yb - ya) + 2 * (HI + R0);
  const int ra0 = ya - HI - R0;
  // Source rows are loaded D rows ahead of their use.  On gfx9 loads and stores share one counter
  // (vmcnt) that retires in order: a load consumed one row after its issue makes the wavefront
  // wait for the texel stores issued in between to be acknowledged by HBM (thousands of cycles,
  // several times a row's arithmetic).  With D rows of distance the wait is "all but the last ~5 D
  // operations", which the stores of the most recent rows do not hold up.
  auto row_ptr = [&](int ra) { return src + (size_t)refl1(ra, h) * stride; };
  SrcT pa[D], pb[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const SrcT *row = row_ptr(ra0 + k);
    pa[k] = row[sxa];
    pb[k] = row[sxb];
  }
  for (int base = 0; base < rows; base += U) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int t = base + k, ra = ra0 + t, ri = ra - R0;
      const SrcT ca = pa[k % D], cb = pb[k % D];
      {
        const SrcT *row = row_ptr(ra + D);
        pa[k % D] = row[sxa];
        pb[k % D] = row[sxb];
      }
      v2f iv;
      if constexpr (R0 > 0) {
        // level 0: u8 row -> fp32 pairs in LDS -> row pass (sigma0) -> column pass (sigma0)
        wave_sync();  // the previous row's reads are done
        // (S2 entries outside [R0, 64 + R0) hold the other half's columns across the middle of the
        // strip, as in I2 below; beyond the strip's ends they are never written: the smoothed columns
        // that would read them, [0, R0) and [128 - R0, 128), are not used by any output)
        S2[lane + R0] = v2f{static_cast<float>(ca), static_cast<float>(cb)};
        if (lane < R0) S2[64 + R0 + lane].x = static_cast<float>(cb);
        if (lane >= 64 - R0) S2[lane - 64 + R0].y = static_cast<float>(ca);
        wave_sync();
        aw[k] = corr2<R0, +1>(s0, [&](int j) { return S2[lane + R0 + j]; });
        iv = corr2<R0, +1>(s0, [&](int j) { return aw[(k - R0 + j + 2 * U) % U]; });
      } else {
        iv = v2f{static_cast<float>(ca), static_cast<float>(cb)};
      }
      ic[k] = iv;
      // I row -> LDS; strip columns 64..64+HI-1 (lane's second half) and -HI..-1 are the halves'
      // neighbours across the middle of the strip
      wave_sync();
      I2[lane + HI] = iv;
      if (lane < HI) I2[64 + HI + lane].x = iv.y;
      if (lane >= 64 - HI) I2[lane - 64 + HI].y = iv.x;
      wave_sync();
      auto iat = [&](int j) { return I2[lane + HI + j]; };
      if constexpr (DOWN) rp[k] = corr2<RP, +1>(pyr, iat);
      rd[k] = corr2<RG, -1>(d, iat);
      rg[k] = corr2<RG, +1>(g, iat);

      {  // gradient column pass -> texels of row yg (pyramid.cpp:72-83)
        const int yg = ri - RG;
        const v2f gx = corr2<RG, +1>(g, [&](int j) { return rd[(k - RG + j + 2 * U) % U]; });
        const v2f gy = corr2<RG, -1>(d, [&](int j) { return rg[(k - RG + j + 2 * U) % U]; });
        const v2f c = ic[(k - RG + 2 * U) % U];
        if (yg >= ya && yg < yb) {
          // streaming stores (see k_level): the texels are next read one build later
          if (outA) {
            Texel *const q = tex + (size_t)yg * w + xa;
            __builtin_nontemporal_store(c.x, &q->x);
            __builtin_nontemporal_store(gx.x, &q->y);
            __builtin_nontemporal_store(gy.x, &q->z);
          }
          if (outB) {
            Texel *const q = tex + (size_t)yg * w + xb;
            __builtin_nontemporal_store(c.y, &q->x);
            __builtin_nontemporal_store(gx.y, &q->y);
            __builtin_nontemporal_store(gy.y, &q->z);
          }
        }
      }
      if constexpr (DOWN) {  // pyramid column pass -> 2x2 mean -> next level (pyramid.cpp:62-66)
        const int yp = ri - RP;
        const v2f o = corr2<RP, +1>(pyr, [&](int j) { return rp[(k - RP + j + 2 * U) % U]; });
        if (yp & 1) {  // ya is even: rows (yp - 1, yp) form a block; its right column sits in lane + 1
          const v2f a = o_prev, c = o;
          const v2f b = v2f{quad_swap1(a.x), quad_swap1(a.y)}, e = v2f{quad_swap1(c.x), quad_swap1(c.y)};
          const v2f m = GKLT_MEAN2X2(a, b, c, e);
          const int y2 = yp >> 1;
          if (yp >= ya && yp < yb && y2 < h2) {
            if (dnA) down[(size_t)y2 * w2 + (xa >> 1)] = m.x;
            if (dnB) down[(size_t)y2 * w2 + (xb >> 1)] = m.y;
          }
        }
        o_prev = o;
      }
    }
  }
}

template <typename SrcT, int R0, bool DOWN>
int launch_inst(gklt_ctx *ctx, hipStream_t s, int prof_id, const StreamIO &io, int nframes, int stride, int w, int h,
                const Taps &s0, const Taps &pyr, const Taps &g, const Taps &d, int w2, int h2) {
  const int nstrips = (w + StripGeom<R0>::OUTW - 1) / StripGeom<R0>::OUTW;
  // ~2600 wavefronts per launch (2.5 per SIMD: all resident, enough to overlap VALU, LDS and HBM).
  // A segment of r output rows takes r + warm row iterations (warm = 2 (HI + R0) rows fill the
  // windows), rounded up to the unroll factor U: r is chosen even and with r + warm a multiple of U.
  constexpr int HI = DOWN ? 5 : 3, U = StreamUnroll<R0, HI>::U, warm = 2 * (HI + R0);
  static int target_waves = [] {  // measurement knob
    const char *e = std::getenv("GKLT_STREAM_WAVES");
    return e ? std::atoi(e) : 2600;
  }();
  const int want = std::max(1, (target_waves + nstrips * nframes / 2) / (nstrips * nframes));
  const int m = std::max(((h + want - 1) / want + warm + U / 2) / U, (32 + warm + U - 1) / U);
  const int seg_rows = U * m - warm;  // even: U and warm are
  const int nsegs = (h + seg_rows - 1) / seg_rows;
  ProfScope ps(ctx, s, prof_id);
  hipLaunchKernelGGL((k_level_stream<SrcT, R0, 5, 3, DOWN>), dim3(nstrips, nsegs, nframes), dim3(64), 0, s, io, stride,
                     w, h, s0, pyr, g, d, w2, h2, seg_rows);
  GKLT_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool level_streamable(int w, int h, int level, const Taps &s0) {
  if (w < 192 || h < 192) return false;  // single reflection at the borders
  return level > 0 || s0.r == 2 || s0.r == 3 || s0.r == 6;  // level 0: 7x7, 11x11, 21x21 windows
}

// Level `level` of n pyramids in one launch of the streaming kernel; same contract as launch_level
// (gklt_pyramid.hip) for the cases level_streamable() accepts and radii (pyramid 5, gradient 3).
int launch_level_stream(gklt_ctx *ctx, hipStream_t s, gklt_pyramid *const *py, const uint8_t *const *src, int n,
                        int stride, int level, const Taps &s0, const Taps *pyr, const Taps &g, const Taps &d) {
  if (n < 1 || n > GKLT_MAX_LOOKAHEAD) return GKLT_EINVAL;
  StreamIO io{};
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
  if (level > 0)
    return pyr ? launch_inst<float, 0, true>(ctx, s, GKLT_K_LEVELN, io, n, L.w, L.w, L.h, g, p, g, d, w2, h2)
               : launch_inst<float, 0, false>(ctx, s, GKLT_K_LEVELN, io, n, L.w, L.w, L.h, g, p, g, d, 0, 0);
#define GKLT_STREAM_CASE(RR)                                                                                      \
  case RR:                                                                                                        \
    return pyr ? launch_inst<uint8_t, RR, true>(ctx, s, GKLT_K_LEVEL0, io, n, stride, L.w, L.h, s0, p, g, d, w2, h2) \
               : launch_inst<uint8_t, RR, false>(ctx, s, GKLT_K_LEVEL0, io, n, stride, L.w, L.h, s0, p, g, d, 0, 0);
  switch (s0.r) {
    GKLT_STREAM_CASE(2)
    GKLT_STREAM_CASE(3)
    GKLT_STREAM_CASE(6)
  }
#undef GKLT_STREAM_CASE
  set_error("streaming level kernel: unsupported level-0 radius %d", s0.r);
  return GKLT_EINVAL;
}

}  // namespace gklt
