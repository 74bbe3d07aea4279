// Coarse-to-fine iterative Lucas-Kanade for gfx950 -- replaces KLTTrackFeatures / KLTInvoker /
// _trackFeature and the AVX helpers under them (LK.hpp:56-208, 413-669, 825-1046).
//
// Two kernels, bit-identical to each other and to the oracle:
//  * k_track_w (the default, second half of this file): ONE FEATURE PER WAVEFRONT, the reference's
//    AVX summation order rebuilt through LDS chain sums.  Sampling by window size:
//      footprint (w+1)(h+1) <= 64 texels (7x7): footprint in registers, one load per lane and
//        patch, neighbours through DPP / one shuffle (FRM);
//      up to ~14 px: element i -> lane i % 64, four 12-byte gathers per element;
//      >= 15 px (w*h >= 200): footprint staged in LDS once per pass; for windows 15..21 px wide
//        a lane owns a run of 7 consecutive elements of one window row (RUN), otherwise
//        element i -> lane i % 64.
//    Multi-frame (lookahead) and multi-stream launches.
//  * k_track (track_impl 1, and the fallback for windows the wave kernel cannot hold): one
//    feature per 8-lane slice (8 features per wave64).  Lane k of a slice owns window elements
//    k, k+8, k+16, ... in row-major window order -- exactly the elements AVX lane k accumulates
//    in the reference (LK.hpp:127-166) -- and the 8 partials are added left to right
//    (LK.hpp:143-145,163-164,207) after a cross-lane exchange inside the slice.
// With the translation unit built -ffp-contract=off every fp32 operation rounds as the
// reference's unfused AVX code does, which makes positions and status flags reproducible bit
// for bit.  All pyramid levels and all Newton iterations of a feature run inside one launch.
// Each bilinear tap is one 12-byte load of an (I,gx,gy) texel (layout written by
// gklt_pyramid.hip).
#include <algorithm>
#include <cstdlib>

#include "gklt_internal.h"

namespace gklt {

namespace {

struct TrackLevel {
  const Texel *t1;
  const Texel *t2;
  int nc, nr;
};

struct TrackArgs {
  TrackLevel lv[GKLT_MAX_LEVELS];
  int nlevels;
  float ss;
  float inv_ss;  // 1 / ss when ss is a power of two (exact), else 0
  int ww, wh;
  float step_factor, min_det, min_disp, max_residue;
  int max_iter;
  int w0, h0, borderx, bordery;
  int n;
  int do_epilogue;
  // optional host-visible copy of the list: x[stride] | y[stride] | val[stride], written by the
  // kernel itself (pinned host memory), so no D2H copy sits on the stream after the launch
  float *mirror;
  int mirror_stride;
  int halo;     // k_track_w with staging: margin of the LDS footprint in texels
  int fstride;  // ... and its row stride in LDS, texels (>= footprint width; bank spreading)
  // ceil(2^16 / d) for d = window width and footprint width: j / d == (j * magic) >> 16 for the
  // j < 1024 the lane mappings divide (an integer division is ~30 VALU instructions on this part)
  int ww_magic, fw_magic;
};

// Multi-frame launches (lookahead pipeline): a wavefront follows its feature through nframes
// consecutive frame pairs (tex[j], tex[j+1]); features are independent of each other, so this is
// the same computation as nframes launches, without the launch boundaries in between.
struct TrackFrames {
  const Texel *tex[GKLT_MAX_LOOKAHEAD + 1][GKLT_MAX_LEVELS];
  float *mirror[GKLT_MAX_LOOKAHEAD];  // per frame: host-visible x | y | val (or nullptr)
  int nframes;
};

// Multi-stream launches (gklt_multistream): one launch tracks the lists of several independent
// video streams, each between its own pair of pyramids.  The texel planes of every stream's ring
// slots sit in a device table that never changes; a launch names the two slots.  Stream s owns
// list entries [s * cap, s * cap + n[s]).
struct TrackStreams {
  const Texel *const *tab;  // device: [nstreams][3][GKLT_MAX_LEVELS]
  int nstreams, slot1, slot2, cap, nmax;
  int n[GKLT_MAX_STREAMS];
};

constexpr int SL = 8;  // lanes per feature == simd_step of the reference (LK.hpp:40)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// words between the product arrays of k_track_w (see there): NA + 8 or NA + 16, an odd multiple of 8
// (always > NA: the gap takes the writes of lanes that own no element, see RUN)
__host__ __device__ constexpr int prod_stride(int NA) { return (NA >> 3) & 1 ? NA + 16 : NA + 8; }
constexpr int kStatShards = GKLT_STAT_SHARDS;

struct Bilin {
  int xt, yt;
  float w0, w1, w2, w3;
};

// computeBilinearWeight, LK.hpp:56-68: truncating coordinates, three weights formed in double.
__device__ __forceinline__ Bilin bilinear_weight(float x, float y) {
  Bilin b;
  b.xt = (int)x;
  b.yt = (int)y;
  const float ax = x - b.xt, ay = y - b.yt;
  if (x >= 1.0f && y >= 1.0f) {
    // Same bits without fp64: for x >= 1 the fraction ax is a multiple of ulp(x) >= 2^-23, so
    // 1 - ax is exactly representable in fp32, and the double product of two fp32 numbers is
    // exact, i.e. rounding it to fp32 is the correctly rounded fp32 product.  Every sampled
    // position satisfies x - hw >= 0 with hw >= 1 (window_oob), so this is the path taken.
    const float bx = 1.0f - ax, by = 1.0f - ay;
    b.w0 = bx * by;
    b.w1 = ax * by;
    b.w2 = bx * ay;
  } else {
    b.w0 = (float)((1.0 - ax) * (1.0 - ay));
    b.w1 = (float)((ax) * (1.0 - ay));
    b.w2 = (float)((1.0 - ax) * ay);
  }
  b.w3 = ax * ay;
  // Integer position: the reference copies rows instead of interpolating (LK.hpp:80-85).  With
  // the other three weights forced to zero the interpolation formula returns the same sample
  // ((p00*1 + p01*0) + (p10*0 + p11*0) == p00 for finite texels) and stays branch-free, so all
  // gathers of a pass can be in flight together.
  if (b.w0 == 1.0f) { b.w1 = 0.f; b.w2 = 0.f; b.w3 = 0.f; }
  return b;
}

// One window element of computeBilinearPatch (LK.hpp:71-108) for all three planes at once.
__device__ __forceinline__ float3 sample(const Texel *__restrict__ base, int nc, const Bilin &b) {
  const Texel t00 = base[0], t01 = base[1], t10 = base[nc], t11 = base[nc + 1];
  float3 o;
  o.x = (t00.x * b.w0 + t01.x * b.w1) + (t10.x * b.w2 + t11.x * b.w3);
  o.y = (t00.y * b.w0 + t01.y * b.w1) + (t10.y * b.w2 + t11.y * b.w3);
  o.z = (t00.z * b.w0 + t01.z * b.w1) + (t10.z * b.w2 + t11.z * b.w3);
  return o;
}

__device__ __forceinline__ float sample_i(const Texel *__restrict__ base, int nc, const Bilin &b) {
  const float p00 = base[0].x, p01 = base[1].x, p10 = base[nc].x, p11 = base[nc + 1].x;
  return (p00 * b.w0 + p01 * b.w1) + (p10 * b.w2 + p11 * b.w3);
}

// The value of lane l+1 (lane 63: unspecified) through a DPP wave shift: a VALU move, no trip
// through the LDS crossbar.
__device__ __forceinline__ float from_next_lane(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// lanes 0..7 of the slice added left to right; every lane of the slice gets the total.
__device__ __forceinline__ float slice_sum(float v) {
  float t = __shfl(v, 0, SL);
#pragma unroll
  for (int j = 1; j < SL; ++j) t = t + __shfl(v, j, SL);
  return t;
}

// The bounds test of LK.hpp:558-561 / :620-621 for one point, written so that a NaN coordinate
// counts as out of bounds (the reference would index memory with it).
__device__ __forceinline__ bool window_oob(float x, float y, int hw, int hh, int nc, int nr) {
  const float eps1 = 1.001f;
  return !(x - hw >= 0.0f) || !(nc - (x + hw) >= eps1) || !(y - hh >= 0.0f) ||
         !(nr - (y + hh) >= eps1);
}

// Keeps the 2x2 footprint of every tap inside the plane even if the float bounds test were to
// admit a window touching the last column/row (cannot change a result the test allows).
__device__ __forceinline__ void clamp_origin(Bilin &b, int hw, int hh, int nc, int nr) {
  b.xt = min(max(b.xt, hw), nc - 2 - hw);
  b.yt = min(max(b.yt, hh), nr - 2 - hh);
}

// Window element i of the slice lane, clamped so that the (zero-valued) tail elements of the
// padded patch (alignedPatchSize, LK.hpp:448) still form a valid address: loads stay
// unconditional, which lets the compiler keep a whole pass of gathers in flight.
__device__ __forceinline__ int elem_offset(int i, int npix, int ww, int nc) {
  const int ic = min(i, npix - 1);
  return (ic / ww) * nc + (ic % ww);
}

// WW,WH > 0: compile-time window, template patch in registers.
// WW == 0 : run-time window, template patch in LDS ([plane][element][lane]).
template <int WW, int WH>
__global__ __launch_bounds__(64) void k_track(TrackArgs a, float *__restrict__ fx,
                                              float *__restrict__ fy, int *__restrict__ fval,
                                              const double *__restrict__ invk,
                                              double *__restrict__ norm3,
                                              double *__restrict__ vec3,
                                              unsigned long long *__restrict__ stats) {
  constexpr bool kStatic = WW > 0;
  const int ww = kStatic ? WW : a.ww, wh = kStatic ? WH : a.wh;
  const int npix = ww * wh;
  const int ne = (npix + SL) / SL;  // alignedPatchSize / 8 (LK.hpp:448)
  constexpr int NE = kStatic ? (WW * WH + SL) / SL : 1;
  const int hw = ww / 2, hh = wh / 2;

  extern __shared__ __align__(16) float lds[];  // generic kernel only: 3 * ne * 64 floats
  float P1[NE], GX1[NE], GY1[NE];
  float3 S2[NE];

  const int lane = threadIdx.x;
  const int k = lane & (SL - 1);
  const int f = blockIdx.x * (64 / SL) + (lane >> 3);
  const bool alive = f < a.n && fval[f] >= 0;  // LK.hpp:830

  float xloc = 0.f, yloc = 0.f;
  if (alive) { xloc = fx[f]; yloc = fy[f]; }
  for (int r = 0; r < a.nlevels; ++r) { xloc /= a.ss; yloc /= a.ss; }  // LK.hpp:837-839
  float xo = xloc, yo = yloc;
  int v = GKLT_TRACKED;
  int iters_total = 0;
  bool descending = alive;

  for (int r = a.nlevels - 1; r >= 0; --r) {  // LK.hpp:859-881
    // After SMALL_DET / OOB the reference leaves the loop (LK.hpp:879-880) and tests
    // _outOfBounds on xlocout at the scale of the level it broke at: stop scaling too.
    if (!descending) continue;
    xloc *= a.ss; yloc *= a.ss;
    xo *= a.ss; yo *= a.ss;

    // ---- _trackFeature, LK.hpp:413-669 ----
    const TrackLevel L = a.lv[r];
    const int nc = L.nc, nr = L.nr;
    int status = GKLT_TRACKED, iteration = 0;
    const float x1 = xloc, y1 = yloc;
    float x2 = xo, y2 = yo;

    if (window_oob(x1, y1, hw, hh, nc, nr) || window_oob(x2, y2, hw, hh, nc, nr)) {
      status = GKLT_OOB;  // first pass of the do-loop, :558-564
    } else {
      // template patches of frame 1 (:467-476; sampled after the bounds test)
      Bilin b1 = bilinear_weight(x1, y1);
      clamp_origin(b1, hw, hh, nc, nr);
      const Texel *o1 = L.t1 + (size_t)(b1.yt - hh) * nc + (b1.xt - hw);
      Bilin b2 = bilinear_weight(x2, y2);  // :568-572
      clamp_origin(b2, hw, hh, nc, nr);
      const Texel *o2 = L.t2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);

      if constexpr (kStatic) {
        // template gathers and the first iteration's gathers are issued together
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int i = e * SL + k;
          float3 s = sample(o1 + elem_offset(i, WW * WH, WW, nc), nc, b1);
          if (e * SL + SL - 1 >= WW * WH && i >= WW * WH) s = make_float3(0.f, 0.f, 0.f);
          P1[e] = s.x; GX1[e] = s.y; GY1[e] = s.z;
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int i = e * SL + k;
          S2[e] = sample(o2 + elem_offset(i, WW * WH, WW, nc), nc, b2);
          if (e * SL + SL - 1 >= WW * WH && i >= WW * WH) S2[e] = make_float3(0.f, 0.f, 0.f);
        }
      } else {
#pragma unroll 4
        for (int e = 0; e < ne; ++e) {
          const int i = e * SL + k;
          float3 s = sample(o1 + elem_offset(i, npix, ww, nc), nc, b1);
          if (i >= npix) s = make_float3(0.f, 0.f, 0.f);
          lds[(0 * ne + e) * 64 + lane] = s.x;
          lds[(1 * ne + e) * 64 + lane] = s.y;
          lds[(2 * ne + e) * 64 + lane] = s.z;
        }
      }

      float dx, dy;
      for (;;) {
        float axx = 0.f, ayy = 0.f, axy = 0.f, aex = 0.f, aey = 0.f;
        auto accumulate = [&](float p1, float g1x, float g1y, const float3 &s) {
          const float diff = p1 - s.x;   // computeImageDiff :110-116
          const float gx = g1x + s.y;    // computeImageSum  :118-125
          const float gy = g1y + s.z;
          axx = axx + gx * gx;           // compute2X2GradientMatrix :127-146
          ayy = ayy + gy * gy;
          axy = axy + gx * gy;
          aex = aex + gx * diff;         // compute2X1ErrorVector :148-166
          aey = aey + gy * diff;
        };
        if constexpr (kStatic) {
#pragma unroll
          for (int e = 0; e < NE; ++e) accumulate(P1[e], GX1[e], GY1[e], S2[e]);
        } else {
#pragma unroll 8
          for (int e = 0; e < ne; ++e) {
            const int i = e * SL + k;
            float3 s = sample(o2 + elem_offset(i, npix, ww, nc), nc, b2);
            if (i >= npix) s = make_float3(0.f, 0.f, 0.f);
            accumulate(lds[(0 * ne + e) * 64 + lane], lds[(1 * ne + e) * 64 + lane],
                       lds[(2 * ne + e) * 64 + lane], s);
          }
        }
        const float gxx = slice_sum(axx), gyy = slice_sum(ayy), gxy = slice_sum(axy);
        const float ex = slice_sum(aex) * a.step_factor, ey = slice_sum(aey) * a.step_factor;

        const float det = gxx * gyy - gxy * gxy;  // _solveEquation :168-183
        if (det < a.min_det) { status = GKLT_SMALL_DET; break; }
        dx = (gyy * ex - gxy * ey) / det;
        dy = (gxx * ey - gxy * ex) / det;
        status = GKLT_TRACKED;
        x2 += dx; y2 += dy;
        ++iteration;
        if (!((fabsf(dx) >= a.min_disp || fabsf(dy) >= a.min_disp) && iteration < a.max_iter))
          break;  // :615
        if (window_oob(x2, y2, hw, hh, nc, nr)) { status = GKLT_OOB; break; }  // next pass :558
        b2 = bilinear_weight(x2, y2);  // :568-572
        clamp_origin(b2, hw, hh, nc, nr);
        o2 = L.t2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);
        if constexpr (kStatic) {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int i = e * SL + k;
            S2[e] = sample(o2 + elem_offset(i, WW * WH, WW, nc), nc, b2);
            if (e * SL + SL - 1 >= WW * WH && i >= WW * WH) S2[e] = make_float3(0.f, 0.f, 0.f);
          }
        }
      }

      if (window_oob(x2, y2, hw, hh, nc, nr)) status = GKLT_OOB;  // :620-623

      // Residue, :626-636.  At levels above 0 its only effect would be KLT_LARGE_RESIDUE, which
      // neither stops the descent (:879) nor survives it (the next level overwrites val), so it
      // is evaluated at level 0 only.
      if (status == GKLT_TRACKED && r == 0) {
        b2 = bilinear_weight(x2, y2);
        clamp_origin(b2, hw, hh, nc, nr);
        o2 = L.t2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);
        float asum = 0.f;
        if constexpr (kStatic) {
          float sv[NE];
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int i = e * SL + k;
            sv[e] = sample_i(o2 + elem_offset(i, WW * WH, WW, nc), nc, b2);
            if (e * SL + SL - 1 >= WW * WH && i >= WW * WH) sv[e] = 0.f;
          }
#pragma unroll
          for (int e = 0; e < NE; ++e) asum = asum + fabsf(P1[e] - sv[e]);  // :193-208
        } else {
#pragma unroll 8
          for (int e = 0; e < ne; ++e) {
            const int i = e * SL + k;
            float sI = sample_i(o2 + elem_offset(i, npix, ww, nc), nc, b2);
            if (i >= npix) sI = 0.f;
            asum = asum + fabsf(lds[(0 * ne + e) * 64 + lane] - sI);
          }
        }
        if (slice_sum(asum) / (ww * wh) > a.max_residue) status = GKLT_LARGE_RESIDUE;
      }
    }
    iters_total += iteration;
    xo = x2; yo = y2;
    // return value precedence, :655-667
    if (status == GKLT_SMALL_DET || status == GKLT_OOB || status == GKLT_LARGE_RESIDUE) v = status;
    else v = iteration >= a.max_iter ? GKLT_MAX_ITERATIONS : GKLT_TRACKED;
    if (v == GKLT_SMALL_DET || v == GKLT_OOB) descending = false;  // :879-880
  }

  // ---- record feature, LK.hpp:886-915 ----
  bool tracked = false;
  if (alive) {
    const bool oob = xo < a.borderx || xo > a.w0 - 1 - a.borderx || yo < a.bordery ||
                     yo > a.h0 - 1 - a.bordery;  // _outOfBounds :185-191
    int out;
    if (v == GKLT_OOB) out = GKLT_OOB;
    else if (oob) out = GKLT_OOB;
    else out = v;
    tracked = out == GKLT_TRACKED;
    if (k == 0) {
      fx[f] = tracked ? xo : -1.0f;
      fy[f] = tracked ? yo : -1.0f;
      fval[f] = out;
      if (a.mirror) {
        a.mirror[f] = tracked ? xo : -1.0f;
        a.mirror[a.mirror_stride + f] = tracked ? yo : -1.0f;
        reinterpret_cast<int *>(a.mirror)[2 * a.mirror_stride + f] = out;
      }
      if (tracked && a.do_epilogue) {  // bearing vectors, LK.hpp:1019-1035 (fp64)
        const double px = (double)xo, py = (double)yo;
        double q[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) q[i] = (invk[3 * i] * px + invk[3 * i + 1] * py) + invk[3 * i + 2];
        const double nn = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
        double nv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { nv[i] = q[i] / nn; norm3[3 * (size_t)f + i] = nv[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) vec3[3 * (size_t)f + i] = nv[i] / nv[2];
      }
    }
  }
  if (!alive && f < a.n && k == 0 && a.mirror) {  // lost features: mirror the untouched record
    a.mirror[f] = fx[f];
    a.mirror[a.mirror_stride + f] = fy[f];
    reinterpret_cast<int *>(a.mirror)[2 * a.mirror_stride + f] = fval[f];
  }
  if (stats) {  // one pair of atomics per wave
    const unsigned long long m = __ballot(tracked && k == 0);
    int it = (alive && k == 0) ? iters_total : 0;
#pragma unroll
    for (int o = 32; o >= SL; o >>= 1) it += __shfl_xor(it, o);
    if (lane == 0) {
      unsigned long long *sh = stats + 2 * (blockIdx.x & (kStatShards - 1));
      atomicAdd(&sh[0], (unsigned long long)__popcll(m));
      atomicAdd(&sh[1], (unsigned long long)it);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_track_w: ONE FEATURE PER WAVEFRONT.  The window's elements are spread over all 64 lanes
// (element i -> lane i % 64), so sampling is 8x more parallel than in the slice kernel above and
// every branch (levels, Newton iterations, status) is wave-uniform.  The reference's summation
// order is kept by a two-step reduction through LDS: the per-element products are written to
// LDS, lane (q, k) of the first 40 lanes then adds chain k of quantity q (elements k, k+8, ...)
// in element order -- the AVX lane partial of LK.hpp:133-137,154-157 -- and lanes 0..4 add the 8
// partials of their quantity left to right (LK.hpp:143-145,163-164).  The five totals leave as
// wave-uniform scalars (v_readlane).  EPL = window elements per lane (compile time).
// STAGE: the (w+1) x (h+1) texel footprint of a pass is fetched once (one 12-byte gather per
// footprint texel) into LDS and the four bilinear taps of every element are read from there,
// instead of four gathers per element.
// FPL = footprint texels per lane (0 = no staging).
// NEC = elements per AVX-lane chain when known at compile time (7x7: 7, 11x11: 16, 21x21: 56;
// 0 = run time): the chain sums then unroll into NEC reads followed by NEC ordered adds.
// MS: multi-stream launch (TrackStreams M names the pyramids and list segments, F.nframes == 1).
// FRM: footprint-in-registers sampling for windows whose bilinear footprint (w+1)(h+1) fits one
// wavefront (7x7: exactly 64 texels).  Lane l holds footprint texel (l / (w+1), l % (w+1)), loaded
// with ONE global_load_dwordx3 per patch instead of four gathers per element; the right and lower
// neighbours come from lanes l+1 and l+(w+1).  The interpolation is evaluated as
//   (t00*w0 + t01*w1) + B(l + (w+1)),   B(l) = t(l)*w2 + t(l+1)*w3
// which is the reference's expression, operation for operation (B(l+(w+1)) is t10*w2 + t11*w3).
// The texture path then sees a quarter of the lane-addresses and about a sixth of the cache
// accesses.  Window element (r, c) sits in lane r*(w+1)+c and keeps its row-major index r*w+c
// for the ordered reductions.
// RUN > 0 (staged windows 15..21 px wide): lane l owns the RUN consecutive elements
// (r, c0 .. c0 + RUN - 1) of ONE window row, r = l / S, c0 = (l % S) * RUN, S = ceil(w / RUN),
// instead of elements l, l + 64, ...  Its 2 (RUN + 1) footprint texels are then read from LDS once
// per pass at constant offsets from one address (4 RUN reads and RUN addresses otherwise), and
// its products go to consecutive words.  Which lane computes an element does not enter a result.
template <int EPL, int FPL, int NEC = 0, bool MS = false, bool FRM = false, int RUN = 0>
__global__ __launch_bounds__(64) void k_track_w(TrackArgs a, TrackFrames F, TrackStreams M,
                                                float *__restrict__ fx,
                                                float *__restrict__ fy, int *__restrict__ fval,
                                                const double *__restrict__ invk,
                                                double *__restrict__ norm3,
                                                double *__restrict__ vec3,
                                                unsigned long long *__restrict__ stats) {
  extern __shared__ __align__(16) float lds[];  // prod[5][NAs] | part[40] | fp[NF + 8] (float4)
  constexpr int NT = 64;
  const int lane = threadIdx.x;
  const int ww = a.ww, wh = a.wh, npix = ww * wh;
  const int NA = SL * ((npix + SL) / SL);  // alignedPatchSize (LK.hpp:448)
  const int nec = NEC > 0 ? NEC : NA / SL;  // elements per AVX-lane chain
  const int hw = ww / 2, hh = wh / 2;
  constexpr bool STAGE = FPL > 0;
  constexpr int FP = STAGE ? FPL : 1;
  // The five chains of a reduction read prod[q][k + 8u] together (lanes 8q + k): the stride between
  // quantities is an odd multiple of 8 words, so that the 40 lanes fall into 40 different banks
  // (NA itself is a multiple of 64 for 11x11 and 21x21 windows: a five-way conflict on every read).
  const int NAs = NEC > 0 ? prod_stride(SL * NEC) : prod_stride(NA);
  float *prod = lds;
  float *part = lds + 5 * NAs;
  float4 *fp = reinterpret_cast<float4 *>(part + 40);
  // Staged footprint: (w + 1 + 2 HL) x (h + 1 + 2 HL) texels around the window's origin.  The HL
  // extra texels on every side let the Newton iterations that move the window by less than HL
  // whole pixels -- nearly all but the first of a level -- sample from the texels already in LDS:
  // an iteration then costs no trip to memory.  Same texels, same weights, same sums.
  const int HL = STAGE ? a.halo : 0;
  const int FW = ww + 1 + 2 * HL, NF = FW * (wh + 1 + 2 * HL);
  const int FS = STAGE ? a.fstride : FW;  // footprint texel (x, y) sits at fp[y * FS + x]

  // element -> texel offset inside the window (tail elements clamp to the last pixel, value 0)
  // and -> index of its products in prod[] (-1: none)
  int eoff_x[EPL], eoff_y[EPL], pidx[EPL];
  bool evalid[EPL];
  if constexpr (FRM) {
    static_assert(EPL == 1 && FPL == 0, "one footprint texel per lane");
    const int jc = min(lane, NF - 1);
    eoff_y[0] = (jc * a.fw_magic) >> 16;
    eoff_x[0] = jc - eoff_y[0] * FW;
    evalid[0] = lane < NF && eoff_y[0] < wh && eoff_x[0] < ww;
    // the zero tail npix..NA-1 of the padded patch is written by the first lanes that hold no
    // window element
    const unsigned long long spare = __ballot(!evalid[0]);
    const int rank = __popcll(spare & ((1ull << lane) - 1ull));
    pidx[0] = evalid[0] ? eoff_y[0] * ww + eoff_x[0] : (rank < NA - npix ? npix + rank : -1);
  } else if constexpr (RUN > 0) {
    static_assert(EPL == RUN && FPL > 0, "row-run mapping samples from the staged footprint");
    static_assert(RUN == 7, "three runs per row: windows 15..21 wide (wave_geom)");
    constexpr int S = 3;
    const int r = lane / S, c0 = (lane - r * S) * RUN;
    const int rc = min(r, wh - 1);  // lanes past the last row read (and discard) its texels
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      eoff_y[e] = rc;
      eoff_x[e] = c0 + e;  // up to RUN - 1 texels past the row's end: fp[] is padded for it
      evalid[e] = r < wh && c0 + e < ww;
      // products of elements nobody owns go to the gap behind each array (NA + e < NAs): the
      // writes of a pass are then unconditional and share one address register per element
      pidx[e] = evalid[e] ? rc * ww + c0 + e : NA + e;
    }
    // nobody owns the zero tail npix..NA-1 of the padded patch: written once
    if (lane < NA - npix) {
#pragma unroll
      for (int q = 0; q < 5; ++q) prod[q * NAs + npix + lane] = 0.f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int i = e * NT + lane;
      const int ic = min(i, npix - 1);
      eoff_y[e] = (ic * a.ww_magic) >> 16;
      eoff_x[e] = ic - eoff_y[e] * ww;
      evalid[e] = i < npix;
      pidx[e] = i < NA ? i : -1;
    }
  }
  // one patch element for all three planes / for the image plane only
  // FRM: the interpolation from the lane's raw footprint texel.  The raw texels of the second image
  // are kept between Newton iterations: an iteration that leaves the window in the same pixel cell
  // (most after the first of a level: the steps are sub-pixel) changes the weights only, and costs
  // no trip to memory -- the dependent load was the longest link of the per-feature chain.
  auto frm3 = [&](const Texel &t, const Bilin &b) {
    const float nx = from_next_lane(t.x), ny = from_next_lane(t.y), nz = from_next_lane(t.z);
    const float bx = t.x * b.w2 + nx * b.w3, by = t.y * b.w2 + ny * b.w3, bz = t.z * b.w2 + nz * b.w3;
    float3 o;
    o.x = (t.x * b.w0 + nx * b.w1) + __shfl_down(bx, FW);
    o.y = (t.y * b.w0 + ny * b.w1) + __shfl_down(by, FW);
    o.z = (t.z * b.w0 + nz * b.w1) + __shfl_down(bz, FW);
    return o;
  };
  auto frm1 = [&](float t, const Bilin &b) {
    const float n = from_next_lane(t);
    const float bl = t * b.w2 + n * b.w3;
    return (t * b.w0 + n * b.w1) + __shfl_down(bl, FW);
  };
  auto gather3 = [&](const Texel *__restrict__ p, int nc, const Bilin &b) {
    if constexpr (FRM) {
      const Texel t = *p;
      const float nx = from_next_lane(t.x), ny = from_next_lane(t.y), nz = from_next_lane(t.z);
      const float bx = t.x * b.w2 + nx * b.w3, by = t.y * b.w2 + ny * b.w3, bz = t.z * b.w2 + nz * b.w3;
      float3 o;
      o.x = (t.x * b.w0 + nx * b.w1) + __shfl_down(bx, FW);
      o.y = (t.y * b.w0 + ny * b.w1) + __shfl_down(by, FW);
      o.z = (t.z * b.w0 + nz * b.w1) + __shfl_down(bz, FW);
      return o;
    } else {
      return sample(p, nc, b);
    }
  };
  auto gather1 = [&](const Texel *__restrict__ p, int nc, const Bilin &b) {
    if constexpr (FRM) {
      const float t = p->x;
      const float n = from_next_lane(t);
      const float bl = t * b.w2 + n * b.w3;
      return (t * b.w0 + n * b.w1) + __shfl_down(bl, FW);
    } else {
      return sample_i(p, nc, b);
    }
  };
  // footprint texel j = e*NT + lane -> offset inside the footprint (clamped past the end)
  int foff_x[FP], foff_y[FP];
  if (STAGE) {
#pragma unroll
    for (int e = 0; e < FP; ++e) {
      const int jc = min(e * NT + lane, NF - 1);
      foff_y[e] = (jc * a.fw_magic) >> 16;
      foff_x[e] = jc - foff_y[e] * FW;
    }
  }
  // footprint whose first texel is (X0, Y0); texels of the margin that fall outside the plane are
  // clamped into it (never sampled: a window that would reach them is out of bounds first)
  // X0, Y0 are the same in every lane; telling the compiler so (readfirstlane) moves the address
  // arithmetic of the common case -- footprint inside the plane -- to the scalar unit: one base
  // address plus a per-lane offset per texel.
  auto fp_load = [&](const Texel *plane, int nc, int nr, int X0v, int Y0v, float3 (&t)[FP]) {
    const int X0 = __builtin_amdgcn_readfirstlane(X0v), Y0 = __builtin_amdgcn_readfirstlane(Y0v);
    if (X0 >= 0 && Y0 >= 0 && X0 + FW <= nc && Y0 + (NF / FW) <= nr) {
      const Texel *o = plane + ((size_t)Y0 * nc + X0);
#pragma unroll
      for (int e = 0; e < FP; ++e) {
        const Texel v = o[(unsigned)(foff_y[e] * nc + foff_x[e])];
        t[e] = make_float3(v.x, v.y, v.z);
      }
    } else {
#pragma unroll
      for (int e = 0; e < FP; ++e) {
        const int tx = min(max(X0 + foff_x[e], 0), nc - 1), ty = min(max(Y0 + foff_y[e], 0), nr - 1);
        const Texel v = plane[(size_t)ty * nc + tx];
        t[e] = make_float3(v.x, v.y, v.z);
      }
    }
  };
  // LDS texel: (gx, gy, I, -) -- the gradient pair sits in an even-aligned register pair after the
  // 16-byte read, which is what the packed fp32 instructions take
  auto fp_store = [&](const float3 (&t)[FP]) {
#pragma unroll
    for (int e = 0; e < FP; ++e)
      if (e * NT + lane < NF) fp[foff_y[e] * FS + foff_x[e]] = make_float4(t[e].y, t[e].z, t[e].x, 0.f);
  };
  // element e of the window whose origin sits (ox, oy) texels inside the staged footprint
  // A texel is read with ONE ds_read_b128 (volatile LDS access): the fourth component is unused, and
  // the compiler otherwise narrows the read to ds_read_b96, which takes 8 LDS cycles per wavefront
  // where ds_read_b128 takes 4 (MI355X_MICROARCH.md, LDS).
  typedef float lds_v4f __attribute__((ext_vector_type(4)));
  auto lds_texel = [](const float4 *p) {
    return *(const volatile lds_v4f __attribute__((address_space(3))) *)p;
  };
  auto fp_sample = [&](int e, const Bilin &b, int ox, int oy) {
    const float4 *q = fp + (eoff_y[e] + oy) * FS + (eoff_x[e] + ox);
    const lds_v4f t00 = lds_texel(q), t01 = lds_texel(q + 1), t10 = lds_texel(q + FS), t11 = lds_texel(q + FS + 1);
    // both gradients in one packed multiply / add each (v_pk_mul_f32, v_pk_add_f32: two
    // independent IEEE operations, the same roundings as the scalar expression)
    const f32x2 g = (f32x2{t00.x, t00.y} * b.w0 + f32x2{t01.x, t01.y} * b.w1) +
                    (f32x2{t10.x, t10.y} * b.w2 + f32x2{t11.x, t11.y} * b.w3);
    float3 o;
    o.x = (t00.z * b.w0 + t01.z * b.w1) + (t10.z * b.w2 + t11.z * b.w3);
    o.y = g.x;
    o.z = g.y;
    return o;
  };
  // RUN: all EPL elements of the lane's run, the upper texel row first and the lower one after it
  // (RUN + 1 texels in registers at a time instead of 2 RUN + 2); the same expression per element,
  // (t00 w0 + t01 w1) + (t10 w2 + t11 w3)
  auto fp_sample_run = [&](const Bilin &b, int ox, int oy, float3 (&out)[EPL]) {
    const float4 *q = fp + (eoff_y[0] + oy) * FS + (eoff_x[0] + ox);
    f32x2 ga[EPL];
    float ia[EPL];
    {
      lds_v4f t[EPL + 1];
#pragma unroll
      for (int e = 0; e <= EPL; ++e) t[e] = lds_texel(q + e);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        ga[e] = f32x2{t[e].x, t[e].y} * b.w0 + f32x2{t[e + 1].x, t[e + 1].y} * b.w1;
        ia[e] = t[e].z * b.w0 + t[e + 1].z * b.w1;
      }
    }
    {
      lds_v4f t[EPL + 1];
#pragma unroll
      for (int e = 0; e <= EPL; ++e) t[e] = lds_texel(q + FS + e);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const f32x2 g = ga[e] + (f32x2{t[e].x, t[e].y} * b.w2 + f32x2{t[e + 1].x, t[e + 1].y} * b.w3);
        out[e].x = ia[e] + (t[e].z * b.w2 + t[e + 1].z * b.w3);
        out[e].y = g.x;
        out[e].z = g.y;
      }
    }
  };
  auto sample_all = [&](const Bilin &b, int ox, int oy, float3 (&out)[EPL]) {
    if constexpr (RUN > 0) {
      fp_sample_run(b, ox, oy, out);
    } else {
#pragma unroll
      for (int e = 0; e < EPL; ++e) out[e] = fp_sample(e, b, ox, oy);
    }
  };
  // ordered reduction of nq quantities whose per-element products sit in prod[q][0..NA)
  auto reduce = [&](int nq, float (&tot)[5]) {
    __syncthreads();
    float acc = 0.f;
    if (lane < nq * SL) {
      const int q = lane >> 3, k = lane & 7;
      const float *p = prod + q * NAs + k;
      if constexpr (NEC > 0 && NEC <= 64) {
        // reads in flight in blocks of at most 28 (register pressure), adds in element order
        constexpr int HB = NEC > 32 ? (NEC + 1) / 2 : NEC;
#pragma unroll
        for (int h0 = 0; h0 < NEC; h0 += HB) {
          float t[HB];
#pragma unroll
          for (int u = 0; u < HB; ++u) t[u] = p[min(h0 + u, NEC - 1) * SL];
#pragma unroll
          for (int u = 0; u < HB; ++u)
            if (h0 + u < NEC) acc = acc + t[u];
        }
      } else {
        // groups of 8 independent LDS reads (clamped index), then the adds in element order
        for (int j0 = 0; j0 < nec; j0 += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = p[min(j0 + u, nec - 1) * SL];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (j0 + u < nec) acc = acc + t[u];
        }
      }
    }
    // The 8 partials of quantity q sit in lanes 8q .. 8q+7 and are added left to right
    // (LK.hpp:143-145) without leaving the registers: seven steps s[l] = s[l-1] + acc[l] (DPP
    // row_shr:1).  After step i lane 8q+i holds ((p0 + p1) + ...) + p_i, so lane 8q+7 ends with
    // the reference's sum; what the other lanes hold meanwhile never enters that chain.
    float t = acc;
#pragma unroll
    for (int i = 1; i < SL; ++i)
      t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t),
                                                                 0x111 /* row_shr:1 */, 0xf, 0xf, true)) +
          acc;
#pragma unroll
    for (int q = 0; q < 5; ++q)
      tot[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), q * SL + SL - 1));
  };

  // The grid is capped (launch_track_frames) so that a launch never takes every wave slot of
  // the chip away from the pyramid build it overlaps with; a wavefront then takes features
  // blockIdx.x, blockIdx.x + gridDim.x, ... of a permuted list: workgroups go round-robin over
  // the 8 XCDs (each with its own L2), so slot q is mapped to feature (q % 8) * ceil(n/8) + q / 8
  // and every XCD works on one contiguous eighth of the list.  Lists are spatially ordered in
  // practice (selection order, lattice seeds): neighbouring features, whose windows overlap at
  // the coarse levels, then share an L2.
  // Multi-stream launches run the same slot order stream after stream (8 * per_xcd slots each, so
  // q % 8 is still the XCD).
  const int per_xcd = ((MS ? M.nmax : a.n) + 7) >> 3;
  const int qs = 8 * per_xcd;
  const int qend = MS ? M.nstreams * qs : qs;
  // slot -> list entry (-1: none) and its (x, y, val).  The entry of the wavefront's NEXT slot is
  // loaded before the current feature is tracked, so that the loads' latency is not paid between
  // features (a launch visits every entry once, so nothing else writes it meanwhile).
  auto fetch = [&](int q, int &f, float &x, float &y, int &v) {
    f = -1;
    if (q >= qend) return;
    const int sidx = MS ? q / qs : 0;
    const int ql = q - sidx * qs;
    const int fl = (ql & 7) * per_xcd + (ql >> 3);
    if (fl >= (MS ? M.n[sidx] : a.n)) return;
    f = MS ? sidx * M.cap + fl : fl;
    v = fval[f];
    x = fx[f];
    y = fy[f];
  };
  unsigned w_tracked = 0, w_iters = 0;  // this wavefront's counters (< 2^32 per launch)
  int f_next, v_next = 0;
  float x_next = 0.f, y_next = 0.f;
  fetch(blockIdx.x, f_next, x_next, y_next, v_next);
  for (int q = blockIdx.x; q < qend; q += gridDim.x) {
  const int f = f_next;
  const int val0 = v_next;
  float px = x_next, py = y_next;  // the feature's position in the first frame of the launch
  fetch(q + gridDim.x, f_next, x_next, y_next, v_next);
  if (f < 0) continue;
  const int sidx = MS ? q / qs : 0;
  const Texel *const *const tab1 = MS ? M.tab + (size_t)(sidx * 3 + M.slot1) * GKLT_MAX_LEVELS : nullptr;
  const Texel *const *const tab2 = MS ? M.tab + (size_t)(sidx * 3 + M.slot2) * GKLT_MAX_LEVELS : nullptr;
  int vcur = val0;
  int iters_total = 0, ntracked = 0;
  for (int fr = 0; fr < F.nframes; ++fr) {
  if (vcur < 0) {  // LK.hpp:830: lost features are not touched
    if (F.mirror[fr] && lane == 0) {
      float *m = F.mirror[fr];
      m[f] = px;
      m[a.mirror_stride + f] = py;
      reinterpret_cast<int *>(m)[2 * a.mirror_stride + f] = vcur;
    }
    continue;
  }
  float xloc = px, yloc = py;
  // LK.hpp:837-839.  Dividing by a power of two is the same as multiplying by its (exact)
  // reciprocal; a correctly rounded fp32 division is ~12 instructions, and there are 2 * levels.
  if (a.inv_ss != 0.f) {
    for (int r = 0; r < a.nlevels; ++r) { xloc *= a.inv_ss; yloc *= a.inv_ss; }
  } else {
    for (int r = 0; r < a.nlevels; ++r) { xloc /= a.ss; yloc /= a.ss; }
  }
  float xo = xloc, yo = yloc;
  int v = GKLT_TRACKED;

  for (int r = a.nlevels - 1; r >= 0; --r) {  // LK.hpp:859-881
    xloc *= a.ss; yloc *= a.ss;
    xo *= a.ss; yo *= a.ss;
    // ---- _trackFeature, LK.hpp:413-669 ----
    const int nc = a.lv[r].nc, nr = a.lv[r].nr;
    const Texel *const lt1 = MS ? tab1[r] : F.tex[fr][r];
    const Texel *const lt2 = MS ? tab2[r] : F.tex[fr + 1][r];
    int status = GKLT_TRACKED, iteration = 0;
    const float x1 = xloc, y1 = yloc;
    float x2 = xo, y2 = yo;

    if (window_oob(x1, y1, hw, hh, nc, nr) || window_oob(x2, y2, hw, hh, nc, nr)) {
      status = GKLT_OOB;  // first pass of the do-loop, :558-564
    } else {
      Bilin b1 = bilinear_weight(x1, y1);
      clamp_origin(b1, hw, hh, nc, nr);
      const Texel *o1 = lt1 + (size_t)(b1.yt - hh) * nc + (b1.xt - hw);
      Bilin b2 = bilinear_weight(x2, y2);  // :568-572
      clamp_origin(b2, hw, hh, nc, nr);
      const Texel *o2 = lt2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);
      float3 T1[EPL], S2[EPL];
      [[maybe_unused]] Texel R2{};           // FRM: the lane's raw texel of the second image ...
      [[maybe_unused]] int cx2 = 0, cy2 = 0;  // ... and the pixel cell (clamped window origin) it was loaded for
      int X0 = 0, Y0 = 0;  // first texel of the second image's footprint in LDS
      // stages the footprint around window origin (wx, wy) unless the one in LDS covers it;
      // returns the origin's offset inside the footprint
      auto stage2 = [&](int wx, int wy, int &ox, int &oy) {
        ox = wx - X0; oy = wy - Y0;
        if (ox >= 0 && ox <= 2 * HL && oy >= 0 && oy <= 2 * HL) return;
        float3 f2[FP];
        X0 = wx - HL; Y0 = wy - HL;
        fp_load(lt2, nc, nr, X0, Y0, f2);
        __syncthreads();  // previous readers of fp are done
        fp_store(f2);
        __syncthreads();
        ox = HL; oy = HL;
      };
      if (STAGE) {
        float3 f1[FP], f2[FP];
        X0 = b2.xt - hw - HL; Y0 = b2.yt - hh - HL;
        fp_load(lt1, nc, nr, b1.xt - hw - HL, b1.yt - hh - HL, f1);  // both footprints are in flight together
        fp_load(lt2, nc, nr, X0, Y0, f2);
        __syncthreads();      // previous readers of fp are done
        fp_store(f1);
        __syncthreads();
        sample_all(b1, HL, HL, T1);
        __syncthreads();
        fp_store(f2);
        __syncthreads();
        sample_all(b2, HL, HL, S2);
      } else if constexpr (FRM) {
        const Texel r1 = o1[eoff_y[0] * nc + eoff_x[0]];
        R2 = o2[eoff_y[0] * nc + eoff_x[0]];  // both in flight together
        cx2 = b2.xt; cy2 = b2.yt;
        T1[0] = frm3(r1, b1);
        S2[0] = frm3(R2, b2);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) T1[e] = gather3(o1 + eoff_y[e] * nc + eoff_x[e], nc, b1);
#pragma unroll
        for (int e = 0; e < EPL; ++e) S2[e] = gather3(o2 + eoff_y[e] * nc + eoff_x[e], nc, b2);
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (!evalid[e]) T1[e] = make_float3(0.f, 0.f, 0.f);

      float dx, dy;
      for (;;) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int i = pidx[e];
          const float3 s = evalid[e] ? S2[e] : make_float3(0.f, 0.f, 0.f);
          const float diff = T1[e].x - s.x;  // computeImageDiff :110-116
          const float gx = T1[e].y + s.y;    // computeImageSum  :118-125
          const float gy = T1[e].z + s.z;
          if (RUN > 0 || i >= 0) {
            prod[0 * NAs + i] = gx * gx;      // compute2X2GradientMatrix :127-146
            prod[1 * NAs + i] = gy * gy;
            prod[2 * NAs + i] = gx * gy;
            prod[3 * NAs + i] = gx * diff;    // compute2X1ErrorVector :148-166
            prod[4 * NAs + i] = gy * diff;
          }
        }
        float tot[5];
        reduce(5, tot);
        const float gxx = tot[0], gyy = tot[1], gxy = tot[2];
        const float ex = tot[3] * a.step_factor, ey = tot[4] * a.step_factor;

        const float det = gxx * gyy - gxy * gxy;  // _solveEquation :168-183
        if (det < a.min_det) { status = GKLT_SMALL_DET; break; }
        dx = (gyy * ex - gxy * ey) / det;
        dy = (gxx * ey - gxy * ex) / det;
        status = GKLT_TRACKED;
        x2 += dx; y2 += dy;
        ++iteration;
        if (!((fabsf(dx) >= a.min_disp || fabsf(dy) >= a.min_disp) && iteration < a.max_iter))
          break;  // :615
        if (window_oob(x2, y2, hw, hh, nc, nr)) { status = GKLT_OOB; break; }  // next pass :558
        b2 = bilinear_weight(x2, y2);
        clamp_origin(b2, hw, hh, nc, nr);
        o2 = lt2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);
        if (STAGE) {
          int ox, oy;
          stage2(b2.xt - hw, b2.yt - hh, ox, oy);
          sample_all(b2, ox, oy, S2);
        } else if constexpr (FRM) {
          if (b2.xt != cx2 || b2.yt != cy2) {  // wave-uniform
            R2 = o2[eoff_y[0] * nc + eoff_x[0]];
            cx2 = b2.xt; cy2 = b2.yt;
          }
          S2[0] = frm3(R2, b2);
        } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) S2[e] = gather3(o2 + eoff_y[e] * nc + eoff_x[e], nc, b2);
        }
      }

      if (window_oob(x2, y2, hw, hh, nc, nr)) status = GKLT_OOB;  // :620-623

      // residue (:626-636), level 0 only: see k_track
      if (status == GKLT_TRACKED && r == 0) {
        b2 = bilinear_weight(x2, y2);
        clamp_origin(b2, hw, hh, nc, nr);
        o2 = lt2 + (size_t)(b2.yt - hh) * nc + (b2.xt - hw);
        float sI[EPL];
        if (STAGE) {
          int ox, oy;
          stage2(b2.xt - hw, b2.yt - hh, ox, oy);
#pragma unroll
          for (int e = 0; e < EPL; ++e) sI[e] = fp_sample(e, b2, ox, oy).x;
        } else if constexpr (FRM) {
          if (b2.xt != cx2 || b2.yt != cy2) R2.x = o2[eoff_y[0] * nc + eoff_x[0]].x;
          sI[0] = frm1(R2.x, b2);
        } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) sI[e] = gather1(o2 + eoff_y[e] * nc + eoff_x[e], nc, b2);
        }
        __syncthreads();  // chain sums of the last iteration have been read
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int i = pidx[e];
          if (RUN > 0 || i >= 0) prod[i] = evalid[e] ? fabsf(T1[e].x - sI[e]) : 0.f;  // :193-208
        }
        float tot[5];
        reduce(1, tot);
        if (tot[0] / (ww * wh) > a.max_residue) status = GKLT_LARGE_RESIDUE;
      }
    }
    iters_total += iteration;
    xo = x2; yo = y2;
    // return value precedence, :655-667
    if (status == GKLT_SMALL_DET || status == GKLT_OOB || status == GKLT_LARGE_RESIDUE) v = status;
    else v = iteration >= a.max_iter ? GKLT_MAX_ITERATIONS : GKLT_TRACKED;
    if (v == GKLT_SMALL_DET || v == GKLT_OOB) break;  // :879-880
    __syncthreads();  // prod is rewritten by the next level
  }

  // ---- record feature, LK.hpp:886-915 ----
  const bool oob = xo < a.borderx || xo > a.w0 - 1 - a.borderx || yo < a.bordery ||
                   yo > a.h0 - 1 - a.bordery;  // _outOfBounds :185-191
  const int out = (v == GKLT_OOB || oob) ? GKLT_OOB : v;
  const bool tracked = out == GKLT_TRACKED;
  px = tracked ? xo : -1.0f;
  py = tracked ? yo : -1.0f;
  vcur = out;
  ntracked += tracked;
  if (lane == 0) {
    if (F.mirror[fr]) {
      float *m = F.mirror[fr];
      m[f] = px;
      m[a.mirror_stride + f] = py;
      reinterpret_cast<int *>(m)[2 * a.mirror_stride + f] = out;
    }
    if (tracked && a.do_epilogue) {  // bearing vectors, LK.hpp:1019-1035 (fp64)
      const double qx = (double)xo, qy = (double)yo;
      double q[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] = (invk[3 * i] * qx + invk[3 * i + 1] * qy) + invk[3 * i + 2];
      const double nn = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
      double nv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { nv[i] = q[i] / nn; norm3[3 * (size_t)f + i] = nv[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) vec3[3 * (size_t)f + i] = nv[i] / nv[2];
    }
  }
  __syncthreads();  // LDS is reused by the next frame
  }  // frames

  if (val0 >= 0) {
    if (lane == 0) {
      fx[f] = px;
      fy[f] = py;
      fval[f] = vcur;
    }
    w_tracked += ntracked;
    w_iters += iters_total;
  }
  }  // features of this wavefront
  // Counters: one pair of atomics per wavefront, after its last feature (on gfx9 stores and atomics
  // count in vmcnt like loads, so an atomic per feature would make the next feature's first
  // gather wait for its round trip); sharded, because thousands of adds to one word would
  // serialise at the memory side.
  if (stats && lane == 0 && (w_tracked | w_iters)) {
    unsigned long long *sh = stats + 2 * (blockIdx.x & (kStatShards - 1));
    if (w_tracked) atomicAdd(&sh[0], (unsigned long long)w_tracked);
    atomicAdd(&sh[1], (unsigned long long)w_iters);
  }
}

}  // namespace

int launch_track(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *p1, const gklt_pyramid *p2,
                 int n, float *x, float *y, int *val, const double *d_invk, double *d_norm,
                 double *d_vec, unsigned long long *stats, float *mirror, int mirror_stride) {
  const gklt_pyramid *py[2] = {p1, p2};
  return launch_track_frames(ctx, s, py, 1, n, x, y, val, d_invk, d_norm, d_vec, stats, &mirror,
                             mirror_stride);
}

namespace {

void fill_args(const gklt_params &P, const gklt_pyramid *py0, int n, bool epilogue,
               int mirror_stride, TrackArgs *out) {
  TrackArgs a{};
  a.nlevels = P.nPyramidLevels;
  a.ss = (float)P.subsampling;  // LK.hpp:978
  a.inv_ss = (P.subsampling & (P.subsampling - 1)) == 0 ? 1.0f / (float)P.subsampling : 0.f;
  a.ww = P.window_width;
  a.wh = P.window_height;
  a.ww_magic = (65536 + a.ww - 1) / a.ww;
  a.step_factor = P.step_factor;
  a.min_det = P.min_determinant;
  a.min_disp = P.min_displacement;
  a.max_residue = P.max_residue;
  a.max_iter = P.max_iterations;
  a.w0 = py0->lv[0].w;
  a.h0 = py0->lv[0].h;
  a.borderx = P.borderx;
  a.bordery = P.bordery;
  a.n = n;
  a.do_epilogue = epilogue;
  a.mirror_stride = mirror_stride;
  for (int l = 0; l < P.nPyramidLevels; ++l)
    a.lv[l] = TrackLevel{nullptr, nullptr, py0->lv[l].w, py0->lv[l].h};
  *out = a;
}

struct WaveGeom {
  int NA, NF, epl, fpl, run, halo, fstride;
  bool stage, fits, frm;
  size_t lds;
};

WaveGeom wave_geom_compute(const gklt_ctx *ctx, int ww, int wh);

// (the stride search below is a few hundred microseconds of host time: once per window and mapping)
WaveGeom wave_geom(const gklt_ctx *ctx, int ww, int wh) {
  struct Key { int ww, wh, impl; WaveGeom g; };
  static thread_local Key last{-1, -1, -1, {}};
  if (last.ww != ww || last.wh != wh || last.impl != ctx->track_impl)
    last = Key{ww, wh, ctx->track_impl, wave_geom_compute(ctx, ww, wh)};
  return last.g;
}

WaveGeom wave_geom_compute(const gklt_ctx *ctx, int ww, int wh) {
  WaveGeom g;
  g.NA = 8 * ((ww * wh + 8) / 8);
  g.halo = g.run = 0;
  g.NF = (ww + 1) * (wh + 1);
  // Footprint staging for every window that does not fit the footprint-in-registers kernel
  // ((w+1)(h+1) > 64); track_impl 2 / 3 force gathers / staging.  Rounds 1-2 staged windows of 200
  // elements and more only (11x11 was 86 us with gathers, 98 us staged at C3); since the staged
  // texels are read with ds_read_b128 (round 3) staging wins from 9x9 up in the lookahead pipeline,
  // where the gathers compete with the build for the texture path -- gathers / staged, M
  // feature-frames/s at the C3 geometry: 9x9 154 / 158, 11x11 145 / 154, 13x13 97 / 107
  // (tools/exp_la_window.py); a single-frame launch alone is as fast either way.
  g.stage = ctx->track_impl == 3 || (ctx->track_impl == 0 && (ww + 1) * (wh + 1) > 64);
  if (g.stage) {
    static const int env_halo = getenv("GKLT_TRACK_HALO") ? atoi(getenv("GKLT_TRACK_HALO")) : 0;
    g.halo = std::min(std::max(env_halo, 0), 4);
    g.NF = (ww + 1 + 2 * g.halo) * (wh + 1 + 2 * g.halo);
  }
  // row-run mapping (k_track_w, RUN = 7): three runs of seven cover a row of 15..21 elements and
  // 3 * wh lanes must exist; track_impl 3 keeps the element-per-lane mapping for A/B runs
  g.run = g.stage && ctx->track_impl == 0 && ww >= 15 && ww <= 21 && 3 * wh <= 64 ? 7 : 0;
  // LDS row stride of the footprint.  With the row-run mapping lane l reads 16 bytes at texel
  // (l / 3 + dy) * stride + (l % 3) * 7 + e: the stride decides how the lanes spread over the
  // banks (width 24 -> two rows apart is 3 * 64 words: the same banks; 29 spreads them).  Chosen by
  // counting the bank passes of the 16 reads of a pass under a 16-lane, 32- and 64-bank model.
  g.fstride = ww + 1 + 2 * g.halo;
  if (g.run) {
    auto passes = [&](int fs, int nbanks) {
      int tot = 0;
      for (int dy = 0; dy < 2; ++dy)
        for (int e = 0; e <= g.run; ++e)
          for (int g0 = 0; g0 < 64; g0 += 16) {
            int cnt[64] = {0}, worst = 0;
            for (int l = g0; l < g0 + 16; ++l) {
              const int r = std::min(l / 3, wh - 1), c0 = (l % 3) * g.run;
              const int a0 = ((r + dy) * fs + c0 + e) * 4;
              bool dup = false;  // same texel as an earlier lane of the group: one access
              for (int m = g0; m < l; ++m)
                dup |= std::min(m / 3, wh - 1) == r && (m % 3) * g.run == c0;
              if (dup) continue;
              for (int w = 0; w < 4; ++w) worst = std::max(worst, ++cnt[(a0 + w) % nbanks]);
            }
            tot += worst;
          }
      return tot;
    };
    int best = 1 << 30;
    for (int fs = ww + 1 + 2 * g.halo; fs <= ww + 1 + 2 * g.halo + 8; ++fs) {
      const int c = passes(fs, 32) + passes(fs, 64);
      if (c < best) { best = c; g.fstride = fs; }
    }
  }
  g.epl = g.run ? g.run : (g.NA + 63) / 64;
  g.fpl = (g.NF + 63) / 64;
  // the row-run kernels are instantiated for footprints of up to 13 texels per lane: a wider
  // footprint (GKLT_TRACK_HALO 4 with a 19..21-px window) takes the element-per-lane mapping
  if (g.run && g.fpl > 13) {
    g.run = 0;
    g.fstride = ww + 1 + 2 * g.halo;
    g.epl = (g.NA + 63) / 64;
  }
  g.lds = sizeof(float) * (size_t)(5 * prod_stride(g.NA) + 40) + (g.stage ? sizeof(float4) * (size_t)(g.fstride * (wh + 1 + 2 * g.halo) + 8) : 0);
  g.fits = g.epl <= 16 && g.fpl <= 16 && g.lds <= 48 * 1024;
  // the lane mappings divide by the window and footprint widths with a 16-bit reciprocal
  // (TrackArgs::ww_magic, fw_magic): exact for every index they see, or the window does not fit
  for (int d : {ww, ww + 1 + 2 * g.halo}) {
    const int magic = (65536 + d - 1) / d;
    for (int j = 0; j < 64 * 16; ++j) g.fits = g.fits && ((j * magic) >> 16) == j / d;
  }
  // footprint-in-registers sampling: the whole (w+1)(h+1) footprint in one wavefront
  // (track_impl 2 keeps the four gathers per element for A/B runs)
  g.frm = !g.stage && g.NF <= 64 && ctx->track_impl != 2;
  return g;
}

// picks the instantiation of the wave-per-feature kernel for the window
template <bool MS>
void dispatch_w(const WaveGeom &g, int nwaves, hipStream_t s, const TrackArgs &a,
                const TrackFrames &F, const TrackStreams &M, float *x, float *y, int *val,
                const double *d_invk, double *d_norm, double *d_vec, unsigned long long *stats) {
  const dim3 block(64);
  const size_t wlds = g.lds;
  const int epl = g.epl, fpl = g.fpl;
#define GKLT_TRACK_RUN7(FP, NE)                                                                  \
  hipLaunchKernelGGL((k_track_w<7, FP, NE, MS, false, 7>), dim3(nwaves), block, wlds, s, a, F, M, \
                     x, y, val, d_invk, d_norm, d_vec, stats)
#define GKLT_TRACK_W(E, FP, NE)                                                                 \
  hipLaunchKernelGGL((k_track_w<E, FP, NE, MS>), dim3(nwaves), block, wlds, s, a, F, M, x, y, val, \
                     d_invk, d_norm, d_vec, stats)
#define GKLT_TRACK_FRM(NE)                                                                      \
  hipLaunchKernelGGL((k_track_w<1, 0, NE, MS, true>), dim3(nwaves), block, wlds, s, a, F, M, x, y, \
                     val, d_invk, d_norm, d_vec, stats)
  const int nec = g.NA / 8;
  if (g.run == 7) {
    if (nec == 56 && fpl <= 9) GKLT_TRACK_RUN7(9, 56);  // 21x21
    else if (fpl <= 9) GKLT_TRACK_RUN7(9, 0);
    else GKLT_TRACK_RUN7(13, 0);  // margins up to 4
  } else if (!g.stage) {
    if (g.frm && nec == 7) GKLT_TRACK_FRM(7);  // 7x7, footprint in registers
    else if (g.frm) GKLT_TRACK_FRM(0);
    else if (epl <= 1 && nec == 7) GKLT_TRACK_W(1, 0, 7);  // 7x7
    else if (epl <= 1) GKLT_TRACK_W(1, 0, 0);
    else if (epl <= 2 && nec == 16) GKLT_TRACK_W(2, 0, 16);  // 11x11
    else if (epl <= 2) GKLT_TRACK_W(2, 0, 0);
    else if (epl <= 4) GKLT_TRACK_W(4, 0, 0);
    else if (epl <= 8) GKLT_TRACK_W(8, 0, 0);
    else GKLT_TRACK_W(16, 0, 0);
  } else if (epl <= 2 && fpl <= 3 && nec == 16) {
    GKLT_TRACK_W(2, 3, 16);  // 11x11 staged (track_impl 3)
  } else if (epl <= 7 && fpl <= 9 && nec == 56) {
    GKLT_TRACK_W(7, 9, 56);  // 21x21
  } else if (epl <= 1) {
    if (fpl <= 1) GKLT_TRACK_W(1, 1, 0); else GKLT_TRACK_W(1, 2, 0);
  } else if (epl <= 2) {
    if (fpl <= 2) GKLT_TRACK_W(2, 2, 0); else GKLT_TRACK_W(2, 4, 0);
  } else if (epl <= 4) {
    if (fpl <= 4) GKLT_TRACK_W(4, 4, 0); else GKLT_TRACK_W(4, 8, 0);
  } else if (epl <= 8) {
    if (fpl <= 8) GKLT_TRACK_W(8, 8, 0); else GKLT_TRACK_W(8, 16, 0);
  } else {
    GKLT_TRACK_W(16, 16, 0);
  }
#undef GKLT_TRACK_W
#undef GKLT_TRACK_RUN7
#undef GKLT_TRACK_FRM
}

}  // namespace

// One launch for the lists of nstreams independent streams: stream i is tracked from its ring
// slot slot1 to slot2 (texel planes in the device table tab, [nstreams][3][GKLT_MAX_LEVELS]) and
// owns list entries [i * cap, i * cap + n[i]).  py0: any pyramid of the set (sizes).  mirror
// (optional, host-visible) receives x | y | val with stride mirror_stride.
int launch_track_streams(gklt_ctx *ctx, hipStream_t s, const Texel *const *tab, int nstreams,
                         int slot1, int slot2, int cap, const int *n, const gklt_pyramid *py0,
                         float *x, float *y, int *val, unsigned long long *stats, float *mirror,
                         int mirror_stride) {
  const gklt_params &P = ctx->prm;
  if (nstreams < 1 || nstreams > GKLT_MAX_STREAMS) return GKLT_EINVAL;
  if (py0->nlevels != P.nPyramidLevels || py0->ss != P.subsampling) {
    set_error("pyramids were built for %d levels / subsampling %d, context has %d / %d",
              py0->nlevels, py0->ss, P.nPyramidLevels, P.subsampling);
    return GKLT_EINVAL;
  }
  const WaveGeom g = wave_geom(ctx, P.window_width, P.window_height);
  if (!g.fits) {
    set_error("window %dx%d is too large for the wave-per-feature tracker of multi-stream launches",
              P.window_width, P.window_height);
    return GKLT_EUNSUPPORTED;
  }
  TrackStreams M{};
  M.tab = tab;
  M.nstreams = nstreams;
  M.slot1 = slot1;
  M.slot2 = slot2;
  M.cap = cap;
  long total = 0;
  for (int i = 0; i < nstreams; ++i) {
    M.n[i] = n[i];
    M.nmax = std::max(M.nmax, n[i]);
    total += n[i];
  }
  if (total <= 0) return 0;
  TrackArgs a;
  fill_args(P, py0, M.nmax, false, mirror_stride, &a);
  a.halo = g.halo;
  a.fstride = g.fstride;
  a.fw_magic = (65536 + P.window_width + 2 * g.halo) / (P.window_width + 1 + 2 * g.halo);
  TrackFrames F{};
  F.nframes = 1;
  F.mirror[0] = mirror;
  // slots of the launch: nstreams x 8 x ceil(nmax / 8); the default cap of 4096 wavefronts leaves
  // room for the batched build the launch overlaps with
  const long slots = (long)nstreams * 8 * ((M.nmax + 7) / 8);
  const int nwaves = (int)std::min<long>(slots, ctx->track_waves > 0 ? ctx->track_waves : 4096);
  ProfScope ps(ctx, s, GKLT_K_TRACK);
  dispatch_w<true>(g, nwaves, s, a, F, M, x, y, val, nullptr, nullptr, nullptr, stats);
  GKLT_HIP(hipGetLastError());
  return 0;
}

// Tracks the list through nframes consecutive pairs (py[j], py[j+1]), j = 0..nframes-1; mirrors[j]
// (optional, host-visible) receives the list after frame j.  One launch when the wave-per-feature
// kernel applies, nframes launches otherwise.
int launch_track_frames(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *const *py, int nframes,
                        int n, float *x, float *y, int *val, const double *d_invk, double *d_norm,
                        double *d_vec, unsigned long long *stats, float *const *mirrors,
                        int mirror_stride) {
  const gklt_params &P = ctx->prm;
  if (nframes < 1 || nframes > GKLT_MAX_LOOKAHEAD) return GKLT_EINVAL;
  for (int j = 0; j <= nframes; ++j) {
    if (py[j]->nlevels != P.nPyramidLevels || py[j]->ss != P.subsampling) {
      set_error("pyramids were built for %d levels / subsampling %d, context has %d / %d",
                py[j]->nlevels, py[j]->ss, P.nPyramidLevels, P.subsampling);
      return GKLT_EINVAL;
    }
    for (int l = 0; l < P.nPyramidLevels; ++l)
      if (py[j]->lv[l].w != py[0]->lv[l].w || py[j]->lv[l].h != py[0]->lv[l].h) {
        set_error("pyramid sizes differ at level %d", l);
        return GKLT_EINVAL;
      }
  }
  TrackArgs a;
  fill_args(P, py[0], n, d_invk && d_norm && d_vec, mirror_stride, &a);
  if (n <= 0) return 0;

  const WaveGeom g = wave_geom(ctx, a.ww, a.wh);
  a.halo = g.halo;
  a.fstride = g.fstride;
  a.fw_magic = (65536 + a.ww + 2 * g.halo) / (a.ww + 1 + 2 * g.halo);
  const dim3 block(64);

  if (ctx->track_impl != 1 && g.fits) {  // feature per wave
    if (a.do_epilogue && nframes != 1) {
      set_error("bearing vectors are produced by single-frame launches only");
      return GKLT_EINVAL;
    }
    // Default cap for multi-frame launches: 4 wavefronts per SIMD (256 CUs x 4 SIMDs).  A launch
    // that takes more of the chip's wave slots for the length of a whole batch starves the pyramid
    // build it overlaps with (C3, 16000 features, lookahead 16: cap 4096 -> 117 M, 8192 -> 112 M,
    // uncapped -> 106 M feature-frames/s).  Single-frame launches are short and latency matters:
    // one wavefront per feature.  (Round 3, re-measured with the faster level kernels and the
    // staged 11 x 11 kernel: C3 164.4 / 162.9 / 164.8 M at 4096 / 6144 / 8192, 16 streams of C2
    // 196 M at 4096 and 177 M at 6144: tools/sweep_waves.sh.)
    const int cap = 4096;
    const int nwaves = std::min(n, ctx->track_waves > 0 ? ctx->track_waves : (nframes > 1 ? cap : n));
    TrackFrames F{};
    F.nframes = nframes;
    for (int j = 0; j <= nframes; ++j)
      for (int l = 0; l < P.nPyramidLevels; ++l) F.tex[j][l] = py[j]->lv[l].tex;
    for (int j = 0; j < nframes; ++j) F.mirror[j] = mirrors ? mirrors[j] : nullptr;
    ProfScope ps(ctx, s, GKLT_K_TRACK);
    dispatch_w<false>(g, nwaves, s, a, F, TrackStreams{}, x, y, val, d_invk, d_norm, d_vec, stats);
    GKLT_HIP(hipGetLastError());
    return 0;
  }

  // one feature per 8-lane slice: one launch per frame pair
  const dim3 grid((n + 7) / 8);
  for (int j = 0; j < nframes; ++j) {
    for (int l = 0; l < P.nPyramidLevels; ++l)
      a.lv[l] = TrackLevel{py[j]->lv[l].tex, py[j + 1]->lv[l].tex, py[0]->lv[l].w, py[0]->lv[l].h};
    a.mirror = mirrors ? mirrors[j] : nullptr;
    ProfScope ps(ctx, s, GKLT_K_TRACK);
    if (a.ww == 7 && a.wh == 7) {
      hipLaunchKernelGGL((k_track<7, 7>), grid, block, 0, s, a, x, y, val, d_invk, d_norm, d_vec, stats);
    } else if (a.ww == 11 && a.wh == 11) {
      hipLaunchKernelGGL((k_track<11, 11>), grid, block, 0, s, a, x, y, val, d_invk, d_norm, d_vec, stats);
    } else {
      const int ne = (a.ww * a.wh + 8) / 8;
      const size_t lds = (size_t)3 * ne * 64 * sizeof(float);
      if (lds > 160 * 1024) {
        set_error("window %dx%d needs %zu bytes of LDS per wave (> 160 KiB)", a.ww, a.wh, lds);
        return GKLT_EINVAL;
      }
      if (lds > 48 * 1024)
        GKLT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_track<0, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_track<0, 0>), grid, block, lds, s, a, x, y, val, d_invk, d_norm, d_vec, stats);
    }
    GKLT_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace gklt
