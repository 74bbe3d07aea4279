// Shi-Tomasi feature selection for gfx950 -- replaces _KLTSelectGoodFeatures (plain and rounding
// "2" flavours, selectGoodFeatures.hpp:234-449, 454-806, 997-1022) and the integral-image
// producer its inputs presuppose (System.cpp:385-392).
//
//  S6  k_integral_rows / k_integral_cols : gx*gx, gy*gy, gx*gy and their fp32 integral images in
//      cv::integral's summation order (running row prefix, then sum(y+1,x+1) = sum(y,x+1) + prefix),
//      because the selector's scores are differences of sums ~1e9 and change with that order.
//  S1  k_score : min-eigenvalue score of every interior pixel from 12 integral-image taps.
//  S2/S3  The reference sorts all candidates and walks them serially, accepting a candidate when
//      no accepted one lies within Chebyshev distance r.  A candidate's fate depends only on
//      higher-priority candidates within r, so the same set is the fixpoint of a parallel rule:
//      an undecided candidate is REJECTED once an accepted one lies within r, and ACCEPTED once it
//      is the highest-priority undecided candidate within r.  k_mis_round applies the rule to every
//      pixel (separable max of 64-bit priority keys through LDS); rounds repeat until nothing is
//      undecided.  Priority = (val desc, y asc, x asc), the build's total order where the
//      reference's qsort leaves ties unspecified.  The serial walk stops after nFeatures
//      acceptances, i.e. it keeps the nFeatures highest-priority members of that set:
//      k_rank_assign ranks the accepted set and writes slot[rank].
#include <algorithm>

#include "gklt_internal.h"

namespace gklt {

namespace {

constexpr unsigned kAccepted = 0x80000000u;

// ------------------------------------------------------------------ S6 integral images
// One wave per 64 rows.  Texels of a 64x64 tile are read row-contiguously, the products are
// transposed through LDS (stride 65: conflict-free both ways), each lane extends the running
// prefix of its row over 64 columns, and the prefixes are written back row-contiguously.
__global__ __launch_bounds__(64) void k_integral_rows(const Texel *__restrict__ tex, int w, int h,
                                                      float *__restrict__ sxx,
                                                      float *__restrict__ syy,
                                                      float *__restrict__ sxy) {
  __shared__ float t[3][64][65];
  const int lane = threadIdx.x, y0 = blockIdx.x * 64, w1 = w + 1;
  float rxx = 0.f, ryy = 0.f, rxy = 0.f;  // running prefix of row y0+lane
  for (int x0 = 0; x0 < w; x0 += 64) {
    const int x = x0 + lane;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      // clamped, unconditional load (keeps the 8 loads of an unrolled group in flight)
      const Texel v = tex[(size_t)min(y0 + r, h - 1) * w + min(x, w - 1)];
      const bool in = x < w && y0 + r < h;
      const float a = in ? v.y : 0.f, b = in ? v.z : 0.f;
      t[0][r][lane] = a * a;  // cv::multiply, System.cpp:386-388
      t[1][r][lane] = b * b;
      t[2][r][lane] = a * b;
    }
    __syncthreads();
    const int nc = min(64, w - x0);
    for (int c = 0; c < nc; ++c) {
      rxx = rxx + t[0][lane][c]; t[0][lane][c] = rxx;
      ryy = ryy + t[1][lane][c]; t[1][lane][c] = ryy;
      rxy = rxy + t[2][lane][c]; t[2][lane][c] = rxy;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      if (x < w && y0 + r < h) {
        const size_t o = (size_t)(y0 + r + 1) * w1 + x + 1;
        sxx[o] = t[0][r][lane];
        syy[o] = t[1][r][lane];
        sxy[o] = t[2][r][lane];
      }
    }
    __syncthreads();
  }
}

// sum(y+1, x+1) = sum(y, x+1) + prefix_y(x), in place; row 0 and column 0 are zero.
__global__ void k_integral_cols(float *__restrict__ sxx, float *__restrict__ syy,
                                float *__restrict__ sxy, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;  // 0..w (column of the (h+1)x(w+1) image)
  const int w1 = w + 1;
  if (x > w) return;
  float *s = blockIdx.y == 0 ? sxx : (blockIdx.y == 1 ? syy : sxy);
  s[x] = 0.f;
  if (x == 0) {
    for (int y = 1; y <= h; ++y) s[(size_t)y * w1] = 0.f;
    return;
  }
  float acc = 0.f;
#pragma unroll 8
  for (int y = 1; y <= h; ++y) {
    acc = acc + s[(size_t)y * w1 + x];
    s[(size_t)y * w1 + x] = acc;
  }
}

// ------------------------------------------------------------------ S1 score
// state[y*w+x] = score (1..INT_MAX) of an emitted candidate, else 0.
// selectGoodFeatures.hpp:324-366 (plain), :707-751 (rounding flavour).
__global__ void k_score(const float *__restrict__ sxx, const float *__restrict__ syy,
                        const float *__restrict__ sxy, const unsigned char *__restrict__ blocked,
                        int w, int h, int hw, int hh, int bx, int by, int step, int min_eig,
                        int round2, unsigned *__restrict__ state) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  unsigned out = 0;
  const bool on_grid = x >= bx && x < w - bx && y >= by && y < h - by &&
                       (x - bx) % step == 0 && (y - by) % step == 0;
  if (on_grid && !(blocked && blocked[(size_t)y * w + x])) {
    const int w1 = w + 1;
    const size_t r0 = (size_t)(y - hh) * w1, r1 = (size_t)(y + hh + 1) * w1;
    const int x1 = x - hw, x2 = x + hw + 1;
    const float gxy = sxy[r0 + x1] + sxy[r1 + x2] - sxy[r1 + x1] - sxy[r0 + x2];
    const float gxx = sxx[r0 + x1] + sxx[r1 + x2] - sxx[r1 + x1] - sxx[r0 + x2];
    const float gyy = syy[r0 + x1] + syy[r1 + x2] - syy[r1 + x1] - syy[r0 + x2];
    // _minEigenvalue, :220-223
    const float v = (gxx + gyy - sqrtf((gxx - gyy) * (gxx - gyy) + 4 * gxy * gxy)) / 2.0f;
    int iv = 0;
    if (round2) {
      if (v >= 2147483648.0f) iv = 0x7fffffff;
      else if (v > -2147483648.0f) iv = (int)((double)v + 0.5);  // :737
      if (!(iv > min_eig)) iv = 0;                                 // :740
    } else {
      if (v > (float)min_eig) iv = v >= 2147483648.0f ? 0x7fffffff : (int)v;  // :351-361
    }
    // _enforceMinimumDistance only ever accepts val >= max(1, min_eigenvalue) (:103-104,175)
    if (iv >= 1 && iv >= min_eig) out = (unsigned)iv;
  }
  state[(size_t)y * w + x] = out;
}

// featuremap pre-fill for surviving features when replacing (:280-288): radius mindist
__global__ void k_block_squares(const float *__restrict__ fx, const float *__restrict__ fy,
                                const int *__restrict__ fval, int n, int w, int h, int rad,
                                unsigned char *__restrict__ blocked) {
  const int f = blockIdx.x;
  if (f >= n || fval[f] < 0) return;
  const int cx = (int)fx[f], cy = (int)fy[f];
  const int side = 2 * rad + 1;
  for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
    const int x = cx - rad + i % side, y = cy - rad + i / side;
    if (x >= 0 && x < w && y >= 0 && y < h) blocked[(size_t)y * w + x] = 1;
  }
}

// ------------------------------------------------------------------ S2/S3 fixpoint round
constexpr int MW = 64, MH = 16, MT = 256;

__device__ __forceinline__ unsigned long long prio_key(unsigned val, int x, int y) {
  return ((unsigned long long)val << 32) | ((unsigned long long)(0xFFFF - y) << 16) |
         (unsigned long long)(0xFFFF - x);
}

// counters[0] = undecided candidates left after this round (summed over tiles)
__global__ __launch_bounds__(MT) void k_mis_round(unsigned *__restrict__ state, int w, int h,
                                                  int r, unsigned *__restrict__ counters,
                                                  int round) {
  extern __shared__ __align__(16) unsigned long long lds64[];
  const int inW = MW + 2 * r, inH = MH + 2 * r;
  unsigned long long *key = lds64;             // inH x inW : key of undecided, 0 otherwise
  unsigned long long *rmax = key + inW * inH;  // inH x MW  : row-window max (bit 0 of low word
                                               //             is not used by keys: y,x < 65535)
  unsigned char *acc = reinterpret_cast<unsigned char *>(rmax + inH * MW);  // inH x inW accepted
  unsigned char *racc = acc + inW * inH;                                    // inH x MW
  __shared__ int any_undecided;
  const int x0 = blockIdx.x * MW, y0 = blockIdx.y * MH;
  if (threadIdx.x == 0) any_undecided = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < inW * inH; i += MT) {
    const int ty = i / inW, tx = i - ty * inW;
    const int x = x0 - r + tx, y = y0 - r + ty;
    unsigned s = 0;
    if (x >= 0 && x < w && y >= 0 && y < h) s = state[(size_t)y * w + x];
    const bool a = (s & kAccepted) != 0;
    key[i] = (s != 0 && !a) ? prio_key(s, x, y) : 0ull;
    acc[i] = a;
    if (s != 0 && !a && tx >= r && tx < r + MW && ty >= r && ty < r + MH) mine = 1;
  }
  if (mine) any_undecided = 1;
  __syncthreads();
  if (!any_undecided) return;  // nothing to decide in this tile
  for (int i = threadIdx.x; i < inH * MW; i += MT) {
    const int ty = i / MW, tx = i % MW;
    const unsigned long long *k = key + ty * inW + tx;
    const unsigned char *a = acc + ty * inW + tx;
    unsigned long long m = 0;
    unsigned char o = 0;
    for (int j = 0; j <= 2 * r; ++j) {
      m = max(m, k[j]);
      o |= a[j];
    }
    rmax[i] = m;
    racc[i] = o;
  }
  __syncthreads();
  int left = 0;
  for (int i = threadIdx.x; i < MH * MW; i += MT) {
    const int ty = i / MW, tx = i % MW;
    const unsigned long long own = key[(ty + r) * inW + tx + r];
    if (own == 0) continue;
    unsigned long long m = 0;
    unsigned char o = 0;
    for (int j = 0; j <= 2 * r; ++j) {
      m = max(m, rmax[(ty + j) * MW + tx]);
      o |= racc[(ty + j) * MW + tx];
    }
    const size_t gi = (size_t)(y0 + ty) * w + x0 + tx;
    if (o) state[gi] = 0;                                     // an accepted neighbour covers it
    else if (m == own) state[gi] = (unsigned)(own >> 32) | kAccepted;
    else ++left;
  }
  if (left) atomicAdd(&counters[round], (unsigned)left);
}

// accepted pixels -> key list (order irrelevant, ranked later)
__global__ void k_collect(const unsigned *__restrict__ state, int w, int h,
                          unsigned long long *__restrict__ list, unsigned *__restrict__ count,
                          unsigned cap) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const unsigned s = state[(size_t)y * w + x];
  if (s & kAccepted) {
    const unsigned i = atomicAdd(count, 1u);
    if (i < cap) list[i] = prio_key(s & ~kAccepted, x, y);
  }
}

// slots to fill, in list order: every slot (selecting all) or the lost ones (replacing)
__global__ void k_lost_slots(const int *__restrict__ fval, int n, int overwrite,
                             int *__restrict__ slots, int *__restrict__ nslots) {
  // single block; n is a feature-list length (thousands)
  __shared__ int base;
  __shared__ int wsum[16];
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x;
    const int flag = i < n && (overwrite || fval[i] < 0);
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (flag) slots[off + before] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += wsum[k];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *nslots = base;
}

// rank of each accepted key among all accepted keys (descending); rank < nslots -> slot[rank]
__global__ __launch_bounds__(256) void k_rank_assign(const unsigned long long *__restrict__ list,
                                                     const unsigned *__restrict__ count,
                                                     unsigned cap, const int *__restrict__ slots,
                                                     const int *__restrict__ nslots,
                                                     float *__restrict__ fx,
                                                     float *__restrict__ fy,
                                                     int *__restrict__ fval) {
  __shared__ unsigned long long tile[256];
  const unsigned n = min(*count, cap);
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long own = i < n ? list[i] : 0ull;
  unsigned rank = 0;
  for (unsigned j0 = 0; j0 < n; j0 += 256) {
    tile[threadIdx.x] = j0 + threadIdx.x < n ? list[j0 + threadIdx.x] : 0ull;
    __syncthreads();
    const unsigned m = min(256u, n - j0);
    for (unsigned j = 0; j < m; ++j) rank += tile[j] > own;
    __syncthreads();
  }
  if (i < n && (int)rank < *nslots) {
    const int s = slots[rank];
    fx[s] = (float)(0xFFFF - (int)(own & 0xFFFF));
    fy[s] = (float)(0xFFFF - (int)((own >> 16) & 0xFFFF));
    fval[s] = (int)(own >> 32);
  }
}

// candidates ran out: remaining slots get (-1,-1,KLT_NOT_FOUND) (:131-153)
__global__ void k_fill_rest(const unsigned *__restrict__ count, unsigned cap,
                            const int *__restrict__ slots, const int *__restrict__ nslots,
                            float *__restrict__ fx, float *__restrict__ fy,
                            int *__restrict__ fval) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (int)min(*count, cap);
  if (k >= n && k < *nslots) {
    const int s = slots[k];
    fx[s] = -1.0f;
    fy[s] = -1.0f;
    fval[s] = GKLT_NOT_FOUND;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
struct SelScratch {
  int w = 0, h = 0, cap_list = 0, cap_slots = 0;
  float *sum[3] = {nullptr, nullptr, nullptr};  // (h+1) x (w+1)
  unsigned *state = nullptr;
  unsigned char *blocked = nullptr;
  unsigned long long *list = nullptr;
  unsigned *counters = nullptr;  // [0] accepted count, [1..] per-round undecided
  int *slots = nullptr, *nslots = nullptr;
  unsigned *h_counters = nullptr;  // pinned
};
constexpr int kMaxRounds = 4096;
constexpr int kRoundBatch = 4;

void sel_free(SelScratch *s) {
  if (!s) return;
  for (auto *p : s->sum) (void)hipFree(p);
  (void)hipFree(s->state);
  (void)hipFree(s->blocked);
  (void)hipFree(s->list);
  (void)hipFree(s->counters);
  (void)hipFree(s->slots);
  (void)hipFree(s->nslots);
  if (s->h_counters) (void)hipHostFree(s->h_counters);
  delete s;
}

static int sel_get(gklt_ctx *ctx, int w, int h, int nslots, SelScratch **out) {
  SelScratch *s = ctx->sel;
  if (s && (s->w != w || s->h != h || s->cap_slots < nslots)) {
    GKLT_HIP(hipDeviceSynchronize());
    sel_free(s);
    s = ctx->sel = nullptr;
  }
  if (!s) {
    s = new SelScratch();
    ctx->sel = s;
    s->w = w;
    s->h = h;
    s->cap_list = w * h;  // mindist 1 accepts every candidate
    s->cap_slots = std::max(nslots, 4096);
    const size_t sb = sizeof(float) * (size_t)(w + 1) * (h + 1);
    for (auto &p : s->sum) GKLT_HIP(hipMalloc(&p, sb));
    GKLT_HIP(hipMalloc(&s->state, sizeof(unsigned) * (size_t)w * h));
    GKLT_HIP(hipMalloc(&s->blocked, (size_t)w * h));
    GKLT_HIP(hipMalloc(&s->list, sizeof(unsigned long long) * (size_t)s->cap_list));
    GKLT_HIP(hipMalloc(&s->counters, sizeof(unsigned) * (kMaxRounds + 1)));
    GKLT_HIP(hipMalloc(&s->slots, sizeof(int) * (size_t)s->cap_slots));
    GKLT_HIP(hipMalloc(&s->nslots, sizeof(int)));
    GKLT_HIP(hipHostMalloc(&s->h_counters, sizeof(unsigned) * (kMaxRounds + 1), hipHostMallocDefault));
  }
  *out = s;
  return 0;
}

int select_integrals_from_pyramid(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *py,
                                  SelScratch *sc) {
  const int w = py->lv[0].w, h = py->lv[0].h;
  {
    ProfScope ps(ctx, s, GKLT_K_INTEGRAL);
    hipLaunchKernelGGL(k_integral_rows, dim3((h + 63) / 64), dim3(64), 0, s, py->lv[0].tex, w, h,
                       sc->sum[0], sc->sum[1], sc->sum[2]);
    hipLaunchKernelGGL(k_integral_cols, dim3((w + 1 + 255) / 256, 3), dim3(256), 0, s, sc->sum[0],
                       sc->sum[1], sc->sum[2], w, h);
  }
  GKLT_HIP(hipGetLastError());
  return 0;
}

// S1-S4 on device-resident sums and a device-resident list (x, y, val of n features).
int select_on_device(gklt_ctx *ctx, hipStream_t s, SelScratch *sc, int mode, int round2, int n,
                     float *fx, float *fy, int *fval) {
  const gklt_params &P = ctx->prm;
  const int w = sc->w, h = sc->h;
  const int overwrite = mode == GKLT_SELECT_ALL;
  const int hw = P.window_width / 2, hh = P.window_height / 2;
  const int bx = std::max(P.borderx, hw), by = std::max(P.bordery, hh);  // :308-314
  const int step = P.nSkippedPixels + 1;
  const int r = round2 ? P.mindist : P.mindist - 1;  // :111 vs :525
  if (step < 1 || P.mindist < 0) {
    set_error("nSkippedPixels %d / mindist %d invalid", P.nSkippedPixels, P.mindist);
    return GKLT_EINVAL;
  }
  if (w > 65535 || h > 65535) {
    set_error("selector supports images up to 65535 px per side");
    return GKLT_EINVAL;
  }
  GKLT_HIP(hipMemsetAsync(sc->counters, 0, sizeof(unsigned) * (kMaxRounds + 1), s));
  if (!overwrite) {
    GKLT_HIP(hipMemsetAsync(sc->blocked, 0, (size_t)w * h, s));
    if (n > 0)
      hipLaunchKernelGGL(k_block_squares, dim3(n), dim3(256), 0, s, fx, fy, fval, n, w, h,
                         P.mindist, sc->blocked);
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    hipLaunchKernelGGL(k_score, dim3((w + 255) / 256, h), dim3(256), 0, s, sc->sum[0], sc->sum[1],
                       sc->sum[2], overwrite ? (const unsigned char *)nullptr : sc->blocked, w, h,
                       hw, hh, bx, by, step, P.min_eigenvalue, round2, sc->state);
  }
  hipLaunchKernelGGL(k_lost_slots, dim3(1), dim3(1024), 0, s, fval, n, overwrite, sc->slots,
                     sc->nslots);
  GKLT_HIP(hipGetLastError());

  // fixpoint rounds (r < 0 happens for mindist 0: every candidate is accepted in one round)
  const int rr = std::max(r, 0);
  const size_t lds = (size_t)8 * ((MW + 2 * rr) * (MH + 2 * rr) + (MH + 2 * rr) * MW) +
                     (size_t)(MW + 2 * rr) * (MH + 2 * rr) + (size_t)(MH + 2 * rr) * MW;
  if (lds > 160 * 1024) {
    set_error("mindist %d needs %zu bytes of LDS per tile (> 160 KiB)", P.mindist, lds);
    return GKLT_EINVAL;
  }
  if (lds > 48 * 1024)
    GKLT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mis_round),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((w + MW - 1) / MW, (h + MH - 1) / MH);
  int round = 1;
  bool done = false;
  while (!done) {
    const int first = round;
    {
      ProfScope ps(ctx, s, GKLT_K_SELECT);
      for (int b = 0; b < kRoundBatch && round <= kMaxRounds; ++b, ++round)
        hipLaunchKernelGGL(k_mis_round, grid, dim3(MT), lds, s, sc->state, w, h, rr, sc->counters,
                           round);
    }
    GKLT_HIP(hipGetLastError());
    GKLT_HIP(hipMemcpyAsync(sc->h_counters + first, sc->counters + first,
                            sizeof(unsigned) * (round - first), hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipStreamSynchronize(s));
    for (int k = first; k < round; ++k)
      if (sc->h_counters[k] == 0) done = true;
    if (!done && round > kMaxRounds) {
      set_error("min-distance suppression did not converge in %d rounds", kMaxRounds);
      return GKLT_EHIP;
    }
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    hipLaunchKernelGGL(k_collect, dim3((w + 255) / 256, h), dim3(256), 0, s, sc->state, w, h,
                       sc->list, sc->counters, (unsigned)sc->cap_list);
  }
  GKLT_HIP(hipMemcpyAsync(sc->h_counters, sc->counters, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  const unsigned naccepted = std::min(sc->h_counters[0], (unsigned)sc->cap_list);
  if (sc->h_counters[0] > (unsigned)sc->cap_list) {
    set_error("accepted-feature list overflow (%u > %d)", sc->h_counters[0], sc->cap_list);
    return GKLT_ECAPACITY;
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    if (naccepted > 0)
      hipLaunchKernelGGL(k_rank_assign, dim3((naccepted + 255) / 256), dim3(256), 0, s, sc->list,
                         sc->counters, (unsigned)sc->cap_list, sc->slots, sc->nslots, fx, fy, fval);
    if (n > 0)
      hipLaunchKernelGGL(k_fill_rest, dim3((n + 255) / 256), dim3(256), 0, s, sc->counters,
                         (unsigned)sc->cap_list, sc->slots, sc->nslots, fx, fy, fval);
  }
  GKLT_HIP(hipGetLastError());
  return 0;
}

}  // namespace gklt

using namespace gklt;

// ============================================================================ C ABI
extern "C" int gklt_integral_products(gklt_ctx *ctx, gklt_pyramid *py, float *sumxx, float *sumyy,
                                      float *sumxy) {
  if (!ctx || !py || py->ctx != ctx || !sumxx || !sumyy || !sumxy) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (!py->has_ready) {
    set_error("integral images of a pyramid that was never built");
    return GKLT_EINVAL;
  }
  const int w = py->lv[0].w, h = py->lv[0].h;
  SelScratch *sc;
  if (int rc = sel_get(ctx, w, h, 0, &sc)) return rc;
  hipStream_t s = ctx->s_trk;
  GKLT_HIP(hipStreamWaitEvent(s, py->ready, 0));
  if (int rc = select_integrals_from_pyramid(ctx, s, py, sc)) return rc;
  const size_t sb = sizeof(float) * (size_t)(w + 1) * (h + 1);
  float *dst[3] = {sumxx, sumyy, sumxy};
  for (int i = 0; i < 3; ++i)
    GKLT_HIP(hipMemcpyAsync(dst[i], sc->sum[i], sb, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

// wrapper semantics of selectGoodFeatures.hpp:424-449, 997-1022, 1341-1355
static int resolve_variant(int mode, int variant, int *round2) {
  if (mode != GKLT_SELECT_ALL && mode != GKLT_REPLACE_SOME) return GKLT_EINVAL;
  if (variant == GKLT_VARIANT_UNIFORM) {
    if (mode == GKLT_SELECT_ALL) {
      set_error("KLTSelectGoodFeaturesUniform is not built on the GPU yet (SURVEY 8a row S5)");
      return GKLT_EUNSUPPORTED;
    }
    variant = GKLT_VARIANT_PLAIN;  // KLTReplaceLostFeaturesUniform runs the plain replace (:1353)
  }
  if (variant != GKLT_VARIANT_PLAIN && variant != GKLT_VARIANT_ROUND2) return GKLT_EINVAL;
  // KLTSelectGoodFeatures2 runs the plain body (:1005); only KLTReplaceLostFeatures2 rounds (:1020)
  *round2 = variant == GKLT_VARIANT_ROUND2 && mode == GKLT_REPLACE_SOME;
  return 0;
}

extern "C" int gklt_select(gklt_ctx *ctx, int mode, int variant, const float *sumxx,
                           const float *sumyy, const float *sumxy, int width, int height,
                           int *nFeatures, int capacity, float *x, float *y, int *val, int gridW,
                           int gridH) {
  (void)gridW;
  (void)gridH;
  if (!ctx || !sumxx || !sumyy || !sumxy || !nFeatures || width < 1 || height < 1) return GKLT_EINVAL;
  const int n = *nFeatures;
  if (n < 0 || n > capacity || (n > 0 && (!x || !y || !val))) return GKLT_EINVAL;
  int round2 = 0;
  if (int rc = resolve_variant(mode, variant, &round2)) return rc;
  if (n == 0) return 0;
  if (mode == GKLT_REPLACE_SOME && gklt_count_remaining(n, val) == n) return 0;  // :442-447
  GKLT_HIP(hipSetDevice(ctx->device));
  SelScratch *sc;
  if (int rc = sel_get(ctx, width, height, n, &sc)) return rc;
  if (!ctx->scratch_list || ctx->scratch_list->capacity < n) {
    if (ctx->scratch_list) gklt_flist_destroy(ctx->scratch_list);
    ctx->scratch_list = nullptr;
    if (int rc = gklt_flist_create(ctx, std::max(n, 4096), &ctx->scratch_list)) return rc;
  }
  gklt_flist *fl = ctx->scratch_list;
  hipStream_t s = ctx->s_trk;
  const size_t sb = sizeof(float) * (size_t)(width + 1) * (height + 1);
  const float *src[3] = {sumxx, sumyy, sumxy};
  for (int i = 0; i < 3; ++i)
    GKLT_HIP(hipMemcpyAsync(sc->sum[i], src[i], sb, hipMemcpyHostToDevice, s));
  if (int rc = gklt_flist_upload(ctx, fl, n, x, y, val)) return rc;
  if (int rc = select_on_device(ctx, s, sc, mode, round2, n, fl->x, fl->y, fl->val)) return rc;
  return gklt_flist_download(ctx, fl, x, y, val);
}

extern "C" int gklt_select_from_pyramid(gklt_ctx *ctx, gklt_pyramid *py, int mode, int variant,
                                        gklt_flist *fl) {
  if (!ctx || !py || !fl || py->ctx != ctx || fl->ctx != ctx) return GKLT_EINVAL;
  int round2 = 0;
  if (int rc = resolve_variant(mode, variant, &round2)) return rc;
  if (fl->n == 0) return 0;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (!py->has_ready) {
    set_error("selecting on a pyramid that was never built");
    return GKLT_EINVAL;
  }
  SelScratch *sc;
  if (int rc = sel_get(ctx, py->lv[0].w, py->lv[0].h, fl->n, &sc)) return rc;
  hipStream_t s = ctx->s_trk;
  GKLT_HIP(hipStreamWaitEvent(s, py->ready, 0));
  if (int rc = select_integrals_from_pyramid(ctx, s, py, sc)) return rc;
  // replacing with nothing lost is a no-op (:442-447): k_lost_slots yields 0 slots and the
  // fixpoint result is simply not written anywhere, so no host round trip is needed for it
  if (int rc = select_on_device(ctx, s, sc, mode, round2, fl->n, fl->x, fl->y, fl->val)) return rc;
  GKLT_HIP(hipEventRecord(py->consumed, s));
  py->has_consumed = true;
  return 0;
}
