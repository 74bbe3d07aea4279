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
#include <cstdlib>
#include <functional>
#include <utility>
#include <vector>

#include "gklt_internal.h"

namespace gklt {

namespace {

constexpr unsigned kAccepted = 0x80000000u;

// ------------------------------------------------------------------ S6 integral images
// The running prefix of a row (and then of a column) is one serial fp32 chain: cv::integral's
// order has to be kept because the scores are differences of sums ~1e9.  The parallelism is
// rows x 3 quantities (then columns x 3), so the kernels are built around keeping those chains
// fed: the loads of the next tile are in flight while the chains of the current one run.
//
// Rows: a block of 256 threads owns kIR = 16 rows (68 blocks at 1080p instead of 17 with 64 rows).
// Per 256-column tile every thread loads one column of the 16 rows (16 loads in flight, issued
// one tile ahead), the products go to LDS (row stride 257: conflict-free both ways), 48 lanes of
// wave 0 -- one per (quantity, row) -- extend their running prefixes over the 256 columns, and all
// threads write the prefixes back row-contiguously.
constexpr int kIR = 16, kIC = 256;
__global__ __launch_bounds__(256) void k_integral_rows(const Texel *__restrict__ tex, int w, int h,
                                                       float *__restrict__ sxx,
                                                       float *__restrict__ syy,
                                                       float *__restrict__ sxy) {
  __shared__ float t[3][kIR][kIC + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int y0 = blockIdx.x * kIR, w1 = w + 1;
  float run = 0.f;  // wave 0, lane < 48: running prefix of quantity lane / 16, row y0 + lane % 16
  float gx[kIR], gy[kIR];
  auto load = [&](int x0) {
    const int col = min(x0 + tid, w - 1);
#pragma unroll
    for (int r = 0; r < kIR; ++r) {  // clamped, unconditional: all 16 loads are in flight together
      const Texel v = tex[(size_t)min(y0 + r, h - 1) * w + col];
      gx[r] = v.y;
      gy[r] = v.z;
    }
  };
  load(0);
  for (int x0 = 0; x0 < w; x0 += kIC) {
    const int x = x0 + tid;
#pragma unroll
    for (int r = 0; r < kIR; ++r) {
      const bool in = x < w && y0 + r < h;
      const float a = in ? gx[r] : 0.f, b = in ? gy[r] : 0.f;
      t[0][r][tid] = a * a;  // cv::multiply, System.cpp:386-388
      t[1][r][tid] = b * b;
      t[2][r][tid] = a * b;
    }
    __syncthreads();
    if (x0 + kIC < w) load(x0 + kIC);
    if (wave == 0 && lane < 3 * kIR) {
      float *p = t[lane / kIR][lane % kIR];
      const int nc = min(kIC, w - x0);
#pragma unroll 16
      for (int c = 0; c < nc; ++c) {
        run = run + p[c];
        p[c] = run;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kIR; ++r) {
      if (x < w && y0 + r < h) {
        const size_t o = (size_t)(y0 + r + 1) * w1 + x + 1;
        sxx[o] = t[0][r][tid];
        syy[o] = t[1][r][tid];
        sxy[o] = t[2][r][tid];
      }
    }
    __syncthreads();
  }
}

// sum(y+1, x+1) = sum(y, x+1) + prefix_y(x), in place; row 0 and column 0 are zero.  A column is
// one chain of h adds; a wavefront on its own cannot keep enough loads in flight to feed it (1080
// loads at <= 63 outstanding and ~2 us each), so the loads are cooperative: a block of 4 waves owns
// 64 columns of one quantity, every wave loads 16 rows of the next 64-row tiles (two tiles ahead,
// in registers), the tile goes through LDS, wave 0 -- lane = column -- extends the 64 chains, and
// all waves write the tile back.
constexpr int kCT = 64;  // rows per tile
__global__ __launch_bounds__(256) void k_integral_cols(float *__restrict__ sxx,
                                                       float *__restrict__ syy,
                                                       float *__restrict__ sxy, int w, int h) {
  __shared__ float t[kCT][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = blockIdx.x * 64 + lane;  // column of the (h+1) x (w+1) image
  const int w1 = w + 1, xc = min(x, w);
  float *s = blockIdx.y == 0 ? sxx : (blockIdx.y == 1 ? syy : sxy);
  if (wave == 0 && x <= w) s[x] = 0.f;  // row 0
  const int ntiles = (h + kCT - 1) / kCT;
  float acc = 0.f;  // wave 0: running column sum
  auto load = [&](float (&v)[16], int k) {
    const int kc = min(k, ntiles - 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // clamped, unconditional: 16 loads in flight per register set
      const int y = min(1 + kc * kCT + 16 * wave + i, h);
      v[i] = s[(size_t)y * w1 + xc];
    }
  };
  auto process = [&](float (&v)[16], int k) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[16 * wave + i][lane] = x == 0 ? 0.f : v[i];  // column 0 is zero
    __syncthreads();
    if (k + 2 < ntiles) load(v, k + 2);
    if (wave == 0) {
      const int nr = min(kCT, h - k * kCT);
#pragma unroll 16
      for (int r = 0; r < nr; ++r) {
        acc = acc + t[r][lane];
        t[r][lane] = acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int y = 1 + k * kCT + 16 * wave + i;
      if (y <= h && x <= w) s[(size_t)y * w1 + x] = t[16 * wave + i][lane];
    }
    __syncthreads();
  };
  float a[16], b[16];
  load(a, 0);
  load(b, 1);
  for (int k = 0; k < ntiles; k += 2) {
    process(a, k);
    if (k + 1 < ntiles) process(b, k + 1);
  }
}

// ------------------------------------------------------------------ S1 score
// candidate score from the three window sums: _minEigenvalue (:220-223), threshold and int
// conversion of the plain (:351-361) and rounding (:737-740) flavours; 0 = not a candidate
__device__ __forceinline__ unsigned score_from_sums(float gxx, float gyy, float gxy, int min_eig,
                                                    int round2, int score_max) {
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
  // score_max: INT_MAX, or INT_MAX - 1 for the Uniform selector, where INT_MAX marks a surviving
  // feature (:1132,1232) and the oracle documents the clamp one below it
  return (iv >= 1 && iv >= min_eig) ? (unsigned)min(iv, score_max) : 0u;
}

// state[y*w+x] = score (1..INT_MAX) of an emitted candidate, else 0.
// selectGoodFeatures.hpp:324-366 (plain), :707-751 (rounding flavour).
__global__ void k_score(const float *__restrict__ sxx, const float *__restrict__ syy,
                        const float *__restrict__ sxy, const unsigned char *__restrict__ blocked,
                        int w, int h, int hw, int hh, int bx, int by, int step, int min_eig,
                        int round2, int score_max, unsigned *__restrict__ state) {
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
    out = score_from_sums(gxx, gyy, gxy, min_eig, round2, score_max);
  }
  state[(size_t)y * w + x] = out;
}

// ---- S1 without integral images (SURVEY section 7's fast path; no counterpart in the reference,
// whose selector API takes the integral images): the window sums of gx^2, gy^2, gx gy formed
// directly from the level's texels by a separable box sum -- each window row left to right, the row
// sums top to bottom (oracle: ok_box_products) -- so there is no cancellation between sums ~1e9 and
// no serial prefix chain.  Scores differ from the integral-image ones in the last bits; selections
// are compared by set overlap, and bit for bit with the oracle's direct form.
constexpr int kDW = 64, kDH = 16, kDMaxHalf = 16;
__global__ __launch_bounds__(256) void k_score_direct(const Texel *__restrict__ tex,
                                                      const unsigned char *__restrict__ blocked, int w,
                                                      int h, int hw, int hh, int bx, int by, int step,
                                                      int min_eig, int round2, int score_max,
                                                      unsigned *__restrict__ state) {
  extern __shared__ __align__(16) float ldsf[];
  const int inW = kDW + 2 * hw, inH = kDH + 2 * hh, PW = inW | 1;
  float *P = ldsf;                 // [3][inH][PW] products
  float *Rs = P + 3 * inH * PW;    // [3][inH][kDW + 1] row sums
  const int x0 = blockIdx.x * kDW, y0 = blockIdx.y * kDH;
  for (int i = threadIdx.x; i < inW * inH; i += 256) {
    const int ty = i / inW, tx = i - ty * inW;
    const int x = min(max(x0 - hw + tx, 0), w - 1), y = min(max(y0 - hh + ty, 0), h - 1);
    const Texel t = tex[(size_t)y * w + x];
    P[(0 * inH + ty) * PW + tx] = t.y * t.y;  // cv::multiply, System.cpp:386-388
    P[(1 * inH + ty) * PW + tx] = t.z * t.z;
    P[(2 * inH + ty) * PW + tx] = t.y * t.z;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * inH * kDW; i += 256) {
    const int q = i / (inH * kDW), rem = i - q * inH * kDW, ty = rem / kDW, tx = rem - ty * kDW;
    const float *p = P + (q * inH + ty) * PW + tx;
    float acc = p[0];
    for (int d = 1; d <= 2 * hw; ++d) acc = acc + p[d];
    Rs[(q * inH + ty) * (kDW + 1) + tx] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDW * kDH; i += 256) {
    const int ty = i / kDW, tx = i - ty * kDW;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= w || y >= h) continue;
    unsigned out = 0;
    const bool on_grid = x >= bx && x < w - bx && y >= by && y < h - by &&
                         (x - bx) % step == 0 && (y - by) % step == 0;
    if (on_grid && !(blocked && blocked[(size_t)y * w + x])) {
      float sum[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float *p = Rs + (q * inH + ty) * (kDW + 1) + tx;
        float acc = p[0];
        for (int d = 1; d <= 2 * hh; ++d) acc = acc + p[d * (kDW + 1)];
        sum[q] = acc;
      }
      out = score_from_sums(sum[0], sum[1], sum[2], min_eig, round2, score_max);
    }
    state[(size_t)y * w + x] = out;
  }
}
inline size_t score_direct_lds(int hw, int hh) {
  const size_t inW = kDW + 2 * hw, inH = kDH + 2 * hh;
  return sizeof(float) * 3 * inH * ((inW | 1) + kDW + 1);
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

// LDS layout of a round: key[inH][S] (u64), rmax[inH][MW+1] (u64), acc[inH][S] (u8),
// racc[inH][MW+1] (u8), with S = inW + 1.  The odd row strides make accesses with lane <-> row
// conflict-free (row pass) as well as those with lane <-> column (column pass).
__host__ __device__ inline int mis_stride(int r) { return MW + 2 * r + 1; }
inline size_t mis_lds_bytes(int r) {
  const size_t inH = MH + 2 * r, S = mis_stride(r);
  return 8 * (inH * S + inH * (MW + 1)) + inH * S + inH * (MW + 1);
}

// m[i] / o[i] = max of k / OR of a over the window [i, i + 2r] of a strided sequence, for G
// consecutive i.  All G windows share the middle [G-1, 2r]; each adds a suffix of the first G-1
// and a prefix of the last G-1 elements: 2r + G reads for G outputs instead of G * (2r + 1).
template <int G>
__device__ __forceinline__ void window_group(const unsigned long long *k, const unsigned char *a,
                                             int sk, int sa, int r, unsigned long long (&m)[G],
                                             unsigned (&o)[G]) {
  const int L2 = 2 * r;
  if (L2 >= G - 1) {
    unsigned long long cm = 0;
    unsigned co = 0;
#pragma unroll
    for (int j = G - 1; j <= L2; ++j) {
      cm = max(cm, k[j * sk]);
      co |= a[j * sa];
    }
    m[G - 1] = cm;
    o[G - 1] = co;
    unsigned long long sm = 0;
    unsigned so = 0;
#pragma unroll
    for (int i = G - 2; i >= 0; --i) {
      sm = max(sm, k[i * sk]);
      so |= a[i * sa];
      m[i] = max(cm, sm);
      o[i] = co | so;
    }
    unsigned long long pm = 0;
    unsigned po = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) {
      pm = max(pm, k[(L2 + i) * sk]);
      po |= a[(L2 + i) * sa];
      m[i] = max(m[i], pm);
      o[i] |= po;
    }
  } else {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      unsigned long long mm = 0;
      unsigned oo = 0;
      for (int j = 0; j <= L2; ++j) {
        mm = max(mm, k[(i + j) * sk]);
        oo |= a[(i + j) * sa];
      }
      m[i] = mm;
      o[i] = oo;
    }
  }
}

// Fixpoint rounds on the 64x16 tile of this workgroup, shared by the plain and the Uniform
// selector.  keyf(score, x, y) is the priority key of an undecided candidate; decide(gi, x, y,
// own, covered, is_max) applies the rule to one undecided pixel and returns kStays, kRejected or
// kAcceptedNow.  The haloed tile (keys of undecided pixels, flags of accepted ones) is staged in
// LDS, followed by a separable (2r+1)-window max of the keys / OR of the flags.
//
// One launch iterates the rule on the tile up to `max_local` times, with the halo as it was when
// the tile was staged.  Decisions are final and never depend on information that can be
// withdrawn (an accepted neighbour stays accepted; an undecided neighbour of higher priority only
// makes the pixel wait), so a stale halo delays decisions but cannot change them: dependency
// chains that stay inside a tile cost LDS iterations instead of launches.
// `gate`: the tile is skipped when neither it nor one of its 8 neighbours decided anything in the
// previous launch (prev_changed) -- it has already drawn every conclusion from that halo.
constexpr int kStays = 1, kRejected = 0, kAcceptedNow = 2;
// Staging loads: the addresses are clamped so that the loads can be unconditional and in flight
// together, but left alone the compiler sinks every load into the in-image test that uses its
// value and waits for it there -- one memory round trip per entry (seen in the ISA, round 3: 12
// serial round trips per tile and launch).  Naming the whole batch as operands of one empty asm
// statement makes all of them complete before the first use.
#define GKLT_ISSUE_TOGETHER_12(v)                                                                  \
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),        \
                    "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]))
// R >= 0: the radius as a compile-time constant (window loops unroll, their LDS reads are issued
// together, staging divides by a constant); R < 0: radius r_rt.
template <int R, typename KeyF, typename DecideF>
__device__ __forceinline__ void mis_round_tile(const unsigned *__restrict__ state, int w, int h,
                                               int r_rt, unsigned *__restrict__ counter,
                                               unsigned *__restrict__ tile_left,
                                               const unsigned char *__restrict__ prev_changed,
                                               unsigned char *__restrict__ cur_changed, int gate,
                                               int max_local, KeyF keyf, DecideF decide) {
  extern __shared__ __align__(16) unsigned long long lds64[];
  const int r = R >= 0 ? R : r_rt;
  const int inW = MW + 2 * r, inH = MH + 2 * r, S = mis_stride(r), SR = MW + 1;
  unsigned long long *key = lds64;          // inH x S  : key of undecided, 0 otherwise
  unsigned long long *rmax = key + inH * S; // inH x SR : row-window max
  unsigned char *acc = reinterpret_cast<unsigned char *>(rmax + inH * SR);  // inH x S accepted
  unsigned char *racc = acc + inH * S;                                      // inH x SR
  __shared__ int any_undecided;
  __shared__ unsigned blk_left[2];
  __shared__ int blk_changed[2];
  const int x0 = blockIdx.x * MW, y0 = blockIdx.y * MH;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  // tile_left = undecided core pixels after the tile's last launch (all ones before the first):
  // a tile that has none left never gets any again and is skipped without staging
  unsigned *const my_left = tile_left + tile;
  const unsigned left_before = *my_left;
  if (left_before == 0) {
    if (threadIdx.x == 0) cur_changed[tile] = 0;
    return;
  }
  if (gate) {
    int any = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int bx = (int)blockIdx.x + dx, by = (int)blockIdx.y + dy;
        if (bx >= 0 && bx < (int)gridDim.x && by >= 0 && by < (int)gridDim.y)
          any |= prev_changed[by * gridDim.x + bx];
      }
    if (!any) {  // uniform over the workgroup
      if (threadIdx.x == 0) {
        cur_changed[tile] = 0;
        atomicAdd(counter, left_before);
      }
      return;
    }
  }
  if (threadIdx.x == 0) {
    any_undecided = 0;
    blk_left[0] = blk_left[1] = 0;
    blk_changed[0] = blk_changed[1] = 0;
  }
  __syncthreads();
  // staging in batches of kStage unconditional (clamped) loads per thread, so that a batch is in
  // flight together instead of one load per round trip
  int mine = 0;
  constexpr int kStage = 12;
  const int total = inW * inH;
  for (int i0 = threadIdx.x; i0 < total; i0 += kStage * MT) {
    unsigned sv[kStage];
    int txv[kStage], tyv[kStage];
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int i = min(i0 + u * MT, total - 1);
      tyv[u] = i / inW;
      txv[u] = i - tyv[u] * inW;
      const int x = x0 - r + txv[u], y = y0 - r + tyv[u];
      sv[u] = state[(size_t)min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1)];
    }
    GKLT_ISSUE_TOGETHER_12(sv);
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      if (i0 + u * MT < total) {
        const int tx = txv[u], ty = tyv[u];
        const int x = x0 - r + tx, y = y0 - r + ty;
        const unsigned s = (x >= 0 && x < w && y >= 0 && y < h) ? sv[u] : 0u;
        const bool a = (s & kAccepted) != 0;
        key[ty * S + tx] = (s != 0 && !a) ? keyf(s, x0 - r + tx, y0 - r + ty) : 0ull;
        acc[ty * S + tx] = a;
        if (s != 0 && !a && tx >= r && tx < r + MW && ty >= r && ty < r + MH) mine = 1;
      }
    }
  }
  if (mine) any_undecided = 1;
  __syncthreads();
  if (!any_undecided) {  // nothing to decide in this tile
    if (threadIdx.x == 0) {
      *my_left = 0;
      cur_changed[tile] = 0;
    }
    return;
  }
  constexpr int GR = 8, GC = 4;
  int changed_ever = 0;
  unsigned left_now = 0;
  for (int it = 0; it < max_local; ++it) {
    const int pp = it & 1;
    // row pass: lane <-> row, 8 adjacent outputs per task
    for (int t = threadIdx.x; t < inH * (MW / GR); t += MT) {
      const int ty = t % inH, gq = t / inH;
      unsigned long long m[GR];
      unsigned o[GR];
      window_group<GR>(key + ty * S + GR * gq, acc + ty * S + GR * gq, 1, 1, r, m, o);
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        rmax[ty * SR + GR * gq + i] = m[i];
        racc[ty * SR + GR * gq + i] = (unsigned char)o[i];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // the other parity's slots: last read before this barrier, next
      blk_left[pp ^ 1] = 0;  // written after the one that ends this iteration
      blk_changed[pp ^ 1] = 0;
    }
    // column pass: lane <-> column, 4 outputs (rows) per task
    int left = 0, changed = 0;
    for (int t = threadIdx.x; t < MW * (MH / GC); t += MT) {
      const int tx = t % MW, gq = t / MW;
      unsigned long long m[GC];
      unsigned o[GC];
      window_group<GC>(rmax + GC * gq * SR + tx, racc + GC * gq * SR + tx, SR, SR, r, m, o);
#pragma unroll
      for (int i = 0; i < GC; ++i) {
        const int ty = GC * gq + i;
        const int li = (ty + r) * S + tx + r;
        const unsigned long long own = key[li];
        if (own == 0) continue;
        const int d = decide((size_t)(y0 + ty) * w + x0 + tx, x0 + tx, y0 + ty, own, o[i] != 0, m[i] == own);
        if (d == kStays) {
          ++left;
        } else {  // only this thread reads or writes this pixel's own entry during the pass
          key[li] = 0ull;
          if (d == kAcceptedNow) acc[li] = 1;
          changed = 1;
        }
      }
    }
    if (left) atomicAdd(&blk_left[pp], (unsigned)left);
    if (changed) blk_changed[pp] = 1;
    __syncthreads();
    left_now = blk_left[pp];
    const int ch = blk_changed[pp];
    changed_ever |= ch;
    if (!ch || left_now == 0) break;
  }
  if (threadIdx.x == 0) {
    *my_left = left_now;
    cur_changed[tile] = (unsigned char)changed_ever;
    if (left_now) atomicAdd(counter, left_now);
  }
}

// cell grid of the Uniform selector as k_mis32_round<R, true> takes it (see there)
constexpr int kU32Cells = 32, kU32ScoreBits = 26;
struct U32Grid {
  int gw, gh, cw, ch;
  const unsigned short *rank;
  unsigned *need64;
};

// start of a phase of rounds, one launch instead of up to four fills: every tile live and "changed",
// optionally the round counters and the need64 flag cleared
__global__ void k_rounds_reset(unsigned *__restrict__ tile_left, int ntile_words,
                               unsigned *__restrict__ tile_changed_words, int nchanged_words,
                               unsigned *__restrict__ counters, int ncounters,
                               unsigned *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntile_words) tile_left[i] = 0xFFFFFFFFu;
  if (i < nchanged_words) tile_changed_words[i] = 0x01010101u;
  if (i < ncounters) counters[i] = 0u;
  if (i == 0 && flag) *flag = 0u;
}

// counters[round] = undecided candidates left after this round (summed over tiles)
template <int R>
__global__ __launch_bounds__(MT) void k_mis_round(unsigned *__restrict__ state, int w, int h,
                                                  int r, unsigned *__restrict__ counters,
                                                  unsigned *__restrict__ tile_left,
                                                  const unsigned char *__restrict__ prev_changed,
                                                  unsigned char *__restrict__ cur_changed,
                                                  int gate, int max_local, int round, U32Grid) {
  mis_round_tile<R>(
      state, w, h, r, counters + round, tile_left, prev_changed, cur_changed, gate, max_local,
      [](unsigned s, int x, int y) { return prio_key(s, x, y); },
      [&](size_t gi, int, int, unsigned long long own, bool covered, bool is_max) -> int {
        if (covered) {  // an accepted neighbour covers it
          state[gi] = 0;
          return kRejected;
        }
        if (is_max) {
          state[gi] = (unsigned)(own >> 32) | kAccepted;
          return kAcceptedNow;
        }
        return kStays;
      });
}

// ---- the plain selector's rounds on 32-bit keys
// The plain priority (score desc, y asc, x asc) needs no position bits in LDS: a tile keeps
//   0            nothing to decide (no candidate, or rejected)
//   score << 1   undecided candidate (score 1..INT_MAX)
//   0xFFFFFFFF   accepted
// and reduces windows with tie_max(a, b) = max(a, b) | (a == b): bit 0 of a window maximum says
// that two or more entries share it.  A pixel whose score is the unique window maximum is the
// highest-priority undecided one; a window maximum of 0xFFFFFFFF means an accepted neighbour.
// Only when the top score is shared does the pixel compare positions, by scanning its window
// (exact_window).  Two undecided INT_MAX scores would combine to 0xFFFFFFFF; a tile that stages
// such a score decides all its pixels by scanning.  8 bytes of LDS per pixel instead of 18: all
// tiles of a 1080p image are resident at once (8 workgroups per CU).
inline size_t mis32_lds_bytes(int r) {
  const size_t inH = MH + 2 * r;
  return 4 * inH * (size_t)(mis_stride(r) + MW + 1);
}
__device__ __forceinline__ unsigned tie_max(unsigned a, unsigned b) {
  return max(a, b) | (unsigned)(a == b);
}
// m[i] = tie_max over the window [i, i + 2r] of a strided sequence, for G consecutive i (see
// window_group; the three parts of a window are disjoint, so no entry meets itself)
template <int G>
__device__ __forceinline__ void window_group32(const unsigned *k, int sk, int r, unsigned (&m)[G]) {
  const int L2 = 2 * r;
  if (L2 >= G - 1) {
    unsigned cm = k[(G - 1) * sk];
#pragma unroll
    for (int j = G; j <= L2; ++j) cm = tie_max(cm, k[j * sk]);
    m[G - 1] = cm;
    unsigned sm = 0;
#pragma unroll
    for (int i = G - 2; i >= 0; --i) {
      sm = i == G - 2 ? k[i * sk] : tie_max(sm, k[i * sk]);
      m[i] = tie_max(cm, sm);
    }
    unsigned pm = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) {
      pm = i == 1 ? k[(L2 + 1) * sk] : tie_max(pm, k[(L2 + i) * sk]);
      m[i] = tie_max(m[i], pm);
    }
  } else {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      unsigned mm = k[i * sk];
      for (int j = 1; j <= L2; ++j) mm = tie_max(mm, k[(i + j) * sk]);
      m[i] = mm;
    }
  }
}
// the rule for the undecided pixel at LDS index li by comparing it with every entry of its window
__device__ __forceinline__ int exact_window(const unsigned *key, int li, int S, int r, unsigned own) {
  bool covered = false, higher = false;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      const unsigned e = key[li + dy * S + dx];
      covered |= e == 0xFFFFFFFFu;
      higher |= e > own || (e == own && (dy < 0 || (dy == 0 && dx < 0)));
    }
  return covered ? kRejected : (higher ? kStays : kAcceptedNow);
}

// UNI: the uncapped phase of the Uniform selector on the same 32-bit keys.  Its priority is (visiting
// rank of the cell asc, score desc, y asc, x asc) -- 63 bits in k_u_round's keys -- but a tile only
// ever compares candidates of the few cells that meet its haloed region (at most kU32Cells), so the
// cell part shrinks to the DENSE rank among those, 5 bits, and the score keeps 26:
//   ((31 - dense rank) << 26 | score) << 1        undecided candidate, bit 0 = tie flag as above
// Equal keys are then equal scores in one cell, which raster order decides as in the plain
// selector.  A score of 2^26 or more does not fit: the tile raises *need64 and the host repeats the
// phase with the 64-bit kernel (k_u_round).

template <int R, bool UNI = false>
__global__ __launch_bounds__(MT) void k_mis32_round(unsigned *__restrict__ state, int w, int h,
                                                    int r_rt, unsigned *__restrict__ counters,
                                                    unsigned *__restrict__ tile_left,
                                                    const unsigned char *__restrict__ prev_changed,
                                                    unsigned char *__restrict__ cur_changed,
                                                    int gate, int max_local, int round,
                                                    U32Grid ug) {
  extern __shared__ __align__(16) unsigned lds32[];
  const int r = R >= 0 ? R : r_rt;
  const int inW = MW + 2 * r, inH = MH + 2 * r, S = mis_stride(r), SR = MW + 1;
  unsigned *key = lds32;           // inH x S
  unsigned *rmax = key + inH * S;  // inH x SR : row-window tie_max
  __shared__ int any_undecided, scan_all;
  __shared__ unsigned blk_left[2];
  __shared__ int blk_changed[2];
  unsigned *const counter = counters + round;
  const int x0 = blockIdx.x * MW, y0 = blockIdx.y * MH;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  unsigned *const my_left = tile_left + tile;
  const unsigned left_before = *my_left;
  if (left_before == 0) {  // finished tiles stay finished
    if (threadIdx.x == 0) cur_changed[tile] = 0;
    return;
  }
  // the gate flags and the first staging batch are requested together
  int any = gate ? 0 : 1;
  if (gate) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int bx = min(max((int)blockIdx.x + dx, 0), (int)gridDim.x - 1);
        const int by = min(max((int)blockIdx.y + dy, 0), (int)gridDim.y - 1);
        any |= prev_changed[by * gridDim.x + bx];  // clamped: a tile of the neighbourhood again
      }
  }
  constexpr int kStage = 12;
  const int total = inW * inH;
  unsigned sv[kStage];
  int txv[kStage], tyv[kStage];
#pragma unroll
  for (int u = 0; u < kStage; ++u) {
    const int i = min((int)threadIdx.x + u * MT, total - 1);
    tyv[u] = i / inW;
    txv[u] = i - tyv[u] * inW;
    const int x = x0 - r + txv[u], y = y0 - r + tyv[u];
    sv[u] = state[(size_t)min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1)];
  }
  GKLT_ISSUE_TOGETHER_12(sv);
#pragma unroll
  for (int u = 0; u < kStage; ++u) {
    const int x = x0 - r + txv[u], y = y0 - r + tyv[u];
    if (!(x >= 0 && x < w && y >= 0 && y < h)) sv[u] = 0u;
  }
  if (!any) {  // uniform over the workgroup
    if (threadIdx.x == 0) {
      cur_changed[tile] = 0;
      atomicAdd(counter, left_before);
    }
    return;
  }
  if (threadIdx.x == 0) {
    any_undecided = 0;
    scan_all = 0;
    blk_left[0] = blk_left[1] = 0;
    blk_changed[0] = blk_changed[1] = 0;
  }
  // UNI: dense visiting ranks of the cells that meet the haloed tile
  __shared__ unsigned short cell_rank[kU32Cells];
  __shared__ unsigned char cell_field[kU32Cells];
  int cx0 = 0, cy0 = 0, ncx = 1;
  if constexpr (UNI) {
    cx0 = min(max(x0 - r, 0) / ug.gw, ug.cw - 1);
    cy0 = min(max(y0 - r, 0) / ug.gh, ug.ch - 1);
    const int cx1 = min(min(x0 + MW + r - 1, w - 1) / ug.gw, ug.cw - 1);
    const int cy1 = min(min(y0 + MH + r - 1, h - 1) / ug.gh, ug.ch - 1);
    ncx = cx1 - cx0 + 1;
    const int nreg = ncx * (cy1 - cy0 + 1);  // <= kU32Cells (checked on the host)
    if ((int)threadIdx.x < nreg)
      cell_rank[threadIdx.x] = ug.rank[(cy0 + (int)threadIdx.x / ncx) * ug.cw + cx0 + (int)threadIdx.x % ncx];
    __syncthreads();
    if ((int)threadIdx.x < nreg) {
      int dns = 0;
      for (int j = 0; j < nreg; ++j) dns += cell_rank[j] < cell_rank[threadIdx.x];
      cell_field[threadIdx.x] = (unsigned char)(kU32Cells - 1 - dns);
    }
  }
  __syncthreads();
  int mine = 0, intmax = 0;
  auto put = [&](int tx, int ty, unsigned sval) {
    const bool a = (sval & kAccepted) != 0;
    unsigned k = sval == 0 ? 0u : (a ? 0xFFFFFFFFu : sval << 1);
    if (sval != 0 && !a) {
      if constexpr (UNI) {
        const int x = x0 - r + tx, y = y0 - r + ty;  // inside the image: sval != 0
        const int c = (min(y / ug.gh, ug.ch - 1) - cy0) * ncx + min(x / ug.gw, ug.cw - 1) - cx0;
        // does not fit: the host repeats the phase on 64-bit keys.  (2^26 - 1 fits, but in the
        // tile's last-ranked cell -- field 31 -- its key would be 0xFFFFFFFE, and two of them in a
        // window combine to the 'accepted' marker 0xFFFFFFFF.)
        if (sval >= (1u << kU32ScoreBits) - 1) intmax = 1;
        k = ((unsigned)cell_field[c] << kU32ScoreBits | (sval & ((1u << kU32ScoreBits) - 1))) << 1;
      } else {
        intmax |= sval == 0x7FFFFFFFu;
      }
      if (tx >= r && tx < r + MW && ty >= r && ty < r + MH) mine = 1;
    }
    key[ty * S + tx] = k;
  };
#pragma unroll
  for (int u = 0; u < kStage; ++u)
    if ((int)threadIdx.x + u * MT < total) put(txv[u], tyv[u], sv[u]);
  for (int i = threadIdx.x + kStage * MT; i < total; i += MT) {  // radii beyond 10
    const int ty = i / inW, tx = i - ty * inW;
    const int x = x0 - r + tx, y = y0 - r + ty;
    put(tx, ty, (x >= 0 && x < w && y >= 0 && y < h) ? state[(size_t)y * w + x] : 0u);
  }
  if (mine) any_undecided = 1;
  if (intmax) {
    if constexpr (UNI) *ug.need64 = 1;
    else scan_all = 1;
  }
  __syncthreads();
  if (!any_undecided) {
    if (threadIdx.x == 0) {
      *my_left = 0;
      cur_changed[tile] = 0;
    }
    return;
  }
  const bool exact_all = scan_all != 0;
  constexpr int GR = 8, GC = 4;
  int changed_ever = 0;
  unsigned left_now = 0;
  for (int it = 0; it < max_local; ++it) {
    const int pp = it & 1;
    for (int t = threadIdx.x; t < inH * (MW / GR); t += MT) {  // row pass: lane <-> row
      const int ty = t % inH, gq = t / inH;
      unsigned m[GR];
      window_group32<GR>(key + ty * S + GR * gq, 1, r, m);
#pragma unroll
      for (int i = 0; i < GR; ++i) rmax[ty * SR + GR * gq + i] = m[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // see mis_round_tile
      blk_left[pp ^ 1] = 0;
      blk_changed[pp ^ 1] = 0;
    }
    int left = 0, changed = 0;
    for (int t = threadIdx.x; t < MW * (MH / GC); t += MT) {  // column pass: lane <-> column
      const int tx = t % MW, gq = t / MW;
      unsigned m[GC];
      window_group32<GC>(rmax + GC * gq * SR + tx, SR, r, m);
#pragma unroll
      for (int i = 0; i < GC; ++i) {
        const int ty = GC * gq + i;
        const int li = (ty + r) * S + tx + r;
        const unsigned own = key[li];
        if (own == 0 || own == 0xFFFFFFFFu) continue;
        int d;
        if (exact_all) d = exact_window(key, li, S, r, own);
        else if (m[i] == 0xFFFFFFFFu) d = kRejected;           // an accepted neighbour covers it
        else if ((m[i] >> 1) != (own >> 1)) d = kStays;        // a higher score is undecided
        else if (!(m[i] & 1)) d = kAcceptedNow;                // unique top score
        else d = exact_window(key, li, S, r, own);             // top score shared: positions decide
        if (d == kStays) {
          ++left;
          continue;
        }
        // exact_window of other threads may read this entry during the pass: they see the pixel
        // undecided or decided, and either is a state the rule is safe on (see mis_round_tile)
        key[li] = d == kAcceptedNow ? 0xFFFFFFFFu : 0u;
        const unsigned score = UNI ? ((own >> 1) & ((1u << kU32ScoreBits) - 1)) : (own >> 1);
        state[(size_t)(y0 + ty) * w + x0 + tx] = d == kAcceptedNow ? (score | kAccepted) : 0u;
        changed = 1;
      }
    }
    if (left) atomicAdd(&blk_left[pp], (unsigned)left);
    if (changed) blk_changed[pp] = 1;
    __syncthreads();
    left_now = blk_left[pp];
    const int ch = blk_changed[pp];
    changed_ever |= ch;
    if (!ch || left_now == 0) break;
  }
  if (threadIdx.x == 0) {
    *my_left = left_now;
    cur_changed[tile] = (unsigned char)changed_ever;
    if (left_now) atomicAdd(counter, left_now);
  }
}

// ---- rounds on box tops: the plain selection and the uncapped phase of the Uniform selector
// Accepted candidates are more than r apart in x or in y, so a (r+1) x (r+1) box of pixels holds at
// most one of them, and the only candidate of a box that the rule can accept is the box's TOP: its
// highest-priority undecided candidate (every other candidate of the box has that one inside its
// window).  The dense rounds above evaluate the rule at every pixel; here a round evaluates it at
// one pixel per box -- 17 K boxes instead of 2 M pixels at 1080p and r = 10 -- and the work of a
// round shrinks with the number of boxes that are still open:
//   * the state map keeps the invariant that a pixel covered by an accepted one is 0 (rejected):
//     the wavefront that accepts a candidate zeroes its whole (2r+1)^2 window, in LDS and in the
//     global map (also the part that lies in neighbouring tiles);
//   * a workgroup stages its tile of boxes with a halo of r as 64-bit keys (0 = nothing to decide,
//     all ones = accepted); per iteration every open box is taken by one wavefront: box maximum
//     (two DPP reductions) -> top; window maximum > top -> the box waits; otherwise ACCEPT.  A box
//     is closed once it holds an accepted pixel or no candidate.
// Keys are built per tile: (priority field, score, tile row, tile column) orders any two pixels the
// way the global priority does, and the low 16 bits of a box maximum are the position of the top.
// Concurrency: every map entry goes score -> (0 | accepted) once and never changes again, and a
// wavefront accepts only after it has read every entry of the window.  Two candidates within r of
// each other are never both accepted -- the lower one sees the higher one undecided or accepted,
// whatever the interleaving or the age of its halo copy -- so a stale copy only delays a decision,
// as in the dense rounds, and the iterations of a tile need no barrier for their data (the one
// barrier per iteration collects the progress flags).
#ifdef GKLT_TOPS_BIG  // A/B builds (tools/build_timing.sh)
constexpr int kTopsW = 128, kTopsH = 64, kTopsT = 1024;
constexpr size_t kTopsMaxLds = 112 * 1024;
#else
constexpr int kTopsW = 64, kTopsH = 32, kTopsT = 256;
constexpr size_t kTopsMaxLds = 64 * 1024;
#endif
constexpr unsigned long long kOnes = ~0ull;
constexpr int kTopsCellRows = 8;  // Uniform flavour: cell rows a haloed tile may meet (field table in LDS)
struct TopsGeom {
  int B, nbx, nby, cw, ch, inw, inh, ng, S;
};
__host__ __device__ inline TopsGeom tops_geom(int r) {
  TopsGeom g;
  g.B = r + 1;
  g.nbx = (kTopsW + g.B / 2) / g.B > 1 ? (kTopsW + g.B / 2) / g.B : 1;  // nearest to 64 x 32 pixels
  g.nby = (kTopsH + g.B / 2) / g.B > 1 ? (kTopsH + g.B / 2) / g.B : 1;
  g.cw = g.nbx * g.B;
  g.ch = g.nby * g.B;
  g.inw = g.cw + 2 * r;
  g.inh = g.ch + 2 * r;
  g.ng = (g.inw + 3) / 4;  // a row is staged in groups of four pixels
  g.S = (4 * g.ng) | 1;
  return g;
}
inline size_t tops_lds_bytes(int r) {
  const TopsGeom g = tops_geom(r);
  return 8 * (size_t)g.inh * g.S + 2 * (size_t)((1 + kTopsCellRows) * 4 * g.ng + g.inh);
}
// boxes of fewer than 5 x 5 pixels leave a wavefront mostly idle (and a tile would hold more boxes
// than a wavefront's open-box mask has bits): those radii keep the dense rounds
inline bool tops_usable(int r) {
  const TopsGeom g = tops_geom(r);
  return r >= 4 && g.inw <= 255 && g.inh <= 255 && tops_lds_bytes(r) <= kTopsMaxLds;
}

// maximum over the 64 lanes, wave-uniform (full EXEC required).  The four row steps are written
// as v_max_u32_dpp: from update_dpp + max the compiler emits mov, nop, mov_dpp, max per step.  A
// DPP read of a VGPR the previous VALU instruction wrote needs two wait states (s_nop 1).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"  // every lane now holds the maximum of its row of 16
      : "+v"(v));
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

template <int N, typename F>
__device__ __forceinline__ void for_each(int n, F f) {
  const int lane = threadIdx.x & 63;
  if constexpr (N > 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) f(min(lane + 64 * k, n - 1));
  } else {
    for (int i0 = 0; i0 < n; i0 += 64) f(min(lane + i0, n - 1));
  }
}

#ifdef GKLT_TOPS_TIMING
// experiment builds only (tools/exp_tops_timing.py): per workgroup of the LAST launch, the shader
// clock at entry, after staging, at exit, and the iterations it ran
__device__ unsigned long long g_tops_dbg[4 * 8192];
__device__ int g_tops_dbg_round = 1;
#define TOPS_STAMP(k) if (threadIdx.x == 0 && tile < 8192 && round == g_tops_dbg_round) g_tops_dbg[4 * tile + (k)] = (k) == 3 ? (unsigned long long)its : wall_clock64()
#else
#define TOPS_STAMP(k)
#endif

template <int R, bool UNI>
__global__ __launch_bounds__(kTopsT) void k_tops_round(unsigned *__restrict__ state, int w, int h,
                                                       int r_rt, unsigned *__restrict__ counters,
                                                       unsigned *__restrict__ tile_left,
                                                       const unsigned char *__restrict__ prev_changed,
                                                       unsigned char *__restrict__ cur_changed,
                                                       int gate, int max_local, int round,
                                                       U32Grid ug) {
  extern __shared__ __align__(16) unsigned long long ldsk[];
  const int r = R >= 0 ? R : r_rt;
  const TopsGeom G = tops_geom(r);
  const int B = G.B, S = G.S, WW = 2 * r + 1;
  unsigned long long *key = ldsk;  // inh x S
  unsigned short *colcell = reinterpret_cast<unsigned short *>(key + G.inh * S);  // UNI: cell column of a tile column
  [[maybe_unused]] unsigned short *colfld = colcell + 4 * G.ng;                   //      fields, [cell row][tile column]
  unsigned short *rowcell = colfld + kTopsCellRows * 4 * G.ng;                    //      cell row of a tile row
  __shared__ unsigned blk_left[3];
  __shared__ int blk_changed[3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = kTopsT / 64;
  unsigned *const counter = counters + round;
  const int x0 = blockIdx.x * G.cw, y0 = blockIdx.y * G.ch;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  unsigned *const my_left = tile_left + tile;
  [[maybe_unused]] int its = 0;
  TOPS_STAMP(0);
  TOPS_STAMP(1);
  TOPS_STAMP(2);
  TOPS_STAMP(3);
  const unsigned left_before = *my_left;  // open boxes after the tile's last launch
  if (left_before == 0) {                 // closed tiles stay closed
    if (tid == 0) cur_changed[tile] = 0;
    return;
  }
  if (gate) {
    int any = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int bx = min(max((int)blockIdx.x + dx, 0), (int)gridDim.x - 1);
        const int by = min(max((int)blockIdx.y + dy, 0), (int)gridDim.y - 1);
        any |= prev_changed[by * gridDim.x + bx];
      }
    if (!any) {  // uniform over the workgroup: same halo, same conclusions as last time
      if (tid == 0) {
        cur_changed[tile] = 0;
        atomicAdd(counter, left_before);
      }
      return;
    }
  }
  if (tid == 0) {
    blk_left[0] = blk_left[1] = blk_left[2] = 0;
    blk_changed[0] = blk_changed[1] = blk_changed[2] = 0;
  }
  // UNI: the priority field of every (cell row of the tile, tile column), so that a pixel's field is
  // ONE LDS read (four adjacent pixels: one 8-byte read) instead of cell lookups and a rank load
  // per pixel -- two dependent memory round trips per pixel made staging the longest part of a
  // launch.  The host takes this kernel only when a haloed tile meets at most kTopsCellRows cell rows.
  int cy0 = 0;
  const int ncol = 4 * G.ng;
  if constexpr (UNI) {
    for (int t = tid; t < ncol + G.inh; t += kTopsT) {
      if (t < ncol)
        colcell[t] = (unsigned short)min(min(max(x0 - r + t, 0), w - 1) / ug.gw, ug.cw - 1);
      else
        rowcell[t - ncol] = (unsigned short)min(min(max(y0 - r + t - ncol, 0), h - 1) / ug.gh, ug.ch - 1);
    }
    __syncthreads();
    cy0 = rowcell[0];
    for (int t = tid; t < kTopsCellRows * ncol; t += kTopsT) {
      const int rr = t / ncol, c = t - rr * ncol;
      colfld[t] = (unsigned short)(0xFFFFu - ug.rank[min(cy0 + rr, ug.ch - 1) * ug.cw + colcell[c]]);
    }
    __syncthreads();
  }
  // 64-bit key of a map entry sv at tile position (tx, ty): 0 = nothing to decide, all ones = accepted,
  // else (field, score, 255 - ty, 255 - tx); branch-free
  auto make_key = [&](unsigned sv, unsigned fld, int tx, int ty) -> unsigned long long {
    unsigned hi = (fld << 16) | (sv >> 16), lo = (sv << 16) | (unsigned)((0xFF - ty) << 8 | (0xFF - tx));
    const bool acc = (sv & kAccepted) != 0;
    hi = sv == 0 ? 0u : (acc ? 0xFFFFFFFFu : hi);
    lo = sv == 0 ? 0u : (acc ? 0xFFFFFFFFu : lo);
    return ((unsigned long long)hi << 32) | lo;
  };
  // Staging.  A tile whose staged columns (rounded up to whole groups) lie inside the image loads
  // four pixels per load, all loads of a thread in flight together; the others (left and right
  // image border) load pixel by pixel with clamped addresses.
  const bool wide = x0 - r >= 0 && x0 - r + 4 * G.ng <= w;
  if (wide) {
    struct __attribute__((packed, aligned(4))) Quad { unsigned v[4]; };
    constexpr int kG = 5;
    const int ngroups = G.ng * G.inh;
    for (int g0 = tid; g0 < ngroups; g0 += kG * kTopsT) {
      Quad q[kG];
#pragma unroll
      for (int u = 0; u < kG; ++u) {
        const int g = min(g0 + u * kTopsT, ngroups - 1);
        const int ty = g / G.ng, gx = g - ty * G.ng;
        const int y = min(max(y0 - r + ty, 0), h - 1);
        q[u] = *reinterpret_cast<const Quad *>(state + (size_t)y * w + (x0 - r + 4 * gx));
      }
      asm volatile("" : "+v"(q[0].v[0]), "+v"(q[0].v[1]), "+v"(q[0].v[2]), "+v"(q[0].v[3]),
                        "+v"(q[1].v[0]), "+v"(q[1].v[1]), "+v"(q[1].v[2]), "+v"(q[1].v[3]),
                        "+v"(q[2].v[0]), "+v"(q[2].v[1]), "+v"(q[2].v[2]), "+v"(q[2].v[3]),
                        "+v"(q[3].v[0]), "+v"(q[3].v[1]), "+v"(q[3].v[2]), "+v"(q[3].v[3]),
                        "+v"(q[4].v[0]), "+v"(q[4].v[1]), "+v"(q[4].v[2]), "+v"(q[4].v[3]));
#pragma unroll
      for (int u = 0; u < kG; ++u) {
        const int g = g0 + u * kTopsT;
        if (g < ngroups) {
          const int ty = g / G.ng, gx = g - ty * G.ng;
          const int y = y0 - r + ty;
          const bool yin = y >= 0 && y < h;
          unsigned long long f4 = 0ull;  // the four pixels' fields, 16 bits each
          if constexpr (UNI)
            f4 = *reinterpret_cast<const unsigned long long *>(
                colfld + min((int)rowcell[ty] - cy0, kTopsCellRows - 1) * ncol + 4 * gx);
#pragma unroll
          for (int k = 0; k < 4; ++k)  // columns inw .. 4 ng - 1 are staged too and never read
            key[ty * S + 4 * gx + k] =
                make_key(yin ? q[u].v[k] : 0u, (unsigned)(f4 >> (16 * k)) & 0xFFFFu, 4 * gx + k, ty);
        }
      }
    }
  }
  // in batches of unconditional (clamped) loads, so that a batch is in flight together
  constexpr int kStage = 6;
  static_assert(kStage == 6, "the asm statement below names six registers");
  const int total = wide ? 0 : G.inw * G.inh;
  for (int i0 = tid; i0 < total; i0 += kStage * kTopsT) {
    unsigned sv[kStage];
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int i = min(i0 + u * kTopsT, total - 1);
      const int ty = i / G.inw, tx = i - ty * G.inw;
      const int x = x0 - r + tx, y = y0 - r + ty;
      const unsigned v = state[(size_t)min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1)];
      sv[u] = v;
    }
    // all loads of the batch are issued before any is used: without this the compiler sinks each
    // load into the in-image test below and waits for it there, one round trip per entry
    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]));
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int i = i0 + u * kTopsT;
      if (i < total) {
        const int ty = i / G.inw, tx = i - ty * G.inw;
        const int x = x0 - r + tx, y = y0 - r + ty;
        const unsigned sx = (x >= 0 && x < w && y >= 0 && y < h) ? sv[u] : 0u;
        unsigned fld = 0;
        if constexpr (UNI) fld = colfld[min((int)rowcell[ty] - cy0, kTopsCellRows - 1) * ncol + tx];
        key[ty * S + tx] = make_key(sx, fld, tx, ty);
      }
    }
  }
  __syncthreads();
  TOPS_STAMP(1);
  // f(i) for i = lane, lane + 64, ... below n, indices clamped to n - 1 (so that all loads of a pass
  // are unconditional and in flight together); unrolled when the radius is a compile-time constant
  constexpr int NBOX = R >= 0 ? ((R + 1) * (R + 1) + 63) / 64 : 0;
  constexpr int NWIN = R >= 0 ? ((2 * R + 1) * (2 * R + 1) + 63) / 64 : 0;
  const int nb = G.nbx * G.nby;
  unsigned open = 0;  // bit j: box wave + j * NW is open (wave-uniform)
  for (int j = 0; wave + j * NW < nb; ++j) {
    // boxes that start outside the image hold nothing
    const int b = wave + j * NW;
    if (x0 + (b % G.nbx) * B < w && y0 + (b / G.nbx) * B < h) open |= 1u << j;
  }
  int changed_ever = 0;
  unsigned left_now = 0;
  for (int it = 0; it < max_local; ++it) {
    const int pp = it % 3;
    int accepted = 0;
    for (int j = 0; wave + j * NW < nb; ++j) {
      if (!(open >> j & 1)) continue;
      const int b = wave + j * NW;
      const int base = (r + (b / G.nbx) * B) * S + r + (b % G.nbx) * B;
      unsigned long long m = 0ull;
      for_each<NBOX>(B * B, [&](int i) {  // clamped index: a pixel read twice does not change a maximum
        const int by = i / B;
        m = max(m, key[base + by * S + i - by * B]);
      });
      m = wave_max_u64(m);
      if (m == 0ull || m == kOnes) {  // no candidate left / holds an accepted pixel: closed
        open &= ~(1u << j);
        continue;
      }
      const int tx = 0xFF - (int)(m & 0xFF), ty = 0xFF - (int)((m >> 8) & 0xFF);
      const int ctr = ty * S + tx;
      bool higher = false;
      for_each<NWIN>(WW * WW, [&](int i) {
        const int wy = i / WW;
        higher |= key[ctr + (wy - r) * S + (i - wy * WW - r)] > m;
      });
      if (__any(higher)) continue;  // a higher-priority candidate of the window is undecided, or an
                                    // accepted one is about to reject this one: the box waits
      const int cx = x0 - r + tx, cy = y0 - r + ty;
      for_each<NWIN>(WW * WW, [&](int i) {  // the centre is written last, by lane 0 (it may be the
        const int wy = i / WW, dy = wy - r, dx = i - wy * WW - r;  // clamped duplicate of another lane)
        const int x = cx + dx, y = cy + dy;
        if (dx != 0 || dy != 0) {
          key[ctr + dy * S + dx] = 0ull;
          if (x >= 0 && x < w && y >= 0 && y < h) state[(size_t)y * w + x] = 0u;
        }
      });
      if (lane == 0) {
        key[ctr] = kOnes;
        state[(size_t)cy * w + cx] = (unsigned)((m >> 16) & 0x7FFFFFFFu) | kAccepted;
      }
      open &= ~(1u << j);
      accepted = 1;
    }
    if (lane == 0) {
      const unsigned left = __popc(open);
      if (left) atomicAdd(&blk_left[pp], left);
      if (accepted) blk_changed[pp] = 1;
    }
    __syncthreads();
    if (tid == 0) {  // the slots of iteration it + 2 (= it - 1): last read before this barrier, next
      const int q = (it + 2) % 3;  // written after the barrier of iteration it + 1
      blk_left[q] = 0;
      blk_changed[q] = 0;
    }
    left_now = blk_left[pp];
    const int ch = blk_changed[pp];
    changed_ever |= ch;
    its = it + 1;
    if (!ch || left_now == 0) break;
  }
  TOPS_STAMP(2);
  TOPS_STAMP(3);
  if (tid == 0) {
    *my_left = left_now;
    cur_changed[tile] = (unsigned char)changed_ever;
    if (left_now) atomicAdd(counter, left_now);
  }
}

// Appends the accepted pixels of a 256-column x 32-row block to a list, emit(i, x, y, score)
// writing entry i.  ONE global atomic per block: accepted pixels are at least mindist apart, so
// per-pixel (or per-wave) atomics on the single counter serialise (10^4 of them cost ~100 us).
constexpr int kCollectRows = 32;
inline dim3 collect_grid(int w, int h) {
  return dim3((w + 255) / 256, (h + kCollectRows - 1) / kCollectRows);
}
template <typename Emit>
__device__ __forceinline__ void collect_block(const unsigned *__restrict__ state, int w, int h,
                                              unsigned *__restrict__ count, unsigned cap,
                                              Emit emit) {
  __shared__ unsigned wtot[4];
  __shared__ unsigned base;
  const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * kCollectRows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned mask = 0;  // bit r: pixel (x, y0 + r) is accepted
  if (x < w) {
#pragma unroll 8
    for (int r = 0; r < kCollectRows; ++r) {
      const int y = min(y0 + r, h - 1);
      const unsigned s = state[(size_t)y * w + x];
      if (y0 + r < h && (s & kAccepted)) mask |= 1u << r;
    }
  }
  const unsigned c = __popc(mask);
  unsigned inc = c;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  unsigned off = inc - c;
  for (int k = 0; k < wave; ++k) off += wtot[k];
  if (threadIdx.x == 0) {
    const unsigned total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    base = total ? atomicAdd(count, total) : 0u;
  }
  __syncthreads();
  unsigned i = base + off;
  while (mask) {
    const int r = __ffs(mask) - 1;
    mask &= mask - 1;
    if (i < cap) emit(i, x, y0 + r, state[(size_t)(y0 + r) * w + x] & ~kAccepted);
    ++i;
  }
}

// accepted pixels -> key list (order irrelevant, ranked later)
__global__ __launch_bounds__(256) void k_collect(const unsigned *__restrict__ state, int w, int h,
                                                 unsigned long long *__restrict__ list,
                                                 unsigned *__restrict__ count, unsigned cap) {
  collect_block(state, w, h, count, cap,
                [&](unsigned i, int x, int y, unsigned s) { list[i] = prio_key(s, x, y); });
}

// rank[i] += number of keys of slice blockIdx.y that are greater than keys[i] (rank zeroed
// before the launch).  The key range is split over gridDim.y so that ranking ~10^4 accepted
// features is spread over the whole chip instead of 40 workgroups.
constexpr int kRankSlices = 16;
// n = min(*count, cap) is read on the device: the grid is sized for an upper bound and the host does
// not wait for the count.
__global__ __launch_bounds__(256) void k_rank_partial(const unsigned long long *__restrict__ keys,
                                                      const unsigned *__restrict__ count,
                                                      unsigned cap, unsigned *__restrict__ rank) {
  __shared__ unsigned long long tile[256];
  const unsigned n = min(*count, cap);
  if (blockIdx.x * 256 >= n) return;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long own = i < n ? keys[i] : 0ull;
  const unsigned per = ((n + gridDim.y - 1) / gridDim.y + 255) / 256 * 256;
  const unsigned jb = blockIdx.y * per, je = min(n, jb + per);
  unsigned rk = 0;
  for (unsigned j0 = jb; j0 < je; j0 += 256) {
    tile[threadIdx.x] = j0 + threadIdx.x < je ? keys[j0 + threadIdx.x] : 0ull;
    __syncthreads();
    const unsigned m = min(256u, je - j0);
#pragma unroll 8
    for (unsigned j = 0; j < m; ++j) rk += tile[j] > own;
    __syncthreads();
  }
  if (i < n && rk) atomicAdd(&rank[i], rk);
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
__global__ __launch_bounds__(256) void k_assign(const unsigned long long *__restrict__ list,
                                                const unsigned *__restrict__ ranks,
                                                const unsigned *__restrict__ count, unsigned cap,
                                                const int *__restrict__ slots,
                                                const int *__restrict__ nslots,
                                                float *__restrict__ fx, float *__restrict__ fy,
                                                int *__restrict__ fval) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(*count, cap)) return;
  const unsigned long long own = list[i];
  const unsigned rank = ranks[i];
  if ((int)rank < *nslots) {
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
  size_t cap_px = 0, cap_sum = 0, cap_tiles = 0;  // allocated sizes: the scratch of a large image serves smaller ones
  float *sum[3] = {nullptr, nullptr, nullptr};  // (h+1) x (w+1)
  // gklt_select_prepare: integral images computed ahead of the selection, on the build stream, into
  // one of two further sets; tagged with the pyramid and its build generation
  float *pre[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  const gklt_pyramid *pre_py[2] = {nullptr, nullptr};
  unsigned long long pre_gen[2] = {0, 0};
  hipEvent_t pre_done[2] = {nullptr, nullptr};
  int pre_next = 0;
  float *const *active = nullptr;  // the set the selection in progress reads (sum or pre[k])
  const Texel *direct_tex = nullptr;  // non-null: scores from direct window sums of these texels
  unsigned *state = nullptr;
  unsigned char *blocked = nullptr;
  unsigned long long *list = nullptr;
  unsigned *rank = nullptr;      // rank of list[i] among the accepted keys
  unsigned *tile_left = nullptr; // per 64x16 tile: undecided pixels after its last round
  size_t tile_bytes = 0;
  unsigned char *tile_changed = nullptr;  // [2][ntiles]: tile decided something in launch k (slot k & 1)
  size_t ntiles = 0;
  int last_rounds[2] = {0, 0};  // launches of the last selection (Uniform: phase A, phase B)
  unsigned *counters = nullptr;  // [0] accepted count, [1..] per-round undecided
  int *slots = nullptr, *nslots = nullptr;
  unsigned *h_counters = nullptr;  // pinned
  struct UniScratch *uni = nullptr;  // extra buffers of the Uniform flavour
};
struct UniScratch;
void uni_free(UniScratch *u);
constexpr int kMaxRounds = 4096;
constexpr int kRoundBatch = 24;  // rounds per host sync: a converged round costs ~3 us, a sync ~25 us
// words of h_counters behind the per-round counters: values fetched by the same poll as the
// counters (every host synchronisation leaves the GPU idle for ~25-40 us)
constexpr int kPollExtra = 4, kPollNeed64 = kMaxRounds + 1, kPollCount = kMaxRounds + 2;

// every tile live and changed; with_counters: the per-round counters too; flag (optional) cleared
static int rounds_reset(hipStream_t s, SelScratch *sc, bool with_counters, unsigned *flag) {
  const int ntile = (int)sc->cap_tiles, nch = (int)((2 * sc->cap_tiles + 3) / 4);
  const int ncnt = with_counters ? kMaxRounds + 1 : 0;
  const int n = std::max(std::max(ntile, nch), std::max(ncnt, 1));
  hipLaunchKernelGGL(k_rounds_reset, dim3((n + 255) / 256), dim3(256), 0, s, sc->tile_left, ntile,
                     reinterpret_cast<unsigned *>(sc->tile_changed), nch, sc->counters, ncnt, flag);
  GKLT_HIP(hipGetLastError());
  return 0;
}

void sel_free(SelScratch *s) {
  if (!s) return;
  for (auto *p : s->sum) (void)hipFree(p);
  for (auto &set : s->pre)
    for (auto *p : set) (void)hipFree(p);
  for (auto e : s->pre_done)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(s->state);
  (void)hipFree(s->blocked);
  (void)hipFree(s->list);
  (void)hipFree(s->rank);
  (void)hipFree(s->tile_left);
  (void)hipFree(s->tile_changed);
  (void)hipFree(s->counters);
  (void)hipFree(s->slots);
  (void)hipFree(s->nslots);
  if (s->h_counters) (void)hipHostFree(s->h_counters);
  uni_free(s->uni);
  delete s;
}

static int sel_get(gklt_ctx *ctx, int w, int h, int nslots, SelScratch **out) {
  SelScratch *s = ctx->sel;
  const size_t px = (size_t)w * h, sum = (size_t)(w + 1) * (h + 1);
  // tiles of the dense rounds (64 x 16) or of the box rounds (at least 33 x 17), whichever are more
  const size_t tiles = std::max((size_t)((w + MW - 1) / MW) * ((h + MH - 1) / MH),
                                (size_t)((w + 32) / 33) * ((h + 16) / 17));
  if (s && (s->cap_px < px || s->cap_sum < sum || s->cap_tiles < tiles || s->cap_slots < nslots)) {
    GKLT_HIP(hipDeviceSynchronize());
    sel_free(s);
    s = ctx->sel = nullptr;
  }
  if (!s) {
    s = new SelScratch();
    s->cap_px = px;
    s->cap_sum = sum;
    s->cap_tiles = tiles;
    s->cap_slots = std::max(nslots, 4096);
    const int rc = [&]() -> int {
      for (auto &p : s->sum) GKLT_HIP(hipMalloc(&p, sizeof(float) * sum));
      GKLT_HIP(hipMalloc(&s->state, sizeof(unsigned) * px));
      GKLT_HIP(hipMalloc(&s->blocked, px));
      GKLT_HIP(hipMalloc(&s->list, sizeof(unsigned long long) * px));
      GKLT_HIP(hipMalloc(&s->rank, sizeof(unsigned) * px));
      GKLT_HIP(hipMalloc(&s->tile_left, sizeof(unsigned) * tiles));
      GKLT_HIP(hipMalloc(&s->tile_changed, 2 * tiles + 4));  // (cleared in whole words)
      GKLT_HIP(hipMalloc(&s->counters, sizeof(unsigned) * (kMaxRounds + 1)));
      GKLT_HIP(hipMalloc(&s->slots, sizeof(int) * (size_t)s->cap_slots));
      GKLT_HIP(hipMalloc(&s->nslots, sizeof(int)));
      GKLT_HIP(hipHostMalloc(&s->h_counters, sizeof(unsigned) * (kMaxRounds + 1 + kPollExtra), hipHostMallocDefault));
      return 0;
    }();
    if (rc) {  // a scratch is published only when every buffer exists
      sel_free(s);
      return rc;
    }
    ctx->sel = s;
  }
  if (s->w != w || s->h != h) s->last_rounds[0] = s->last_rounds[1] = 0;  // round hint of another image size
  s->w = w;
  s->h = h;
  s->active = s->sum;
  s->direct_tex = nullptr;
  s->cap_list = (int)px;  // mindist 1 accepts every candidate
  s->ntiles = s->cap_tiles;  // the two halves of tile_changed lie cap_tiles apart
  s->tile_bytes = sizeof(unsigned) * s->cap_tiles;
  *out = s;
  return 0;
}

int select_integrals_from_pyramid(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *py,
                                  SelScratch *sc, int level = 0, float *const *dst = nullptr) {
  const int w = py->lv[level].w, h = py->lv[level].h;
  if (!dst) dst = sc->sum;
  {
    ProfScope ps(ctx, s, GKLT_K_INTEGRAL);
    hipLaunchKernelGGL(k_integral_rows, dim3((h + kIR - 1) / kIR), dim3(256), 0, s, py->lv[level].tex, w, h,
                       dst[0], dst[1], dst[2]);
    hipLaunchKernelGGL(k_integral_cols, dim3((w + 1 + 63) / 64, 3), dim3(256), 0, s, dst[0], dst[1], dst[2],
                       w, h);
  }
  GKLT_HIP(hipGetLastError());
  return 0;
}

// Integral images of level 0 for the selection on `py`: the set gklt_select_prepare computed ahead
// (the tracker stream then only waits for it), else computed now on stream s.
static int select_sums_for(gklt_ctx *ctx, hipStream_t s, const gklt_pyramid *py, SelScratch *sc) {
  if (ctx->sel_scores == 1 && ctx->prm.window_width / 2 <= kDMaxHalf && ctx->prm.window_height / 2 <= kDMaxHalf) {
    sc->direct_tex = py->lv[0].tex;  // no integral images at all
    return 0;
  }
  for (int k = 0; k < 2; ++k)
    if (sc->pre_py[k] == py && sc->pre_gen[k] == py->build_gen && sc->pre[k][0]) {
      GKLT_HIP(hipStreamWaitEvent(s, sc->pre_done[k], 0));
      sc->active = sc->pre[k];
      sc->pre_py[k] = nullptr;  // consumed: the set is free again once this selection has returned
      return 0;
    }
  sc->active = sc->sum;
  return select_integrals_from_pyramid(ctx, s, py, sc);
}

// the score pass of a selection: from the active integral images, or directly from the texels
static void launch_score(hipStream_t s, SelScratch *sc, const unsigned char *blocked, int hw, int hh,
                         int bx, int by, int step, int min_eig, int round2, int score_max,
                         unsigned *dst) {
  const int w = sc->w, h = sc->h;
  if (sc->direct_tex && score_direct_lds(hw, hh) > 48 * 1024)  // (at most 93 KB: halves <= kDMaxHalf)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_direct),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)score_direct_lds(hw, hh));
  if (sc->direct_tex)
    hipLaunchKernelGGL(k_score_direct, dim3((w + kDW - 1) / kDW, (h + kDH - 1) / kDH), dim3(256),
                       score_direct_lds(hw, hh), s, sc->direct_tex, blocked, w, h, hw, hh, bx, by, step,
                       min_eig, round2, score_max, dst);
  else
    hipLaunchKernelGGL(k_score, dim3((w + 255) / 256, h), dim3(256), 0, s, sc->active[0], sc->active[1],
                       sc->active[2], blocked, w, h, hw, hh, bx, by, step, min_eig, round2, score_max, dst);
}

// S1-S4 on device-resident sums and a device-resident list (x, y, val of n features).
int select_on_device(gklt_ctx *ctx, hipStream_t s, SelScratch *sc, int mode, int round2, int n,
                     float *fx, float *fy, int *fval, const gklt_params *prm = nullptr) {
  const gklt_params &P = prm ? *prm : ctx->prm;
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
  if (int rc = rounds_reset(s, sc, true, nullptr)) return rc;
  if (!overwrite) {
    GKLT_HIP(hipMemsetAsync(sc->blocked, 0, (size_t)w * h, s));
    if (n > 0)
      hipLaunchKernelGGL(k_block_squares, dim3(n), dim3(256), 0, s, fx, fy, fval, n, w, h,
                         P.mindist, sc->blocked);
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    launch_score(s, sc, overwrite ? (const unsigned char *)nullptr : sc->blocked, hw, hh, bx, by, step,
                 P.min_eigenvalue, round2, 0x7fffffff, sc->state);
  }
  hipLaunchKernelGGL(k_lost_slots, dim3(1), dim3(1024), 0, s, fval, n, overwrite, sc->slots,
                     sc->nslots);
  GKLT_HIP(hipGetLastError());

  // fixpoint rounds (r < 0 happens for mindist 0: every candidate is accepted in one round)
  const int rr = std::max(r, 0);
  // select impl 0: rounds on box tops where the radius allows them, else dense rounds on 32-bit
  // keys; 1: dense rounds on the 64-bit keys shared with the Uniform selector; 2: dense, 32-bit
  const bool tops = ctx->sel_impl == 0 && tops_usable(rr);
  const bool k32 = !tops && ctx->sel_impl != 1;
  const size_t lds = tops ? tops_lds_bytes(rr) : k32 ? mis32_lds_bytes(rr) : mis_lds_bytes(rr);
  if (lds > 160 * 1024) {
    set_error("mindist %d needs %zu bytes of LDS per tile (> 160 KiB)", P.mindist, lds);
    return GKLT_EINVAL;
  }
  // the radii of the reference's defaults (mindist 10: 9 plain, 10 for the ...2 / Uniform flavours)
  // are compiled in; any other radius runs the same code with r at run time
  auto kern = tops ? (rr == 9 ? k_tops_round<9, false> : rr == 10 ? k_tops_round<10, false> : k_tops_round<-1, false>)
              : k32 ? (rr == 9 ? k_mis32_round<9> : rr == 10 ? k_mis32_round<10> : k_mis32_round<-1>)
                    : (rr == 9 ? k_mis_round<9> : rr == 10 ? k_mis_round<10> : k_mis_round<-1>);
  if (lds > 48 * 1024)
    GKLT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const TopsGeom tg = tops_geom(rr);
  const dim3 grid = tops ? dim3((w + tg.cw - 1) / tg.cw, (h + tg.ch - 1) / tg.ch)
                         : dim3((w + MW - 1) / MW, (h + MH - 1) / MH);
  // a tile is skipped while neither it nor one of its 8 neighbours decided anything: valid only if
  // nothing farther away can change what the tile sees (its halo of r; the box rounds also write
  // up to r deep into the neighbouring tiles)
  const int gate = tops ? (tg.cw >= 2 * rr && tg.ch >= 2 * rr) : (rr <= MH);
  const int local_iters = tops ? ctx->sel_tops_iters : ctx->sel_local_iters;
  int round = 1;
  bool done = false;
  while (!done) {
    const int first = round;
    {
      ProfScope ps(ctx, s, GKLT_K_SELECT);
      // first batch: what the previous selection on this context needed plus a margin (video:
      // the count hardly changes from frame to frame); a converged launch costs ~4 us, a sync ~25
      const int nbatch = first == 1 && sc->last_rounds[0] > 0
                             ? std::min(std::max(sc->last_rounds[0] + 2, 4), kRoundBatch) : kRoundBatch;
      for (int b = 0; b < nbatch && round <= kMaxRounds; ++b, ++round)
        hipLaunchKernelGGL(kern, grid, dim3(tops ? kTopsT : MT), lds, s, sc->state, w, h, rr, sc->counters,
                           sc->tile_left, sc->tile_changed + ((round - 1) & 1) * sc->ntiles,
                           sc->tile_changed + (round & 1) * sc->ntiles, gate, local_iters,
                           round, U32Grid{});
    }
    GKLT_HIP(hipGetLastError());
    GKLT_HIP(hipMemcpyAsync(sc->h_counters + first, sc->counters + first,
                            sizeof(unsigned) * (round - first), hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipStreamSynchronize(s));
    for (int k = first; k < round && !done; ++k)
      if (sc->h_counters[k] == 0) {
        done = true;
        sc->last_rounds[0] = k;
        sc->last_rounds[1] = 0;
      }
    if (!done && round > kMaxRounds) {
      set_error("min-distance suppression did not converge in %d rounds", kMaxRounds);
      return GKLT_EHIP;
    }
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    hipLaunchKernelGGL(k_collect, collect_grid(w, h), dim3(256), 0, s, sc->state, w, h, sc->list,
                       sc->counters, (unsigned)sc->cap_list);
  }
  // accepted features are more than r apart in x or in y: at most one per (r+1) x (r+1) block.
  // Ranking and slot assignment are sized for that bound and read the count on the device.
  const unsigned nbound = (unsigned)std::min<long long>(
      sc->cap_list, (long long)((w + rr) / (rr + 1)) * ((h + rr) / (rr + 1)));
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    const unsigned nb = (nbound + 255) / 256;
    GKLT_HIP(hipMemsetAsync(sc->rank, 0, sizeof(unsigned) * nbound, s));
    hipLaunchKernelGGL(k_rank_partial, dim3(nb, kRankSlices), dim3(256), 0, s, sc->list,
                       sc->counters, nbound, sc->rank);
    hipLaunchKernelGGL(k_assign, dim3(nb), dim3(256), 0, s, sc->list, sc->rank, sc->counters,
                       nbound, sc->slots, sc->nslots, fx, fy, fval);
    if (n > 0)
      hipLaunchKernelGGL(k_fill_rest, dim3((n + 255) / 256), dim3(256), 0, s, sc->counters,
                         nbound, sc->slots, sc->nslots, fx, fy, fval);
  }
  GKLT_HIP(hipGetLastError());
  return 0;
}


// ====================================================================== S5: Uniform selection
// _KLTSelectGoodFeaturesUniform (selectGoodFeatures.hpp:1064-1316) walks the grid cells centre-out
// and each cell's candidates in descending score.  While list slots remain it accepts every
// candidate no accepted feature covers (radius mindist, :1288).  Once the slots are used up
// (indx >= nFeatures) it keeps going but grows the list by one slot at a time and leaves a cell
// as soon as that cell holds more than minFeatureInGrid features (:1252-1271).  Which candidates
// are accepted is therefore a priority-greedy walk with a per-cell cap:
//   cells before the one where the slots ran out: no cap;
//   that cell: cap = max(features it held when they ran out, minFeatureInGrid + 1);
//   later cells: cap = minFeatureInGrid + 1,
// priority = (cell visiting order, val desc, y asc, x asc).  The same fixpoint as the plain
// selector computes it, with two extra conditions on ACCEPT in capped cells: the candidate must
// also be the highest-priority undecided candidate of its own cell (one acceptance per cell and
// round keeps the count exact) and the cell must still be below its cap; a full cell rejects all
// its remaining candidates.  Phase A runs uncapped and locates the candidate T that takes the
// last original slot; everything of lower priority is then re-decided under the caps (phase B).
// A grown slot that no later candidate fills stays as one trailing KLT_NOT_FOUND entry.
namespace {

struct UGrid {
  int gw, gh, cw, ch;
  const unsigned short *rank;  // visiting rank of each cell (0 = first)
};

__device__ __forceinline__ int ucell(int x, int y, const UGrid &g) {
  // the reference indexes past its grid in the remainder strip (W % gw != 0); clamped here
  return min(y / g.gh, g.ch - 1) * g.cw + min(x / g.gw, g.cw - 1);
}
__device__ __forceinline__ unsigned long long ukey(unsigned val, int x, int y, const UGrid &g) {
  const int cx = min(x / g.gw, g.cw - 1), cy = min(y / g.gh, g.ch - 1);
  const int lx = x - cx * g.gw, ly = y - cy * g.gh;  // < 256 (checked on the host)
  return ((unsigned long long)(0xFFFF - g.rank[cy * g.cw + cx]) << 48) |
         ((unsigned long long)val << 16) | ((unsigned long long)(0xFF - ly) << 8) |
         (unsigned long long)(0xFF - lx);
}

struct UCaps {
  int enabled;
  int rankT, capT, capAfter;
  const unsigned long long *cellmax;
  int *cellcount;
};

__global__ void k_u_cellmax(const unsigned *__restrict__ state, int w, int h, UGrid g,
                            unsigned long long *__restrict__ cellmax) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const unsigned s = state[(size_t)y * w + x];
  if (s != 0 && !(s & kAccepted)) atomicMax(&cellmax[ucell(x, y, g)], ukey(s, x, y, g));
}

template <int R>
__global__ __launch_bounds__(MT) void k_u_round(unsigned *__restrict__ state, int w, int h, int r,
                                                UGrid g, UCaps caps,
                                                unsigned *__restrict__ counters,
                                                unsigned *__restrict__ tile_left,
                                                const unsigned char *__restrict__ prev_changed,
                                                unsigned char *__restrict__ cur_changed, int gate,
                                                int max_local, int round) {
  mis_round_tile<R>(
      state, w, h, r, counters + round, tile_left, prev_changed, cur_changed, gate, max_local,
      [&](unsigned s, int x, int y) { return ukey(s, x, y, g); },
      [&](size_t gi, int x, int y, unsigned long long own, bool covered, bool is_max) -> int {
        const unsigned val = (unsigned)((own >> 16) & 0x7fffffffu);
        if (covered) {
          state[gi] = 0;
          return kRejected;
        }
        if (!caps.enabled) {
          if (!is_max) return kStays;
          state[gi] = val | kAccepted;
          return kAcceptedNow;
        }
        const int c = ucell(x, y, g);
        const int rk = 0xFFFF - (int)(own >> 48);
        const int cap = rk < caps.rankT ? 0x7fffffff : (rk == caps.rankT ? caps.capT : caps.capAfter);
        if (caps.cellcount[c] >= cap) {
          state[gi] = 0;  // the cell is full (:1253-1254)
          return kRejected;
        }
        if (is_max && caps.cellmax[c] == own) {
          state[gi] = val | kAccepted;
          atomicAdd(&caps.cellcount[c], 1);
          return kAcceptedNow;
        }
        return kStays;
      });
}

__global__ __launch_bounds__(256) void k_u_collect(const unsigned *__restrict__ state, int w,
                                                   int h, UGrid g,
                                                   unsigned long long *__restrict__ keys,
                                                   unsigned *__restrict__ pos,
                                                   unsigned *__restrict__ count, unsigned cap) {
  collect_block(state, w, h, count, cap, [&](unsigned i, int x, int y, unsigned s) {
    keys[i] = ukey(s, x, y, g);
    pos[i] = (unsigned)(y * w + x);
  });
}

// out[0] = key, out[1] = pos of the entry whose rank is `target`
__global__ void k_u_find(const unsigned long long *__restrict__ keys,
                         const unsigned *__restrict__ pos, const unsigned *__restrict__ rank,
                         unsigned n, unsigned target, unsigned long long *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rank[i] == target) {
    out[0] = keys[i];
    out[1] = pos[i];
  }
}

// out[2] = number of accepted keys >= T in T's cell
__global__ void k_u_count_cell(const unsigned long long *__restrict__ keys, unsigned n,
                               const unsigned long long *__restrict__ T,
                               unsigned long long *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keys[i] >= T[0] && (keys[i] >> 48) == (T[0] >> 48)) atomicAdd(&out[2], 1ull);
}

// candidates of lower priority than T go back to "undecided"
__global__ void k_u_reset(unsigned *__restrict__ state, const unsigned *__restrict__ state0, int w,
                          int h, UGrid g, const unsigned long long *__restrict__ T) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const unsigned s0 = state0[(size_t)y * w + x];
  if (s0 != 0 && ukey(s0, x, y, g) < T[0]) state[(size_t)y * w + x] = s0;
}

// flags[0]: a candidate of lower priority than A exists in A's cell; flags[1]: in a later cell
__global__ void k_u_tail_flags(const unsigned *__restrict__ state0, int w, int h, UGrid g,
                               const unsigned long long *__restrict__ A,
                               unsigned *__restrict__ flags) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const unsigned s0 = state0[(size_t)y * w + x];
  if (s0 == 0) return;
  const unsigned long long k = ukey(s0, x, y, g);
  if (k < A[0]) flags[(k >> 48) == (A[0] >> 48) ? 0 : 1] = 1;
}

__global__ void k_u_assign(const unsigned long long *__restrict__ keys,
                           const unsigned *__restrict__ pos, const unsigned *__restrict__ rank,
                           unsigned n, int w, int nslots, float *__restrict__ fx,
                           float *__restrict__ fy, int *__restrict__ fval) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (int)rank[i] < nslots) {
    const int s = (int)rank[i];
    fx[s] = (float)(pos[i] % (unsigned)w);
    fy[s] = (float)(pos[i] / (unsigned)w);
    fval[s] = (int)((keys[i] >> 16) & 0x7fffffffu);
  }
}

__global__ void k_u_fill(int from, int to, float *__restrict__ fx, float *__restrict__ fy,
                         int *__restrict__ fval) {
  const int s = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (s < to) {
    fx[s] = -1.0f;
    fy[s] = -1.0f;
    fval[s] = GKLT_NOT_FOUND;
  }
}


}  // namespace

struct UniScratch {
  int ncell = 0, w = 0, h = 0;
  int rank_cw = 0, rank_ch = 0;  // grid the visiting ranks on the device were computed for
  int *h_cellcount = nullptr;    // pinned copy of cellcount (read in one poll with the tail flags)
  unsigned *state0 = nullptr, *pos = nullptr, *rank = nullptr, *flags = nullptr;
  unsigned long long *cellmax = nullptr, *tinfo = nullptr;
  int *cellcount = nullptr;
  unsigned short *cellrank = nullptr;
};

void uni_free(UniScratch *u) {
  if (!u) return;
  (void)hipFree(u->state0); (void)hipFree(u->pos); (void)hipFree(u->rank); (void)hipFree(u->flags);
  (void)hipFree(u->cellmax); (void)hipFree(u->tinfo); (void)hipFree(u->cellcount);
  (void)hipFree(u->cellrank);
  if (u->h_cellcount) (void)hipHostFree(u->h_cellcount);
  delete u;
}

static int uni_get(SelScratch *sc, int ncell, UniScratch **out) {
  UniScratch *u = sc->uni;
  if (u && (u->ncell < ncell || u->w != sc->w || u->h != sc->h)) {
    GKLT_HIP(hipDeviceSynchronize());
    uni_free(u);
    u = sc->uni = nullptr;
  }
  if (!u) {
    u = new UniScratch();
    u->ncell = ncell; u->w = sc->w; u->h = sc->h;
    const size_t px = (size_t)sc->w * sc->h;
    const int rc = [&]() -> int {
      GKLT_HIP(hipMalloc(&u->state0, sizeof(unsigned) * px));
      GKLT_HIP(hipMalloc(&u->pos, sizeof(unsigned) * (size_t)sc->cap_list));
      GKLT_HIP(hipMalloc(&u->rank, sizeof(unsigned) * (size_t)sc->cap_list));
      GKLT_HIP(hipMalloc(&u->flags, 4 * sizeof(unsigned)));  // tail flags [0..1], need64 [2]
      GKLT_HIP(hipMalloc(&u->cellmax, sizeof(unsigned long long) * ncell));
      GKLT_HIP(hipMalloc(&u->tinfo, 4 * sizeof(unsigned long long)));
      GKLT_HIP(hipMalloc(&u->cellcount, sizeof(int) * ncell));
      GKLT_HIP(hipMalloc(&u->cellrank, sizeof(unsigned short) * ncell));
      GKLT_HIP(hipHostMalloc(&u->h_cellcount, sizeof(int) * ncell, hipHostMallocDefault));
      return 0;
    }();
    if (rc) {
      uni_free(u);
      return rc;
    }
    sc->uni = u;
  }
  *out = u;
  return 0;
}

// runs fixpoint rounds until no candidate is undecided
// `tail` (optional) is enqueued behind every batch of rounds, before the poll: work that only
// makes sense on the converged state but is cheap enough to run speculatively (the collection of
// the accepted candidates), so that its result comes back with the poll that finds the rounds
// converged instead of costing a host synchronisation of its own.  The poll also fetches the
// need64 flag (h_counters[kPollNeed64]) and counters[0] (h_counters[kPollCount]).
enum { kUniKeys64 = 0, kUniKeys32 = 1, kUniTops = 2 };  // round kernels of the uncapped phase
// clear_counters: the phase starts a selection (or restarts one): round counters and need64 cleared
static int uni_rounds(gklt_ctx *ctx, hipStream_t s, SelScratch *sc, UniScratch *u, const UGrid &g,
                      UCaps caps, int r, int *round_io, int keys, const std::function<int()> &tail,
                      bool clear_counters) {
  const int w = sc->w, h = sc->h;
  const bool keys32 = keys == kUniKeys32, tops = keys == kUniTops;
  const size_t lds = tops ? tops_lds_bytes(r) : keys32 ? mis32_lds_bytes(r) : mis_lds_bytes(r);
  if (lds > 160 * 1024) {
    set_error("mindist %d needs %zu bytes of LDS per tile (> 160 KiB)", r, lds);
    return GKLT_EINVAL;
  }
  auto kern = r == 10 ? k_u_round<10> : k_u_round<-1>;
  auto kern32 = tops ? (r == 10 ? k_tops_round<10, true> : k_tops_round<-1, true>)
                     : (r == 10 ? k_mis32_round<10, true> : k_mis32_round<-1, true>);
  if (lds > 48 * 1024)
    GKLT_HIP(hipFuncSetAttribute(keys32 || tops ? reinterpret_cast<const void *>(kern32) : reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const TopsGeom tg = tops_geom(r);
  const dim3 grid = tops ? dim3((w + tg.cw - 1) / tg.cw, (h + tg.ch - 1) / tg.ch)
                         : dim3((w + MW - 1) / MW, (h + MH - 1) / MH);
  const dim3 pgrid((w + 255) / 256, h);
  if (int rc = rounds_reset(s, sc, clear_counters, clear_counters ? u->flags + 2 : nullptr)) return rc;  // a phase starts: all tiles live
  // with caps a decision also depends on the cell's counters and maximum, which change without
  // any tile of the neighbourhood changing: no gating there
  // (and only where nothing beyond the 8 neighbouring tiles can change what a tile sees)
  const int gate = caps.enabled ? 0 : tops ? (tg.cw >= 2 * r && tg.ch >= 2 * r) : (r <= MH);
  const int local_iters = tops ? ctx->sel_tops_iters : ctx->sel_local_iters;
  const U32Grid ug{g.gw, g.gh, g.cw, g.ch, g.rank, u->flags + 2};
  const int phase = caps.enabled ? 1 : 0;
  int round = *round_io;
  bool done = false;
  while (!done) {
    const int first = round;
    {
      ProfScope ps(ctx, s, GKLT_K_SELECT);
      // first batch: what this phase needed in the previous selection plus a margin (video: the
      // count hardly changes from frame to frame); a converged launch costs a few us, a sync ~25
      const int nbatch = first == 1 && sc->last_rounds[phase] > 0
                             ? std::min(std::max(sc->last_rounds[phase] + 2, 4), kRoundBatch) : kRoundBatch;
      for (int b = 0; b < nbatch && round <= kMaxRounds; ++b, ++round) {
        if (caps.enabled) {
          GKLT_HIP(hipMemsetAsync(u->cellmax, 0, sizeof(unsigned long long) * u->ncell, s));
          hipLaunchKernelGGL(k_u_cellmax, pgrid, dim3(256), 0, s, sc->state, w, h, g, u->cellmax);
        }
        unsigned char *const prev = sc->tile_changed + ((round - 1) & 1) * sc->ntiles;
        unsigned char *const cur = sc->tile_changed + (round & 1) * sc->ntiles;
        if (keys32 || tops)
          hipLaunchKernelGGL(kern32, grid, dim3(tops ? kTopsT : MT), lds, s, sc->state, w, h, r, sc->counters, sc->tile_left, prev,
                             cur, gate, local_iters, round, ug);
        else
          hipLaunchKernelGGL(kern, grid, dim3(MT), lds, s, sc->state, w, h, r, g, caps, sc->counters,
                             sc->tile_left, prev, cur, gate, local_iters, round);
      }
    }
    GKLT_HIP(hipGetLastError());
    if (tail)
      if (int rc = tail()) return rc;
    if (tops) {  // one copy: counters[0] is the number of accepted candidates; no need64 with these keys
      GKLT_HIP(hipMemcpyAsync(sc->h_counters, sc->counters, sizeof(unsigned) * round, hipMemcpyDeviceToHost, s));
      GKLT_HIP(hipStreamSynchronize(s));
      sc->h_counters[kPollCount] = sc->h_counters[0];
      sc->h_counters[kPollNeed64] = 0;
    } else {
      GKLT_HIP(hipMemcpyAsync(sc->h_counters + first, sc->counters + first,
                              sizeof(unsigned) * (round - first), hipMemcpyDeviceToHost, s));
      GKLT_HIP(hipMemcpyAsync(sc->h_counters + kPollNeed64, u->flags + 2, sizeof(unsigned),
                              hipMemcpyDeviceToHost, s));
      GKLT_HIP(hipMemcpyAsync(sc->h_counters + kPollCount, sc->counters, sizeof(unsigned),
                              hipMemcpyDeviceToHost, s));
      GKLT_HIP(hipStreamSynchronize(s));
    }
    for (int k = first; k < round && !done; ++k)
      if (sc->h_counters[k] == 0) {
        done = true;
        sc->last_rounds[phase] = k;
      }
    if (!done && round > kMaxRounds) {
      set_error("uniform selection did not converge in %d rounds", kMaxRounds);
      return GKLT_EHIP;
    }
  }
  *round_io = round;
  return 0;
}

// KLTSelectGoodFeaturesUniform (:1318-1338) on device sums; list of capacity `capacity` on device.
int select_uniform_on_device(gklt_ctx *ctx, hipStream_t s, SelScratch *sc, int gridW, int gridH,
                             int n0, int capacity, float *fx, float *fy, int *fval, int *n_out) {
  const gklt_params &P = ctx->prm;
  const int w = sc->w, h = sc->h;
  if (gridW < 1 || gridH < 1 || w / gridW < 1 || h / gridH < 1) {
    set_error("grid cell %dx%d does not fit a %dx%d image", gridW, gridH, w, h);
    return GKLT_EINVAL;
  }
  const int cw = w / gridW, ch = h / gridH, ncell = cw * ch;  // sizeByGrid, :1330-1331
  // a clamped remainder-strip cell is at most 2*grid-1 wide; local coordinates take 8 bits
  if (ncell > 65535 || w - (cw - 1) * gridW > 256 || h - (ch - 1) * gridH > 256 || P.mindist < 0) {
    set_error("uniform selection supports up to 65535 cells of at most 256x256 px");
    return GKLT_EINVAL;
  }
  const int minFeat = (int)((n0 / (float)ncell) + 0.5);  // :1333
  const int hw = P.window_width / 2, hh = P.window_height / 2;
  const int bx = std::max(P.borderx, hw), by = std::max(P.bordery, hh);
  const int step = P.nSkippedPixels + 1;
  const int r = P.mindist;  // :1288
  UniScratch *u;
  if (int rc = uni_get(sc, ncell, &u)) return rc;

  // initializeGridSequence (:1033-1061): cells by squared distance from the grid centre
  // (ties, unspecified in the reference's std::sort, by index)
  // (kept on the device from one selection to the next: the order depends on the grid only)
  if (u->rank_cw != cw || u->rank_ch != ch) {
    std::vector<std::pair<float, int>> seq(ncell);
    for (int i = 0; i < ncell; ++i) {
      const float dY = (float)(i / cw - (float)ch / 2.0);
      const float dX = (float)(i % cw - (float)cw / 2.0);
      seq[i] = {dX * dX + dY * dY, i};
    }
    std::sort(seq.begin(), seq.end());
    std::vector<unsigned short> rank(ncell);
    for (int k = 0; k < ncell; ++k) rank[seq[k].second] = (unsigned short)k;
    GKLT_HIP(hipMemcpyAsync(u->cellrank, rank.data(), sizeof(unsigned short) * ncell,
                            hipMemcpyHostToDevice, s));
    GKLT_HIP(hipStreamSynchronize(s));
    u->rank_cw = cw;
    u->rank_ch = ch;
  }
  const UGrid g{gridW, gridH, cw, ch, u->cellrank};
  const dim3 pgrid((w + 255) / 256, h);
  // the candidates' scores; evaluated again (same bits) where a later step needs the initial map:
  // a selection whose slots never run out -- every frame of a video after the first few -- then
  // costs no copy of the map
  auto score_into = [&](unsigned *dst) {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    launch_score(s, sc, nullptr, hw, hh, bx, by, step, P.min_eigenvalue, 0, 0x7ffffffe, dst);
  };
  score_into(sc->state);

  int round = 1;
  UCaps nocaps{0, 0, 0, 0, nullptr, nullptr};
  // Phase A on 32-bit keys when a haloed tile meets at most kU32Cells cells (select impl 1: always
  // the 64-bit keys), repeated on 64-bit keys in the rare case that a score does not fit 26 bits.
  // (A cell-by-cell schedule -- one wavefront per cell in visiting order, waiting for its
  // lower-ranked neighbours -- was built and dropped: the longest chain of dependent cells of the
  // centre-out order is 93 cells on the 60 x 33 grid of a 1080p frame, and a lone wavefront needs
  // several microseconds per cell: 12 ms, against 0.64 ms for these rounds.)
  const int tcx = (MW + 2 * r - 2) / gridW + 2, tcy = (MH + 2 * r - 2) / gridH + 2;
  // (box rounds: the tile's field table holds kTopsCellRows cell rows)
  const int keys = ctx->sel_impl == 0 && tops_usable(r) && (tops_geom(r).inh - 1) / gridH + 2 <= kTopsCellRows ? kUniTops
                   : ctx->sel_impl != 1 && tcx * tcy <= kU32Cells ? kUniKeys32 : kUniKeys64;
  const bool keys32 = keys == kUniKeys32;
  // The accepted candidates are collected behind every batch of rounds, so that their number
  // arrives with the poll that sees the rounds converged (one host synchronisation instead of
  // three: rounds, need64, count); the ranking needs the count for its grid and follows the poll.
  auto enqueue_collect = [&]() -> int {
    GKLT_HIP(hipMemsetAsync(sc->counters, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_u_collect, collect_grid(w, h), dim3(256), 0, s, sc->state, w, h, g,
                       sc->list, u->pos, sc->counters, (unsigned)sc->cap_list);
    GKLT_HIP(hipGetLastError());
    return 0;
  };
  auto enqueue_rank = [&](unsigned n_acc) -> int {
    if (n_acc > (unsigned)sc->cap_list) {
      set_error("accepted-feature list overflow");
      return GKLT_ECAPACITY;
    }
    if (n_acc) {
      GKLT_HIP(hipMemsetAsync(u->rank, 0, sizeof(unsigned) * n_acc, s));
      hipLaunchKernelGGL(k_rank_partial, dim3((n_acc + 255) / 256, kRankSlices), dim3(256), 0, s,
                         sc->list, sc->counters, n_acc, u->rank);
    }
    GKLT_HIP(hipGetLastError());
    return 0;
  };
  if (int rc = uni_rounds(ctx, s, sc, u, g, nocaps, r, &round, keys, enqueue_collect, true)) return rc;
  if (keys32 && sc->h_counters[kPollNeed64]) {
    score_into(sc->state);
    round = 1;
    if (int rc = uni_rounds(ctx, s, sc, u, g, nocaps, r, &round, kUniKeys64, enqueue_collect, true)) return rc;
  }

  unsigned nacc = sc->h_counters[kPollCount];
  if (int rc = enqueue_rank(nacc)) return rc;
  int free_slot = 0;
  if ((int)nacc >= n0 && n0 > 0) {
    // the slots ran out at T, the candidate that took the last original slot
    unsigned long long hT[4] = {0, 0, 0, 0};
    GKLT_HIP(hipMemsetAsync(u->tinfo, 0, sizeof hT, s));
    hipLaunchKernelGGL(k_u_find, dim3((nacc + 255) / 256), dim3(256), 0, s, sc->list, u->pos,
                       u->rank, nacc, (unsigned)(n0 - 1), u->tinfo);
    hipLaunchKernelGGL(k_u_count_cell, dim3((nacc + 255) / 256), dim3(256), 0, s, sc->list, nacc,
                       u->tinfo, u->tinfo);
    GKLT_HIP(hipMemcpyAsync(hT, u->tinfo, sizeof hT, hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipStreamSynchronize(s));
    const int a1 = (int)hT[2];
    const int tx = (int)(hT[1] % (unsigned)w), ty = (int)(hT[1] / (unsigned)w);
    const int cellT = std::min(ty / gridH, ch - 1) * cw + std::min(tx / gridW, cw - 1);
    UCaps caps{1, 0xFFFF - (int)(hT[0] >> 48), std::max(a1, minFeat + 1), minFeat + 1, u->cellmax,
               u->cellcount};
    score_into(u->state0);
    hipLaunchKernelGGL(k_u_reset, pgrid, dim3(256), 0, s, sc->state, u->state0, w, h, g, u->tinfo);
    GKLT_HIP(hipMemsetAsync(u->cellcount, 0, sizeof(int) * ncell, s));
    GKLT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(u->cellcount + cellT), a1, 1, s));
    round = 1;
    if (int rc = uni_rounds(ctx, s, sc, u, g, caps, r, &round, kUniKeys64, enqueue_collect, true)) return rc;  // phase B
    nacc = sc->h_counters[kPollCount];
    if (int rc = enqueue_rank(nacc)) return rc;
    // is a grown slot left unfilled?  (see the comment at the top of this section)
    unsigned long long hA[4] = {0, 0, 0, 0};
    unsigned hflags[2] = {0, 0};
    int cntA = 0;
    GKLT_HIP(hipMemsetAsync(u->flags, 0, sizeof hflags, s));
    hipLaunchKernelGGL(k_u_find, dim3((nacc + 255) / 256), dim3(256), 0, s, sc->list, u->pos,
                       u->rank, nacc, nacc - 1, u->tinfo);
    hipLaunchKernelGGL(k_u_tail_flags, pgrid, dim3(256), 0, s, u->state0, w, h, g, u->tinfo,
                       u->flags);
    GKLT_HIP(hipMemcpyAsync(hA, u->tinfo, sizeof hA, hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipMemcpyAsync(hflags, u->flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    // every cell's count comes along (8 KB): which one is needed is known only from hA
    GKLT_HIP(hipMemcpyAsync(u->h_cellcount, u->cellcount, sizeof(int) * ncell, hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipStreamSynchronize(s));
    const int ax = (int)(hA[1] % (unsigned)w), ay = (int)(hA[1] / (unsigned)w);
    const int cellA = std::min(ay / gridH, ch - 1) * cw + std::min(ax / gridW, cw - 1);
    cntA = u->h_cellcount[cellA];
    free_slot = (hflags[1] || (hflags[0] && cntA <= minFeat)) ? 1 : 0;
  }
  const int total = std::max(n0, (int)nacc + free_slot);  // :1257 grows, :1296-1312 pads
  if (total > capacity) {
    set_error("uniform selection needs %d list slots, capacity is %d", total, capacity);
    return GKLT_ECAPACITY;
  }
  {
    ProfScope ps(ctx, s, GKLT_K_SELECT);
    if (nacc)
      hipLaunchKernelGGL(k_u_assign, dim3((nacc + 255) / 256), dim3(256), 0, s, sc->list, u->pos,
                         u->rank, nacc, w, total, fx, fy, fval);
    if (total > (int)nacc)
      hipLaunchKernelGGL(k_u_fill, dim3((total - nacc + 255) / 256), dim3(256), 0, s, (int)nacc,
                         total, fx, fy, fval);
  }
  GKLT_HIP(hipGetLastError());
  *n_out = total;
  return 0;
}

}  // namespace gklt

using namespace gklt;

// ============================================================================ C ABI
extern "C" int gklt_ctx_set_select_iters(gklt_ctx *ctx, int iters) {
  if (!ctx || iters < 1 || iters > 64) return GKLT_EINVAL;
  ctx->sel_local_iters = iters;
  ctx->sel_tops_iters = iters;
  return 0;
}

extern "C" int gklt_ctx_set_select_impl(gklt_ctx *ctx, int impl) {
  if (!ctx || impl < 0 || impl > 2) return GKLT_EINVAL;
  ctx->sel_impl = impl;
  return 0;
}

extern "C" int gklt_ctx_set_select_scores(gklt_ctx *ctx, int mode) {
  if (!ctx || mode < 0 || mode > 1) return GKLT_EINVAL;
  ctx->sel_scores = mode;
  return 0;
}

extern "C" int gklt_ctx_select_rounds(gklt_ctx *ctx, int rounds[2]) {
  if (!ctx || !rounds) return GKLT_EINVAL;
  rounds[0] = ctx->sel ? ctx->sel->last_rounds[0] : 0;
  rounds[1] = ctx->sel ? ctx->sel->last_rounds[1] : 0;
  return 0;
}

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
    if (mode == GKLT_SELECT_ALL) return 1;  // KLTSelectGoodFeaturesUniform: own path
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
  if (!ctx || !sumxx || !sumyy || !sumxy || !nFeatures || width < 1 || height < 1) return GKLT_EINVAL;
  const int n = *nFeatures;
  if (n < 0 || n > capacity || (capacity > 0 && (!x || !y || !val))) return GKLT_EINVAL;
  int round2 = 0;
  const int rv = resolve_variant(mode, variant, &round2);
  if (rv < 0) return rv;
  if (rv == 1) {  // KLTSelectGoodFeaturesUniform (:1318-1338)
    GKLT_HIP(hipSetDevice(ctx->device));
    SelScratch *sc;
    if (int rc = sel_get(ctx, width, height, n, &sc)) return rc;
    if (!ctx->scratch_list || ctx->scratch_list->capacity < capacity) {
      if (ctx->scratch_list) gklt_flist_destroy(ctx->scratch_list);
      ctx->scratch_list = nullptr;
      if (int rc = gklt_flist_create(ctx, std::max(capacity, 4096), &ctx->scratch_list)) return rc;
    }
    gklt_flist *fl = ctx->scratch_list;
    hipStream_t s = ctx->s_trk;
    const size_t sb = sizeof(float) * (size_t)(width + 1) * (height + 1);
    const float *src[3] = {sumxx, sumyy, sumxy};
    for (int i = 0; i < 3; ++i)
      GKLT_HIP(hipMemcpyAsync(sc->sum[i], src[i], sb, hipMemcpyHostToDevice, s));
    int n_out = n;
    if (int rc = select_uniform_on_device(ctx, s, sc, gridW, gridH, n, capacity, fl->x, fl->y,
                                          fl->val, &n_out))
      return rc;
    fl->n = n_out;
    *nFeatures = n_out;
    return gklt_flist_download(ctx, fl, x, y, val);
  }
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
                                        gklt_flist *fl, int gridW, int gridH) {
  if (!ctx || !py || !fl || py->ctx != ctx || fl->ctx != ctx) return GKLT_EINVAL;
  int round2 = 0;
  const int rv = resolve_variant(mode, variant, &round2);
  if (rv < 0) return rv;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (rv == 1) {
    if (!py->has_ready) {
      set_error("selecting on a pyramid that was never built");
      return GKLT_EINVAL;
    }
    SelScratch *sc;
    if (int rc = sel_get(ctx, py->lv[0].w, py->lv[0].h, fl->n, &sc)) return rc;
    hipStream_t s = ctx->s_trk;
    GKLT_HIP(hipStreamWaitEvent(s, py->ready, 0));
    if (int rc = select_sums_for(ctx, s, py, sc)) return rc;
    int n_out = fl->n;
    if (int rc = select_uniform_on_device(ctx, s, sc, gridW, gridH, fl->n, fl->capacity, fl->x,
                                          fl->y, fl->val, &n_out))
      return rc;
    fl->n = n_out;
    GKLT_HIP(hipEventRecord(py->consumed, s));
    py->has_consumed = true;
    return 0;
  }
  if (fl->n == 0) return 0;
  if (!py->has_ready) {
    set_error("selecting on a pyramid that was never built");
    return GKLT_EINVAL;
  }
  SelScratch *sc;
  if (int rc = sel_get(ctx, py->lv[0].w, py->lv[0].h, fl->n, &sc)) return rc;
  hipStream_t s = ctx->s_trk;
  GKLT_HIP(hipStreamWaitEvent(s, py->ready, 0));
  if (int rc = select_sums_for(ctx, s, py, sc)) return rc;
  // replacing with nothing lost is a no-op (:442-447): k_lost_slots yields 0 slots and the
  // fixpoint result is simply not written anywhere, so no host round trip is needed for it
  if (int rc = select_on_device(ctx, s, sc, mode, round2, fl->n, fl->x, fl->y, fl->val)) return rc;
  GKLT_HIP(hipEventRecord(py->consumed, s));
  py->has_consumed = true;
  return 0;
}

extern "C" int gklt_select_prepare(gklt_ctx *ctx, gklt_pyramid *py) {
  if (!ctx || !py || py->ctx != ctx) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (!py->has_ready || !py->built_on_pyr) {
    set_error("gklt_select_prepare needs a pyramid built by gklt_pyramid_build");
    return GKLT_EINVAL;
  }
  const int w = py->lv[0].w, h = py->lv[0].h;
  SelScratch *sc;
  if (int rc = sel_get(ctx, w, h, 0, &sc)) return rc;
  const int k = sc->pre_next;
  sc->pre_next ^= 1;
  if (!sc->pre[k][0]) {
    for (auto &p : sc->pre[k]) GKLT_HIP(hipMalloc(&p, sizeof(float) * sc->cap_sum));
    GKLT_HIP(hipEventCreateWithFlags(&sc->pre_done[k], hipEventDisableTiming));
  }
  // on the build stream, behind the build: runs beside whatever the tracker stream is doing (the
  // integral kernels are chains of dependent adds that occupy a fraction of the chip)
  hipStream_t s = ctx->s_pyr;
  sc->pre_py[k] = nullptr;
  if (int rc = select_integrals_from_pyramid(ctx, s, py, sc, 0, sc->pre[k])) return rc;
  GKLT_HIP(hipEventRecord(sc->pre_done[k], s));
  sc->pre_py[k] = py;
  sc->pre_gen[k] = py->build_gen;
  return 0;
}

// ------------------------------------------------------------------ keyframe seeding
namespace {
// seeds found at `level` in that level's pixels -> level-0 coordinates, alive as KLT_TRACKED
// (System.cpp:415-420: feature.x = key.pt.x * mScaleFactor, feature.val = KLT_TRACKED)
__global__ void k_seed_scale(float *__restrict__ x, float *__restrict__ y, int *__restrict__ val, int n,
                             float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (val[i] >= 0) {
    x[i] = x[i] * scale;
    y[i] = y[i] * scale;
    val[i] = GKLT_TRACKED;
  }
}
}  // namespace

extern "C" int gklt_seed_from_level(gklt_ctx *ctx, gklt_pyramid *py, int level, gklt_flist *fl,
                                    int nFeatures) {
  if (!ctx || !py || !fl || py->ctx != ctx || fl->ctx != ctx || level < 0 || level >= py->nlevels)
    return GKLT_EINVAL;
  if (nFeatures < 0 || nFeatures > fl->capacity) {
    set_error("%d seeds exceed the list capacity %d", nFeatures, fl->capacity);
    return GKLT_ECAPACITY;
  }
  GKLT_HIP(hipSetDevice(ctx->device));
  if (!py->has_ready) {
    set_error("seeding from a pyramid that was never built");
    return GKLT_EINVAL;
  }
  fl->n = nFeatures;
  if (nFeatures == 0) return 0;
  int scale = 1;
  for (int l = 0; l < level; ++l) scale *= py->ss;  // mScaleFactor = 2^(nPyramidLevels-1), System.cpp:111
  // selection in the pixels of `level`: the border shrinks with the level so that every seed,
  // scaled back, lies inside the level-0 border the tracker tests (LK.hpp:886-915)
  gklt_params P = ctx->prm;
  P.borderx = (P.borderx + scale - 1) / scale;
  P.bordery = (P.bordery + scale - 1) / scale;
  SelScratch *sc;
  if (int rc = sel_get(ctx, py->lv[level].w, py->lv[level].h, nFeatures, &sc)) return rc;
  hipStream_t s = ctx->s_trk;
  GKLT_HIP(hipStreamWaitEvent(s, py->ready, 0));
  if (ctx->sel_scores == 1 && P.window_width / 2 <= kDMaxHalf && P.window_height / 2 <= kDMaxHalf)
    sc->direct_tex = py->lv[level].tex;
  else if (int rc = select_integrals_from_pyramid(ctx, s, py, sc, level))
    return rc;
  if (int rc = select_on_device(ctx, s, sc, GKLT_SELECT_ALL, 0, nFeatures, fl->x, fl->y, fl->val, &P))
    return rc;
  hipLaunchKernelGGL(k_seed_scale, dim3((nFeatures + 255) / 256), dim3(256), 0, s, fl->x, fl->y, fl->val,
                     nFeatures, (float)scale);
  GKLT_HIP(hipGetLastError());
  GKLT_HIP(hipEventRecord(py->consumed, s));
  py->has_consumed = true;
  return 0;
}

#ifdef GKLT_TOPS_TIMING
extern "C" int gklt_debug_tops_round(int round) {
  GKLT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gklt::g_tops_dbg_round), &round, sizeof(int)));
  return 0;
}
extern "C" int gklt_debug_tops_timing(unsigned long long *out, int n) {
  GKLT_HIP(hipDeviceSynchronize());
  GKLT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(gklt::g_tops_dbg), sizeof(unsigned long long) * (size_t)n));
  return 0;
}
#endif
