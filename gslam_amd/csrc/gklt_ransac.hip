// Geometric outlier rejection on the tracks of a frame:
// RelativeOutlierRejection::relativeRansac (Estimation.cpp:621-683, called from System.cpp:550-564
// right after KLTTrackFeatures and the bearing-vector epilogue).
//
// The reference gates the candidates by view angle, projects both bearings with K and hands the
// pixel pairs to cv::findFundamentalMat(FM_RANSAC, 10 px, 0.98) -- OpenCV, not vendored: its
// sampler and subset checks are internal, so PARITY with it is UNPINNED.  Built here: the gating
// of Estimation.cpp:628-670 operation for operation, and OpenCV's published RANSAC scheme
// (7-point minimal solver, larger squared point-to-epipolar-line distance <= threshold^2, adaptive
// iteration count, mask of the best hypothesis) with the choices oracle/ransac_oracle.c documents.
// Where a CPU loops over hypotheses until the adaptive count is reached, the GPU evaluates ALL
// max_iters hypotheses at once -- one thread solves one minimal sample (Gauss-Jordan null space,
// cubic by bisection), one workgroup scores one model against every pair -- and the host then
// replays the sequential bookkeeping (first strictly better model wins, niters shrinks) over the
// returned counts, which gives exactly the sequential algorithm's answer.  All arithmetic is fp64,
// unfused, in the oracle's operation order: masks and counts are identical to it.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "gklt_internal.h"

namespace gklt {

struct RansacScratch {
  int cap_n = 0, cap_h = 0;
  double *d_key = nullptr, *d_cur = nullptr;      // n x 3
  int *d_val = nullptr;
  uint8_t *d_valid = nullptr;
  double *d_pts = nullptr;                        // pix1 | pix2 | nrm1 | nrm2, each 2 n
  int *d_index = nullptr;
  double *d_mats = nullptr;                       // R | K | Kinv (27)
  int *d_info = nullptr;                          // m, med, overall
  double *d_F = nullptr;                          // H x 3 x 9, pixels
  int *d_nmodels = nullptr, *d_counts = nullptr;  // H, H x 3
  int *h_pinned = nullptr;                        // info (4) | nmodels (H) | counts (3 H)
};

void ransac_free(RansacScratch *r) {
  if (!r) return;
  (void)hipFree(r->d_key); (void)hipFree(r->d_cur); (void)hipFree(r->d_val); (void)hipFree(r->d_valid);
  (void)hipFree(r->d_pts); (void)hipFree(r->d_index); (void)hipFree(r->d_mats); (void)hipFree(r->d_info);
  (void)hipFree(r->d_F); (void)hipFree(r->d_nmodels); (void)hipFree(r->d_counts);
  if (r->h_pinned) (void)hipHostFree(r->h_pinned);
  delete r;
}

namespace {

constexpr int kModelPoints = 7;
constexpr int kBisectIters = 100;
constexpr int kPrepThreads = 1024;

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Estimation.cpp:628-666 -- one workgroup walks the list in index order, 1024 features at a time, so
// that the candidates are compacted in the order the reference push_back()s them.
__global__ __launch_bounds__(kPrepThreads) void k_rs_prepare(
    int n, const double *__restrict__ key, const double *__restrict__ cur, const int *__restrict__ val,
    uint8_t *__restrict__ valid, const double *__restrict__ mats, double min_va, double med_va,
    double *__restrict__ pts, int *__restrict__ index, int *__restrict__ info) {
  __shared__ int s_scan[kPrepThreads];
  __shared__ int s_base, s_med, s_overall;
  const double *R = mats, *K = mats + 9;
  double *pix1 = pts, *pix2 = pts + 2 * (size_t)n, *nrm1 = pts + 4 * (size_t)n, *nrm2 = pts + 6 * (size_t)n;
  const int t = threadIdx.x;
  if (t == 0) { s_base = 0; s_med = 0; s_overall = 0; }
  __syncthreads();
  for (int first = 0; first < n; first += kPrepThreads) {
    const int i = first + t;
    int keep = 0, med = 0, cand = 0;
    double p1[3], q[3];
    if (i < n && val[i] == 0 && valid[i]) {
      cand = 1;
      for (int r = 0; r < 3; ++r) p1[r] = key[3 * (size_t)i + r];
      const double c0 = cur[3 * (size_t)i], c1 = cur[3 * (size_t)i + 1], c2 = cur[3 * (size_t)i + 2];
      for (int r = 0; r < 3; ++r) q[r] = (R[r] * c0 + R[3 + r] * c1) + R[6 + r] * c2;  // R^T * norm
      const double len = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
      for (int r = 0; r < 3; ++r) q[r] = q[r] / len;
      const double va = (p1[0] * q[0] + p1[1] * q[1]) + p1[2] * q[2];
      if (va >= min_va) {
        valid[i] = 0;  // :643-646
      } else {
        keep = 1;
        med = va < med_va;
      }
    }
    // inclusive scan of `keep` over the workgroup (Hillis-Steele; integers, order-free)
    s_scan[t] = keep;
    __syncthreads();
    for (int off = 1; off < kPrepThreads; off <<= 1) {
      const int v = t >= off ? s_scan[t - off] : 0;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    const int base = s_base, pos = base + s_scan[t] - keep, total = s_scan[kPrepThreads - 1];
    if (keep) {
      double P[3], Q[3];
      for (int r = 0; r < 3; ++r) {
        P[r] = (K[3 * r] * p1[0] + K[3 * r + 1] * p1[1]) + K[3 * r + 2] * p1[2];
        Q[r] = (K[3 * r] * q[0] + K[3 * r + 1] * q[1]) + K[3 * r + 2] * q[2];
      }
      pix1[2 * pos] = P[0] / P[2]; pix1[2 * pos + 1] = P[1] / P[2];
      pix2[2 * pos] = Q[0] / Q[2]; pix2[2 * pos + 1] = Q[1] / Q[2];
      nrm1[2 * pos] = p1[0] / p1[2]; nrm1[2 * pos + 1] = p1[1] / p1[2];
      nrm2[2 * pos] = q[0] / q[2]; nrm2[2 * pos + 1] = q[1] / q[2];
      index[pos] = i;
    }
    if (med) atomicAdd(&s_med, 1);
    if (cand) atomicAdd(&s_overall, 1);
    __syncthreads();
    if (t == 0) s_base = base + total;
    __syncthreads();
  }
  if (t == 0) { info[0] = s_base; info[1] = s_med; info[2] = s_overall; }
}

__device__ __forceinline__ double det3(const double *m) {
  return (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])) +
         m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ __forceinline__ double cubic(double x, double a, double b, double c) { return ((x + a) * x + b) * x + c; }

// one thread = one hypothesis: sample, 7-point solve in K-normalised coordinates, models to pixels
__global__ void k_rs_hypotheses(int H, const int *__restrict__ info, unsigned long long seed, int n,
                                const double *__restrict__ pts, const double *__restrict__ mats,
                                double *__restrict__ Fout, int *__restrict__ nmodels) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const int m = info[0];
  const bool enough_med = !(info[1] < 0.5 * info[2]);
  nmodels[h] = 0;
  if (m < kModelPoints || !enough_med) return;
  const double *nrm1 = pts + 4 * (size_t)n, *nrm2 = pts + 6 * (size_t)n, *Ki = mats + 18;
  int idx[kModelPoints], k = 0;
  for (unsigned long long ctr = 0; k < kModelPoints && ctr < 200; ++ctr) {
    const int cand = (int)(mix64(seed + ((unsigned long long)h << 8) + ctr) % (unsigned long long)m);
    bool dup = false;
    for (int j = 0; j < k; ++j) dup |= idx[j] == cand;
    if (!dup) idx[k++] = cand;
  }
  if (k < kModelPoints) return;
  double A[7][9];
  for (k = 0; k < 7; ++k) {
    const double x1 = nrm1[2 * idx[k]], y1 = nrm1[2 * idx[k] + 1], x2 = nrm2[2 * idx[k]], y2 = nrm2[2 * idx[k] + 1];
    A[k][0] = x2 * x1; A[k][1] = x2 * y1; A[k][2] = x2;
    A[k][3] = y2 * x1; A[k][4] = y2 * y1; A[k][5] = y2;
    A[k][6] = x1; A[k][7] = y1; A[k][8] = 1.0;
  }
  for (int col = 0; col < 7; ++col) {
    int p = col;
    double best = fabs(A[col][col]);
    for (int r = col + 1; r < 7; ++r)
      if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); p = r; }
    if (best < 1e-12) return;
    if (p != col)
      for (int j = 0; j < 9; ++j) { const double t = A[p][j]; A[p][j] = A[col][j]; A[col][j] = t; }
    const double piv = A[col][col];
    for (int j = col; j < 9; ++j) A[col][j] = A[col][j] / piv;
    for (int r = 0; r < 7; ++r) {
      if (r == col) continue;
      const double f = A[r][col];
      for (int j = col; j < 9; ++j) A[r][j] = A[r][j] - f * A[col][j];
    }
  }
  double f1[9], f2[9], M[9];
  for (k = 0; k < 7; ++k) { f1[k] = -A[k][7]; f2[k] = -A[k][8]; }
  f1[7] = 1.0; f1[8] = 0.0; f2[7] = 0.0; f2[8] = 1.0;
  const double d0 = det3(f2);
  for (k = 0; k < 9; ++k) M[k] = f1[k] + f2[k];
  const double d1 = det3(M);
  for (k = 0; k < 9; ++k) M[k] = f2[k] - f1[k];
  const double dm1 = det3(M);
  for (k = 0; k < 9; ++k) M[k] = 2.0 * f1[k] + f2[k];
  const double d2 = det3(M);
  const double c0 = d0, c2 = (d1 + dm1) * 0.5 - c0, s13 = (d1 - dm1) * 0.5;
  const double c3 = (((d2 - c0) - 4.0 * c2) - 2.0 * s13) / 6.0, c1 = s13 - c3;
  const double scale = ((fabs(c0) + fabs(c1)) + fabs(c2)) + fabs(c3);
  if (!(fabs(c3) > 1e-14 * scale)) return;
  const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
  const double B = 1.0 + fmax(fmax(fabs(a), fabs(b)), fabs(c));
  const double disc = a * a - 3.0 * b;
  double lo[3], hi[3];
  int nint;
  if (disc > 0.0) {
    const double sq = sqrt(disc), x1 = (-a - sq) / 3.0, x2 = (-a + sq) / 3.0;
    lo[0] = -B; hi[0] = x1; lo[1] = x1; hi[1] = x2; lo[2] = x2; hi[2] = B;
    nint = 3;
  } else {
    lo[0] = -B; hi[0] = B;
    nint = 1;
  }
  int nroots = 0;
  for (k = 0; k < nint; ++k) {
    double l = lo[k], hh = hi[k], x;
    const double fl = cubic(l, a, b, c), fh = cubic(hh, a, b, c);
    if (!(l < hh)) continue;
    if (fl == 0.0) {
      if (k > 0) continue;
      x = l;
    } else if (fh == 0.0) {
      x = hh;
    } else {
      if ((fl < 0.0) == (fh < 0.0)) continue;
      const bool neg = fl < 0.0;
      for (int it = 0; it < kBisectIters; ++it) {
        const double mid = 0.5 * (l + hh), fm = cubic(mid, a, b, c);
        if ((fm < 0.0) == neg) l = mid; else hh = mid;
      }
      x = 0.5 * (l + hh);
    }
    double Fn[9], T[9];
    for (int j = 0; j < 9; ++j) Fn[j] = x * f1[j] + f2[j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        T[3 * i + j] = (Fn[3 * i] * Ki[j] + Fn[3 * i + 1] * Ki[3 + j]) + Fn[3 * i + 2] * Ki[6 + j];
    double *Fp = Fout + ((size_t)h * 3 + nroots) * 9;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        Fp[3 * i + j] = (Ki[i] * T[j] + Ki[3 + i] * T[3 + j]) + Ki[6 + i] * T[6 + j];
    ++nroots;
  }
  nmodels[h] = nroots;
}

__device__ __forceinline__ double epi_error(const double *F, double x1, double y1, double x2, double y2) {
  double a = (F[0] * x1 + F[1] * y1) + F[2];
  double b = (F[3] * x1 + F[4] * y1) + F[5];
  double c = (F[6] * x1 + F[7] * y1) + F[8];
  const double s2 = 1.0 / (a * a + b * b);
  const double d2 = (x2 * a + y2 * b) + c;
  a = (F[0] * x2 + F[3] * y2) + F[6];
  b = (F[1] * x2 + F[4] * y2) + F[7];
  c = (F[2] * x2 + F[5] * y2) + F[8];
  const double s1 = 1.0 / (a * a + b * b);
  const double d1 = (x1 * a + y1 * b) + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return e1 > e2 ? e1 : e2;
}

// one workgroup = one model of one hypothesis, scored against every pair
__global__ __launch_bounds__(256) void k_rs_score(const int *__restrict__ info, int n, const double *__restrict__ pts,
                                                  const double *__restrict__ Fall, const int *__restrict__ nmodels,
                                                  double thr2, int *__restrict__ counts) {
  const int h = blockIdx.x, k = blockIdx.y;
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  if (k < nmodels[h]) {
    const int m = info[0];
    const double *pix1 = pts, *pix2 = pts + 2 * (size_t)n;
    double F[9];
    for (int j = 0; j < 9; ++j) F[j] = Fall[((size_t)h * 3 + k) * 9 + j];
    int good = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x)
      good += epi_error(F, pix1[2 * i], pix1[2 * i + 1], pix2[2 * i], pix2[2 * i + 1]) <= thr2;
    for (int off = 32; off > 0; off >>= 1) good += __shfl_down(good, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, good);
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[h * 3 + k] = s_cnt;
}

// pairs the best model rejects lose isValid (Estimation.cpp:676-682)
__global__ void k_rs_mask(const int *__restrict__ info, int n, const double *__restrict__ pts,
                          const double *__restrict__ F9, double thr2, const int *__restrict__ index,
                          uint8_t *__restrict__ valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= info[0]) return;
  const double *pix1 = pts, *pix2 = pts + 2 * (size_t)n;
  double F[9];
  for (int j = 0; j < 9; ++j) F[j] = F9[j];
  if (!(epi_error(F, pix1[2 * i], pix1[2 * i + 1], pix2[2 * i], pix2[2 * i + 1]) <= thr2)) valid[index[i]] = 0;
}

void inv3(const double *K, double *Ki) {
  const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[5] * K[6] - K[3] * K[8], c02 = K[3] * K[7] - K[4] * K[6];
  const double det = (K[0] * c00 + K[1] * c01) + K[2] * c02;
  Ki[0] = c00 / det; Ki[1] = (K[2] * K[7] - K[1] * K[8]) / det; Ki[2] = (K[1] * K[5] - K[2] * K[4]) / det;
  Ki[3] = c01 / det; Ki[4] = (K[0] * K[8] - K[2] * K[6]) / det; Ki[5] = (K[2] * K[3] - K[0] * K[5]) / det;
  Ki[6] = c02 / det; Ki[7] = (K[1] * K[6] - K[0] * K[7]) / det; Ki[8] = (K[0] * K[4] - K[1] * K[3]) / det;
}

// OpenCV's RANSACUpdateNumIters
int update_iters(double p, double ep, int model_points, int max_iters) {
  p = std::min(std::max(p, 0.0), 1.0);
  ep = std::min(std::max(ep, 0.0), 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

int ransac_get(gklt_ctx *ctx, int n, int H, RansacScratch **out) {
  RansacScratch *r = ctx->ransac;
  if (r && (r->cap_n < n || r->cap_h < H)) {
    GKLT_HIP(hipDeviceSynchronize());
    ransac_free(r);
    r = ctx->ransac = nullptr;
  }
  if (!r) {
    r = new RansacScratch();
    const size_t N = (size_t)std::max(n, 4096), HH = (size_t)std::max(H, 1000);
    const int rc = [&]() -> int {
      GKLT_HIP(hipMalloc(&r->d_key, sizeof(double) * 3 * N));
      GKLT_HIP(hipMalloc(&r->d_cur, sizeof(double) * 3 * N));
      GKLT_HIP(hipMalloc(&r->d_val, sizeof(int) * N));
      GKLT_HIP(hipMalloc(&r->d_valid, N));
      GKLT_HIP(hipMalloc(&r->d_pts, sizeof(double) * 8 * N));
      GKLT_HIP(hipMalloc(&r->d_index, sizeof(int) * N));
      GKLT_HIP(hipMalloc(&r->d_mats, sizeof(double) * 27));
      GKLT_HIP(hipMalloc(&r->d_info, sizeof(int) * 4));
      GKLT_HIP(hipMalloc(&r->d_F, sizeof(double) * 27 * HH));
      GKLT_HIP(hipMalloc(&r->d_nmodels, sizeof(int) * HH));
      GKLT_HIP(hipMalloc(&r->d_counts, sizeof(int) * 3 * HH));
      GKLT_HIP(hipHostMalloc(reinterpret_cast<void **>(&r->h_pinned), sizeof(int) * (4 + 4 * HH), hipHostMallocDefault));
      return 0;
    }();
    if (rc) {  // a scratch is published only when every buffer exists
      ransac_free(r);
      return rc;
    }
    r->cap_n = (int)N;
    r->cap_h = (int)HH;
    ctx->ransac = r;
  }
  *out = r;
  return 0;
}

}  // namespace
}  // namespace gklt

using namespace gklt;

extern "C" int gklt_relative_ransac(gklt_ctx *ctx, int n, const double *key_norm3, const double *cur_norm3,
                                    const int *val, uint8_t *is_valid, const double *R, const double *K,
                                    double min_view_angle, double med_view_angle, double threshold, double prob,
                                    int max_iters, uint64_t seed, int *inliers, double *F_out, int *iters_out) {
  if (!ctx || n < 0 || !inliers || !R || !K || (n > 0 && (!key_norm3 || !cur_norm3 || !val || !is_valid)))
    return GKLT_EINVAL;
  if (max_iters < 1 || max_iters > 65535) {
    set_error("relative_ransac: max_iters %d outside [1, 65535]", max_iters);
    return GKLT_EINVAL;
  }
  *inliers = 0;
  if (iters_out) *iters_out = 0;
  if (F_out) std::memset(F_out, 0, 9 * sizeof(double));
  if (n == 0) return 0;
  GKLT_HIP(hipSetDevice(ctx->device));
  RansacScratch *r;
  const int H = max_iters;
  if (int rc = ransac_get(ctx, n, H, &r)) return rc;
  hipStream_t s = ctx->s_trk;
  double mats[27];
  std::memcpy(mats, R, 9 * sizeof(double));
  std::memcpy(mats + 9, K, 9 * sizeof(double));
  inv3(K, mats + 18);
  GKLT_HIP(hipMemcpyAsync(r->d_key, key_norm3, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(r->d_cur, cur_norm3, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(r->d_val, val, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(r->d_valid, is_valid, (size_t)n, hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(r->d_mats, mats, sizeof mats, hipMemcpyHostToDevice, s));
  const double thr2 = threshold * threshold;
  {
    ProfScope ps(ctx, s, GKLT_K_RANSAC);
    hipLaunchKernelGGL(k_rs_prepare, dim3(1), dim3(kPrepThreads), 0, s, n, r->d_key, r->d_cur, r->d_val, r->d_valid,
                       r->d_mats, min_view_angle, med_view_angle, r->d_pts, r->d_index, r->d_info);
    hipLaunchKernelGGL(k_rs_hypotheses, dim3((H + 63) / 64), dim3(64), 0, s, H, r->d_info, (unsigned long long)seed,
                       n, r->d_pts, r->d_mats, r->d_F, r->d_nmodels);
    hipLaunchKernelGGL(k_rs_score, dim3(H, 3), dim3(256), 0, s, r->d_info, n, r->d_pts, r->d_F, r->d_nmodels, thr2,
                       r->d_counts);
  }
  GKLT_HIP(hipGetLastError());
  int *h_info = r->h_pinned, *h_nm = r->h_pinned + 4, *h_cnt = r->h_pinned + 4 + H;
  GKLT_HIP(hipMemcpyAsync(h_info, r->d_info, sizeof(int) * 3, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipMemcpyAsync(h_nm, r->d_nmodels, sizeof(int) * H, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipMemcpyAsync(h_cnt, r->d_counts, sizeof(int) * 3 * H, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  const int m = h_info[0], med = h_info[1], overall = h_info[2];
  const bool too_few_med = med < 0.5 * overall;  // Estimation.cpp:668-670
  int best = 0, best_h = -1, best_k = 0, niters = H, examined = 0;
  if (!too_few_med && m >= kModelPoints) {
    // the sequential loop of a CPU RANSAC, replayed over the counts of all hypotheses
    for (int h = 0; h < niters; ++h) {
      examined = h + 1;
      for (int k = 0; k < h_nm[h]; ++k) {
        const int good = h_cnt[3 * h + k];
        if (good > std::max(best, kModelPoints - 1)) {
          best = good;
          best_h = h;
          best_k = k;
          niters = update_iters(prob, (double)(m - good) / m, kModelPoints, niters);
        }
      }
    }
  }
  if (best > 0) {
    const double *dF = r->d_F + ((size_t)best_h * 3 + best_k) * 9;
    hipLaunchKernelGGL(k_rs_mask, dim3((m + 255) / 256), dim3(256), 0, s, r->d_info, n, r->d_pts, dF, thr2,
                       r->d_index, r->d_valid);
    GKLT_HIP(hipGetLastError());
    if (F_out) GKLT_HIP(hipMemcpyAsync(F_out, dF, 9 * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  // the view-angle gate clears isValid even when the call returns -1 (Estimation.cpp:643-646)
  GKLT_HIP(hipMemcpyAsync(is_valid, r->d_valid, (size_t)n, hipMemcpyDeviceToHost, s));
  GKLT_HIP(hipStreamSynchronize(s));
  *inliers = too_few_med ? -1 : best;
  if (iters_out) *iters_out = examined;
  return 0;
}
