/*
 * ransac_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, like klt_oracle.c: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it) of the geometric outlier
 * rejection that follows the tracker on every frame:
 *   RelativeOutlierRejection::relativeRansac   /root/reference/GSLAM/Estimation.cpp:621-683
 *   its call site                               /root/reference/GSLAM/System.cpp:550-564
 *
 * What the reference fixes, and this file follows line by line (ok_ransac_prepare):
 *   candidates = features with val == KLT_TRACKED and isValid; p1 = keyframe bearing, p2 =
 *   normalize(R^T * current bearing); viewAngle = p1 . p2; viewAngle >= minViewAngle -> isValid =
 *   false, skipped; viewAngle < medViewAngle counted; both projected with K to pixels; fewer than
 *   half of the candidates below medViewAngle -> return -1; otherwise fundamental-matrix RANSAC on the
 *   pixel pairs with threshold 10 px and confidence 0.98; points it rejects lose isValid; returns
 *   the inlier count.
 *
 * What the reference does NOT fix: the RANSAC itself is cv::findFundamentalMat(FM_RANSAC), i.e.
 * OpenCV (not vendored, version unpinned, its own RNG and subset checks).  PARITY UNPINNED for it.
 * (One more known difference: OpenCV is believed to replace FM_RANSAC by LMedS below 15 point
 * pairs; this restatement and the GPU path run threshold RANSAC for every m >= 7.)
 * The algorithm below is OpenCV's published scheme -- 7-point minimal solver (null space of the
 * 7 x 9 epipolar system, cubic det(x f1 + f2) = 0), score = number of points whose larger
 * point-to-epipolar-line distance (squared, both images) is <= threshold^2, adaptive iteration count
 * niters = log(1-p) / log(1 - w^7) capped at max_iters, best hypothesis' mask as the result -- with
 * choices of this build where OpenCV's are internal: a counter-based sample generator
 * (splitmix64), no subset-collinearity retries, the solver run in K-normalised coordinates, Gauss-
 * Jordan in place of an SVD, cubic roots by bisection (pure + - * / sqrt, so that a GPU evaluates
 * exactly the same numbers).  The GPU path (gslam_amd/csrc/gklt_ransac.hip) must return identical
 * masks and counts.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RS_MODEL_POINTS 7
#define RS_BISECT_ITERS 100

static uint64_t rs_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* 7 distinct indices in [0, m) for hypothesis h; 0 if 200 draws did not give 7 distinct ones */
static int rs_sample(uint64_t seed, int h, int m, int idx[RS_MODEL_POINTS]) {
  int k = 0, j;
  uint64_t ctr = 0;
  while (k < RS_MODEL_POINTS && ctr < 200) {
    const int cand = (int)(rs_mix64(seed + ((uint64_t)h << 8) + ctr) % (uint64_t)m);
    int dup = 0;
    ctr++;
    for (j = 0; j < k; j++) dup |= idx[j] == cand;
    if (!dup) idx[k++] = cand;
  }
  return k == RS_MODEL_POINTS;
}

static double rs_det3(const double *m) {
  return (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])) +
         m[2] * (m[3] * m[7] - m[4] * m[6]);
}

static double rs_cubic(double x, double a, double b, double c) { return ((x + a) * x + b) * x + c; }

/* 7-point solver on K-normalised pairs (x1,y1) <-> (x2,y2), x2^T F x1 = 0.  Up to 3 matrices (row-major) */
static int rs_solve7(const double *n1, const double *n2, const int *idx, double F[3][9]) {
  double A[7][9], f1[9], f2[9], M[9], d0, d1, dm1, d2, c0, c1, c2, c3, s13, scale;
  double a, b, c, B, disc, lo[3], hi[3];
  int r, col, j, k, nint, nroots = 0;
  for (k = 0; k < 7; k++) {
    const double x1 = n1[2 * idx[k]], y1 = n1[2 * idx[k] + 1], x2 = n2[2 * idx[k]], y2 = n2[2 * idx[k] + 1];
    A[k][0] = x2 * x1; A[k][1] = x2 * y1; A[k][2] = x2;
    A[k][3] = y2 * x1; A[k][4] = y2 * y1; A[k][5] = y2;
    A[k][6] = x1; A[k][7] = y1; A[k][8] = 1.0;
  }
  for (col = 0; col < 7; col++) { /* Gauss-Jordan, partial pivoting, first maximum wins */
    int p = col;
    double best = fabs(A[col][col]), piv;
    for (r = col + 1; r < 7; r++)
      if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); p = r; }
    if (best < 1e-12) return 0;
    if (p != col)
      for (j = 0; j < 9; j++) { const double t = A[p][j]; A[p][j] = A[col][j]; A[col][j] = t; }
    piv = A[col][col];
    for (j = col; j < 9; j++) A[col][j] = A[col][j] / piv;
    for (r = 0; r < 7; r++) {
      double f;
      if (r == col) continue;
      f = A[r][col];
      for (j = col; j < 9; j++) A[r][j] = A[r][j] - f * A[col][j];
    }
  }
  for (k = 0; k < 7; k++) { f1[k] = -A[k][7]; f2[k] = -A[k][8]; }
  f1[7] = 1.0; f1[8] = 0.0; f2[7] = 0.0; f2[8] = 1.0;
  /* det(x f1 + f2) = c3 x^3 + c2 x^2 + c1 x + c0 from its values at x = 0, 1, -1, 2 */
  d0 = rs_det3(f2);
  for (k = 0; k < 9; k++) M[k] = f1[k] + f2[k];
  d1 = rs_det3(M);
  for (k = 0; k < 9; k++) M[k] = f2[k] - f1[k];
  dm1 = rs_det3(M);
  for (k = 0; k < 9; k++) M[k] = 2.0 * f1[k] + f2[k];
  d2 = rs_det3(M);
  c0 = d0;
  c2 = (d1 + dm1) * 0.5 - c0;
  s13 = (d1 - dm1) * 0.5;
  c3 = (((d2 - c0) - 4.0 * c2) - 2.0 * s13) / 6.0;
  c1 = s13 - c3;
  scale = ((fabs(c0) + fabs(c1)) + fabs(c2)) + fabs(c3);
  if (!(fabs(c3) > 1e-14 * scale)) return 0; /* (also rejects NaN) */
  a = c2 / c3; b = c1 / c3; c = c0 / c3;
  B = 1.0 + fmax(fmax(fabs(a), fabs(b)), fabs(c)); /* Cauchy bound on the roots */
  disc = a * a - 3.0 * b;
  if (disc > 0.0) { /* two critical points split [-B, B] into three monotonic pieces */
    const double sq = sqrt(disc), x1 = (-a - sq) / 3.0, x2 = (-a + sq) / 3.0;
    lo[0] = -B; hi[0] = x1; lo[1] = x1; hi[1] = x2; lo[2] = x2; hi[2] = B;
    nint = 3;
  } else {
    lo[0] = -B; hi[0] = B;
    nint = 1;
  }
  for (k = 0; k < nint; k++) {
    double l = lo[k], h = hi[k], fl = rs_cubic(l, a, b, c), fh = rs_cubic(h, a, b, c), x;
    int it, neg;
    if (!(l < h)) continue;
    if (fl == 0.0) {
      if (k > 0) continue; /* a root at a shared end point is taken once, by the piece it closes */
      x = l;
    } else if (fh == 0.0) {
      x = h;
    } else {
      if ((fl < 0.0) == (fh < 0.0)) continue;
      neg = fl < 0.0;
      for (it = 0; it < RS_BISECT_ITERS; it++) {
        const double mid = 0.5 * (l + h), fm = rs_cubic(mid, a, b, c);
        if ((fm < 0.0) == neg) l = mid; else h = mid;
      }
      x = 0.5 * (l + h);
    }
    for (j = 0; j < 9; j++) F[nroots][j] = x * f1[j] + f2[j];
    nroots++;
  }
  return nroots;
}

/* F in pixels from F in K-normalised coordinates: Kinv^T Fn Kinv */
static void rs_to_pixels(const double *Fn, const double *Ki, double *Fp) {
  double T[9];
  int i, j;
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++)
      T[3 * i + j] = (Fn[3 * i] * Ki[j] + Fn[3 * i + 1] * Ki[3 + j]) + Fn[3 * i + 2] * Ki[6 + j];
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++)
      Fp[3 * i + j] = (Ki[i] * T[j] + Ki[3 + i] * T[3 + j]) + Ki[6 + i] * T[6 + j];
}

/* larger of the two squared point-to-epipolar-line distances (OpenCV's error for FM_RANSAC) */
static double rs_error(const double *F, double x1, double y1, double x2, double y2) {
  double a, b, c, s1, s2, d1, d2, e1, e2;
  a = (F[0] * x1 + F[1] * y1) + F[2];
  b = (F[3] * x1 + F[4] * y1) + F[5];
  c = (F[6] * x1 + F[7] * y1) + F[8];
  s2 = 1.0 / (a * a + b * b);
  d2 = (x2 * a + y2 * b) + c;
  a = (F[0] * x2 + F[3] * y2) + F[6];
  b = (F[1] * x2 + F[4] * y2) + F[7];
  c = (F[2] * x2 + F[5] * y2) + F[8];
  s1 = 1.0 / (a * a + b * b);
  d1 = (x1 * a + y1 * b) + c;
  e1 = d1 * d1 * s1;
  e2 = d2 * d2 * s2;
  return e1 > e2 ? e1 : e2; /* NaN compares false everywhere: such a point is never an inlier */
}

static void rs_inv3(const double *K, double *Ki) {
  const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[5] * K[6] - K[3] * K[8], c02 = K[3] * K[7] - K[4] * K[6];
  const double det = (K[0] * c00 + K[1] * c01) + K[2] * c02;
  Ki[0] = c00 / det; Ki[1] = (K[2] * K[7] - K[1] * K[8]) / det; Ki[2] = (K[1] * K[5] - K[2] * K[4]) / det;
  Ki[3] = c01 / det; Ki[4] = (K[0] * K[8] - K[2] * K[6]) / det; Ki[5] = (K[2] * K[3] - K[0] * K[5]) / det;
  Ki[6] = c02 / det; Ki[7] = (K[1] * K[6] - K[0] * K[7]) / det; Ki[8] = (K[0] * K[4] - K[1] * K[3]) / det;
}

/* OpenCV's RANSACUpdateNumIters */
static int rs_update_iters(double p, double ep, int model_points, int max_iters) {
  double num, denom;
  p = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
  ep = ep < 0.0 ? 0.0 : (ep > 1.0 ? 1.0 : ep);
  num = 1.0 - p > DBL_MIN ? 1.0 - p : DBL_MIN;
  denom = 1.0 - pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* Estimation.cpp:628-666: gating, projection, compaction in index order.  Returns the number of
 * candidates m, or -1 when fewer than half of them are below medViewAngle (:668-670). */
int ok_ransac_prepare(int n, const double *key_norm3, const double *cur_norm3, const int *val, uint8_t *is_valid,
                      const double *R, const double *K, double min_view_angle, double med_view_angle,
                      double *pix1, double *pix2, double *nrm1, double *nrm2, int *index) {
  int i, r, m = 0, med = 0, overall = 0;
  for (i = 0; i < n; i++) {
    double p1[3], q[3], len, va, P[3], Q[3];
    if (!(val[i] == 0 && is_valid[i])) continue;
    overall++;
    for (r = 0; r < 3; r++) p1[r] = key_norm3[3 * (size_t)i + r];
    for (r = 0; r < 3; r++) /* rotation.transpose() * norm */
      q[r] = (R[r] * cur_norm3[3 * (size_t)i] + R[3 + r] * cur_norm3[3 * (size_t)i + 1]) + R[6 + r] * cur_norm3[3 * (size_t)i + 2];
    len = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
    for (r = 0; r < 3; r++) q[r] = q[r] / len;
    va = (p1[0] * q[0] + p1[1] * q[1]) + p1[2] * q[2];
    if (va >= min_view_angle) { is_valid[i] = 0; continue; }
    if (va < med_view_angle) med++;
    for (r = 0; r < 3; r++) {
      P[r] = (K[3 * r] * p1[0] + K[3 * r + 1] * p1[1]) + K[3 * r + 2] * p1[2];
      Q[r] = (K[3 * r] * q[0] + K[3 * r + 1] * q[1]) + K[3 * r + 2] * q[2];
    }
    pix1[2 * m] = P[0] / P[2]; pix1[2 * m + 1] = P[1] / P[2];
    pix2[2 * m] = Q[0] / Q[2]; pix2[2 * m + 1] = Q[1] / Q[2];
    nrm1[2 * m] = p1[0] / p1[2]; nrm1[2 * m + 1] = p1[1] / p1[2];
    nrm2[2 * m] = q[0] / q[2]; nrm2[2 * m + 1] = q[1] / q[2];
    index[m++] = i;
  }
  if (med < 0.5 * overall) return -1;
  return m;
}

/* relativeRansac.  Returns the inlier count (-1 as the reference); F_out (may be NULL) = best F in pixels;
 * iters_out (may be NULL) = hypotheses the adaptive loop examined. */
int ok_relative_ransac(int n, const double *key_norm3, const double *cur_norm3, const int *val, uint8_t *is_valid,
                       const double *R, const double *K, double min_view_angle, double med_view_angle,
                       double threshold, double prob, int max_iters, uint64_t seed, double *F_out, int *iters_out) {
  double *buf = (double *)malloc(sizeof(double) * 8 * (size_t)(n > 0 ? n : 1));
  int *index = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  double *pix1 = buf, *pix2 = buf + 2 * (size_t)n, *nrm1 = buf + 4 * (size_t)n, *nrm2 = buf + 6 * (size_t)n;
  double Ki[9], bestF[9], thr2 = threshold * threshold;
  int m, h, k, i, best = 0, niters = max_iters, examined = 0;
  memset(bestF, 0, sizeof bestF);
  m = ok_ransac_prepare(n, key_norm3, cur_norm3, val, is_valid, R, K, min_view_angle, med_view_angle, pix1, pix2,
                        nrm1, nrm2, index);
  if (iters_out) *iters_out = 0;
  if (m < RS_MODEL_POINTS) { /* -1, or too few pairs: findFundamentalMat returns no mask, nothing is rejected */
    free(buf); free(index);
    return m < 0 ? -1 : 0;
  }
  rs_inv3(K, Ki);
  for (h = 0; h < niters; h++) {
    int idx[RS_MODEL_POINTS], nm;
    double Fn[3][9];
    examined = h + 1;
    if (!rs_sample(seed, h, m, idx)) continue;
    nm = rs_solve7(nrm1, nrm2, idx, Fn);
    for (k = 0; k < nm; k++) {
      double Fp[9];
      int good = 0;
      rs_to_pixels(Fn[k], Ki, Fp);
      for (i = 0; i < m; i++) good += rs_error(Fp, pix1[2 * i], pix1[2 * i + 1], pix2[2 * i], pix2[2 * i + 1]) <= thr2;
      if (good > (best > RS_MODEL_POINTS - 1 ? best : RS_MODEL_POINTS - 1)) {
        best = good;
        memcpy(bestF, Fp, sizeof bestF);
        niters = rs_update_iters(prob, (double)(m - good) / m, RS_MODEL_POINTS, niters);
      }
    }
  }
  if (iters_out) *iters_out = examined;
  if (F_out) memcpy(F_out, bestF, sizeof bestF);
  if (best > 0)
    for (i = 0; i < m; i++)
      if (!(rs_error(bestF, pix1[2 * i], pix1[2 * i + 1], pix2[2 * i], pix2[2 * i + 1]) <= thr2)) is_valid[index[i]] = 0;
  free(buf); free(index);
  return best;
}
