// Host-side sanitizer run ("ASAN host build"): the CPU oracle (oracle/klt_oracle.c,
// ransac_oracle.c), the library's host code (gslam_amd/csrc/gklt_host.cpp) and the shim's host-only
// entry points (KLTCreateTrackingContext, KLTCreateFeatureList, KLTChangeTCPyramid, KLTUpdateTCBorder,
// computeKernels, _KLTComputeSmoothSigma, KLTCountRemainingFeatures), compiled with
// -fsanitize=address,undefined and run on a small synthetic case by tests/test_sanitize.py in the
// build container (never on the GPU box; nothing here touches a GPU).  Exit code 0 = every check held
// and the sanitizers reported nothing (they abort the process otherwise).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "KLTUtil.h"
#include "LK.hpp"
#include "error.h"
#include "selectGoodFeatures.hpp"

#include <gklt.h>

#include "../../oracle/klt_oracle.h"

void KLTChangeTCPyramid(KLT_TrackingContext tc, int search_range);
void KLTUpdateTCBorder(KLT_TrackingContext tc);
extern "C" int ok_relative_ransac(int n, const double *key_norm3, const double *cur_norm3, const int *val,
                                  uint8_t *is_valid, const double *R, const double *K, double min_view_angle,
                                  double med_view_angle, double threshold, double prob, int max_iters,
                                  uint64_t seed, double *F_out, int *iters_out);

void KLTError(char *fmt, ...) {
  va_list a;
  va_start(a, fmt);
  vfprintf(stderr, fmt, a);
  va_end(a);
  exit(1);
}
void KLTWarning(char *, ...) {}

#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
      return 1;                                                   \
    }                                                             \
  } while (0)

static unsigned lcg(unsigned &s) { return s = s * 1664525u + 1013904223u; }

int main() {
  // ---- shim host-only entry points + gklt_host.cpp
  KLT_TrackingContext tc = KLTCreateTrackingContext();
  CHECK(tc->window_width == 21 && tc->nPyramidLevels == 2 && tc->subsampling == 2 && tc->borderx == 32);
  for (int win : {3, 6, 7, 11, 21, 31}) {
    tc->window_width = tc->window_height = win;
    KLTChangeTCPyramid(tc, 15);
    KLTUpdateTCBorder(tc);
    CHECK(tc->window_width % 2 == 1 && tc->borderx > 0);
  }
  ConvolutionKernel kg, kd;
  for (float sigma : {0.3f, 0.7f, 1.0f, 1.8f, 2.1f, 5.0f, 9.0f}) {
    computeKernels(sigma, &kg, &kd);
    CHECK(kg.width >= 1 && kg.width <= MAX_KERNEL_WIDTH && kd.width <= MAX_KERNEL_WIDTH);
    float g2[OK_MAX_KERNEL_WIDTH], d2[OK_MAX_KERNEL_WIDTH];
    int gw2, dw2;
    CHECK(ok_compute_kernels(sigma, g2, &gw2, d2, &dw2) == 0 && gw2 == kg.width && dw2 == kd.width);
    CHECK(std::memcmp(g2, kg.data, sizeof(float) * gw2) == 0 && std::memcmp(d2, kd.data, sizeof(float) * dw2) == 0);
  }
  CHECK(_KLTComputeSmoothSigma(tc) > 0);
  KLT_FeatureList fl = KLTCreateFeatureList(tc, 37);
  for (int i = 0; i < 3700; ++i) {  // all 100 n records are writable
    fl->feature[i]->x = (float)i;
    fl->feature[i]->val = i % 5 ? i : -1;
  }
  CHECK(KLTCountRemainingFeatures(fl) == 37 - 8);
  CHECK(gklt_feature_list_capacity(37) == 3700);
  gklt_params gp;
  CHECK(gklt_default_params(&gp) == 0 && gp.window_width == 21);
  CHECK(gklt_count_remaining(37, nullptr) < 0);  // a missing array is an error, not a crash
  free(fl);
  free(tc);

  // ---- oracle: pyramid, tracker, integral images, the three selectors, RANSAC
  const int W = 160, H = 120, n = 120;
  unsigned s = 12345;
  std::vector<uint8_t> f0((size_t)W * H), f1((size_t)W * H);
  std::vector<float> tex((size_t)(W + 8) * (H + 8));
  for (auto &v : tex) v = (float)(lcg(s) >> 24);
  for (int it = 0; it < 3; ++it)  // a little smoothing so that windows are trackable
    for (int y = 1; y < H + 7; ++y)
      for (int x = 1; x < W + 7; ++x) {
        float *p = &tex[(size_t)y * (W + 8) + x];
        *p = 0.25f * (p[-1] + p[1] + p[-(W + 8)] + p[W + 8]);
      }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      f0[(size_t)y * W + x] = (uint8_t)tex[(size_t)(y + 4) * (W + 8) + x + 4];
      f1[(size_t)y * W + x] = (uint8_t)tex[(size_t)(y + 4) * (W + 8) + x + 5];  // one pixel to the left
    }
  ok_params p;
  ok_default_params(&p);
  p.window_width = p.window_height = 7;
  p.nPyramidLevels = 3;
  CHECK(ok_update_tc_border(&p) == 0);
  std::vector<std::vector<float>> store[2];
  std::vector<float *> pyr[2];
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < p.nPyramidLevels; ++l) {
      int wl, hl;
      ok_level_size(&p, W, H, l, &wl, &hl);
      for (int q = 0; q < 3; ++q) store[k].emplace_back((size_t)wl * hl);
    }
  for (int k = 0; k < 2; ++k)
    for (auto &v : store[k]) pyr[k].push_back(v.data());
  CHECK(ok_compute_pyramid(&p, f0.data(), W, H, W, pyr[0].data()) == 0);
  CHECK(ok_compute_pyramid(&p, f1.data(), W, H, W, pyr[1].data()) == 0);
  std::vector<float> x(n), y(n);
  std::vector<int> val(n, 0);
  for (int i = 0; i < n; ++i) {
    x[i] = (float)(lcg(s) % W);   // anywhere, incl. the border strip and the image edge
    y[i] = (float)(lcg(s) % H);
  }
  x[0] = -5.f; y[1] = 1e9f; val[2] = -3;
  const double invK[9] = {1.0 / 200, 0, -0.4, 0, 1.0 / 200, -0.3, 0, 0, 1};
  std::vector<double> nrm(3 * n), vec(3 * n);
  std::vector<int> iters(n);
  CHECK(ok_track_features(&p, W, H, pyr[0].data(), pyr[1].data(), n, x.data(), y.data(), val.data(), invK,
                          nrm.data(), vec.data(), iters.data()) == 0);
  int tracked = 0;
  for (int i = 0; i < n; ++i) tracked += val[i] == OK_KLT_TRACKED;
  CHECK(tracked > 0 && val[2] == -3 && val[0] < 0);
  std::vector<float> sxx((size_t)(W + 1) * (H + 1)), syy(sxx.size()), sxy(sxx.size());
  ok_integral_products(pyr[0][1], pyr[0][2], W, H, sxx.data(), syy.data(), sxy.data());
  for (int variant = 0; variant < 3; ++variant)
    for (int mode = 0; mode < 2; ++mode)
      for (int mindist : {1, 4, 10}) {
        ok_params q = p;
        q.mindist = mindist;
        q.min_eigenvalue = 20;
        const int n0 = 40, cap = 100 * n0;
        std::vector<float> fx(cap, -1.f), fy(cap, -1.f);
        std::vector<int> fv(cap, -1);
        if (mode == OK_REPLACE_SOME)
          for (int i = 0; i < n0; i += 2) { fx[i] = 30.f + i; fy[i] = 40.f; fv[i] = 0; }
        int nf = n0;
        CHECK(ok_select_features(&q, mode, variant, sxx.data(), syy.data(), sxy.data(), W, H, &nf, cap, fx.data(),
                                 fy.data(), fv.data(), 32, 24) == 0);
        CHECK(nf >= n0 && nf <= cap);
      }
  // RANSAC: a pure sideways translation, a third of the pairs corrupted
  const int m = 90;
  std::vector<double> kn(3 * m), cn(3 * m);
  std::vector<int> rv(m, 0);
  std::vector<uint8_t> valid(m, 1);
  for (int i = 0; i < m; ++i) {
    const double X = (lcg(s) % 2000) / 1000.0 - 1.0, Y = (lcg(s) % 2000) / 1000.0 - 1.0, Z = 3.0 + (lcg(s) % 1000) / 500.0;
    double a[3] = {X, Y, Z}, b[3] = {X - 0.4, Y, Z};
    if (i % 3 == 0) b[1] += 0.5;
    const double na = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int k = 0; k < 3; ++k) { kn[3 * i + k] = a[k] / na; cn[3 * i + k] = b[k] / nb; }
  }
  const double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, K[9] = {400, 0, 320, 0, 400, 240, 0, 0, 1};
  double F[9];
  int nit = 0;
  const int inl = ok_relative_ransac(m, kn.data(), cn.data(), rv.data(), valid.data(), R, K, 0.99998629, 0.99984770, 3.0, 0.98, 300, 7,
                                     F, &nit);
  CHECK(inl >= 7 && inl <= m && nit >= 1);  // (what it finds is tests/test_gpu_ransac.py's business)
  printf("sanitizer driver: ok (%d tracked, %d RANSAC inliers)\n", tracked, inl);
  return 0;
}
