// Test driver: calls GSLAM's KLT host API (as implemented by shim/gklt_shim.cpp) the way System
// does, behind a few extern "C" functions that tests/test_shim.py reaches through ctypes.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "KLTUtil.h"
#include "LK.hpp"
#include "error.h"
#include "pyramid.h"
#include "selectGoodFeatures.hpp"

void KLTChangeTCPyramid(KLT_TrackingContext tc, int search_range);  // defined without a header in
void KLTUpdateTCBorder(KLT_TrackingContext tc);                     // the reference (KLTUtil.cpp:87,225)

static int g_warnings = 0;
void KLTError(char *fmt, ...) {  // error.cpp:21-31
  va_list args;
  va_start(args, fmt);
  fprintf(stderr, "KLT Error: ");
  vfprintf(stderr, fmt, args);
  fprintf(stderr, "\n");
  va_end(args);
  exit(1);
}
void KLTWarning(char *fmt, ...) {  // error.cpp:41-53
  va_list args;
  va_start(args, fmt);
  fprintf(stderr, "KLT Warning: ");
  vfprintf(stderr, fmt, args);
  fprintf(stderr, "\n");
  va_end(args);
  ++g_warnings;
}

static KLT_TrackingContext make_tc(int win, int levels, int border) {
  KLT_TrackingContext tc = KLTCreateTrackingContext();
  if (win > 0) {  // the overrides System applies from the YAML file (System.cpp:101-109)
    tc->window_width = tc->window_height = win;
    tc->nPyramidLevels = levels;
    if (border >= 0) tc->borderx = tc->bordery = border;
    else KLTUpdateTCBorder(tc);
  }
  return tc;
}

extern "C" {

int drv_warnings() { return g_warnings; }

// KLTCreateTrackingContext: the fields as ints/floats in a fixed order
void drv_defaults(int *iv, float *fv) {
  KLT_TrackingContext tc = KLTCreateTrackingContext();
  int i[] = {tc->mindist, tc->window_width, tc->window_height, tc->min_eigenvalue, tc->max_iterations,
             tc->nSkippedPixels, tc->borderx, tc->bordery, tc->nPyramidLevels, tc->subsampling,
             tc->affineConsistencyCheck, tc->affine_window_width, tc->affine_max_iterations,
             (int)tc->sequentialMode, (int)tc->smoothBeforeSelecting, (int)tc->lighting_insensitive};
  float f[] = {tc->min_determinant, tc->min_displacement, tc->max_residue, tc->grad_sigma, tc->smooth_sigma_fact,
               tc->pyramid_sigma_fact, tc->step_factor, tc->affine_max_residue, tc->affine_min_displacement,
               tc->affine_max_displacement_differ, _KLTComputeSmoothSigma(tc)};
  for (unsigned k = 0; k < sizeof i / sizeof *i; ++k) iv[k] = i[k];
  for (unsigned k = 0; k < sizeof f / sizeof *f; ++k) fv[k] = f[k];
  free(tc);
}

// KLTChangeTCPyramid / KLTUpdateTCBorder on a context with the given window
void drv_pyramid_and_border(int ww, int wh, int search_range, int *out) {
  KLT_TrackingContext tc = KLTCreateTrackingContext();
  tc->window_width = ww;
  tc->window_height = wh;
  KLTChangeTCPyramid(tc, search_range);
  KLTUpdateTCBorder(tc);
  out[0] = tc->window_width; out[1] = tc->window_height; out[2] = tc->nPyramidLevels;
  out[3] = tc->subsampling; out[4] = tc->borderx; out[5] = tc->bordery;
  free(tc);
}

int drv_kernels(float sigma, float *g, int *gw, float *d, int *dw) {
  ConvolutionKernel kg, kd;
  computeKernels(sigma, &kg, &kd);
  *gw = kg.width; *dw = kd.width;
  for (int i = 0; i < kg.width; ++i) g[i] = kg.data[i];
  for (int i = 0; i < kd.width; ++i) d[i] = kd.data[i];
  return 0;
}

// KLTCreateFeatureList(tc, n): layout facts of the one-block AoS list (KLTUtil.cpp:49-83)
void drv_feature_list(int n, long *out) {
  KLT_TrackingContext tc = KLTCreateTrackingContext();
  KLT_FeatureList fl = KLTCreateFeatureList(tc, n);
  const long cap = 100L * n;
  out[0] = fl->nFeatures;
  out[1] = (char *)fl->feature - (char *)fl == (long)sizeof(KLT_FeatureListRec);          // pointers follow the header
  out[2] = (char *)fl->feature[0] - (char *)(fl->feature + cap) == 0;                     // records follow the pointers
  out[3] = fl->feature[cap - 1] == fl->feature[0] + (cap - 1);                            // 100 n contiguous records
  long ok = 1;
  for (long i = 0; i < cap; ++i) {
    const KLT_Feature f = fl->feature[i];
    ok = ok && f->aff_x == -1.0f && f->aff_y == -1.0f && !f->aff_img && !f->aff_gradx && !f->aff_grady;
    f->x = (float)i; f->val = KLT_NOT_FOUND;                                               // every record is writable
  }
  out[4] = ok;
  for (int i = 0; i < n; ++i) fl->feature[i]->val = i % 3 ? i : -1;
  out[5] = KLTCountRemainingFeatures(fl);
  free(fl);
  free(tc);
}

// The front end of two System::Track calls (System.cpp:367-373,501-518): pyramid of frame 0, then
// the pyramid of frame 1 on a preload thread, then KLTTrackFeatures; also the coarsest plane as
// System reads it (System.cpp:407).
int drv_track(int w, int h, int stride, const unsigned char *img0, const unsigned char *img1, int win, int levels,
              int border, int n, float *x, float *y, int *val, const double *invK, double *norm, double *vec,
              float *coarse1, int *coarse_wh) {
  KLT_TrackingContext tc = make_tc(win, levels, border);
  KLT_FeatureList fl = KLTCreateFeatureList(tc, n);
  for (int i = 0; i < n; ++i) { fl->feature[i]->x = x[i]; fl->feature[i]->y = y[i]; fl->feature[i]->val = val[i]; }
  static std::vector<cv::Mat> pyr[2];  // PyramidBuffer keeps its vectors alive (System.h:30-56)
  cv::Mat f0(h, w, CV_8UC1, (void *)img0, stride), f1(h, w, CV_8UC1, (void *)img1, stride);
  computePyramid(tc, f0, pyr[0]);
  std::thread preload(computePyramid, tc, std::ref(f1), std::ref(pyr[1]));
  preload.join();
  if ((int)pyr[1].size() != 3 * levels) return -1;
  Eigen::Matrix3d K;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K(r, c) = invK[3 * r + c];
  fl->isOutlierRejected = true;
  KLTTrackFeatures(tc, pyr[0], pyr[1], fl, K);
  if (fl->isOutlierRejected) return -2;
  for (int i = 0; i < n; ++i) {
    const KLT_Feature f = fl->feature[i];
    x[i] = f->x; y[i] = f->y; val[i] = f->val;
    if (f->val == KLT_TRACKED)
      for (int k = 0; k < 3; ++k) { norm[3 * i + k] = f->norm(k); vec[3 * i + k] = f->vec(k); }
  }
  cv::Mat &c = pyr[1][3 * (levels - 1)];
  coarse_wh[0] = c.cols; coarse_wh[1] = c.rows;
  for (int r = 0; r < c.rows; ++r)
    for (int q = 0; q < c.cols; ++q) coarse1[r * c.cols + q] = c.at<float>(r, q);
  free(fl);
  free(tc);
  return 0;
}

// The frame loop of System::Track as the reference runs it (System.h:30-56, System.cpp:367-373,
// 501-518), on frames in PAGEABLE host memory: PyramidBuffer::next() joins the preload thread and
// rotates the three pyramids, computePyramid of frame t+2 starts on a fresh std::thread, and
// KLTTrackFeatures(t -> t+1) runs meanwhile on the calling thread.  The list is re-seeded from
// (x0, y0) every `reseed_every` frames (0 = never).  All of it is C++: this is what bench.py's
// shim_drop_in block times.  out: [0] seconds for `steps` frames (after `warmup`), [1] tracked
// features in those frames, [2] p50 and [3] p99 of the per-frame time in ms.
int drv_track_stream(int w, int h, int nframes, const unsigned char *frames, int win, int levels, int border,
                     int n, const float *x0, const float *y0, int reseed_every, int warmup, int steps,
                     const double *invK, double *out, float *x, float *y, int *val) {
  if (nframes < 2 || n < 1 || steps < 1) return -1;
  KLT_TrackingContext tc = make_tc(win, levels, border);
  KLT_FeatureList fl = KLTCreateFeatureList(tc, n);
  std::vector<unsigned char *> frame(nframes);  // every frame in its own malloc'd (pageable) block
  for (int k = 0; k < nframes; ++k) {
    frame[k] = static_cast<unsigned char *>(malloc((size_t)w * h));
    memcpy(frame[k], frames + (size_t)k * w * h, (size_t)w * h);
  }
  auto mat = [&](int k) { return cv::Mat(h, w, CV_8UC1, frame[k % nframes], w); };
  Eigen::Matrix3d K;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K(r, c) = invK[3 * r + c];
  static std::vector<cv::Mat> pyr[3];
  std::vector<cv::Mat> *ptrs[3] = {&pyr[0], &pyr[1], &pyr[2]};
  cv::Mat cur = mat(0), nxt = mat(1);
  computePyramid(tc, cur, *ptrs[1]);                                            // System.cpp:367-368
  // GKLT_DRV_BREAKDOWN=1: mean time of the three parts of a frame on stderr
  const bool breakdown = getenv("GKLT_DRV_BREAKDOWN") != nullptr;
  double t_pyr = 0, t_join = 0, t_trk = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto msec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  auto build = [&](const cv::Mat &img, std::vector<cv::Mat> &dst) {  // body of the preload thread
    const auto a = now();
    computePyramid(tc, img, dst);
    t_pyr += msec(a, now());
  };
  std::thread preload(build, std::ref(nxt), std::ref(*ptrs[2]));                 // :370-373
  std::vector<double> ms;
  ms.reserve(steps);
  double tracked = 0;
  std::chrono::steady_clock::time_point t_begin;
  for (int t = 0; t < warmup + steps; ++t) {
    if (t == warmup) t_begin = std::chrono::steady_clock::now();
    const auto q0 = std::chrono::steady_clock::now();
    preload.join();                                                              // PyramidBuffer::next()
    t_join += msec(q0, now());
    std::vector<cv::Mat> *tmp = ptrs[0];
    ptrs[0] = ptrs[1]; ptrs[1] = ptrs[2]; ptrs[2] = tmp;
    nxt = mat(t + 2);
    preload = std::thread(build, std::ref(nxt), std::ref(*ptrs[2]));             // :508-511
    if (reseed_every > 0 ? t % reseed_every == 0 : t == 0)
      for (int i = 0; i < n; ++i) { fl->feature[i]->x = x0[i]; fl->feature[i]->y = y0[i]; fl->feature[i]->val = 0; }
    const auto q1 = now();
    KLTTrackFeatures(tc, *ptrs[0], *ptrs[1], fl, K);                             // :515-518
    t_trk += msec(q1, now());
    if (t >= warmup) {
      for (int i = 0; i < n; ++i) tracked += fl->feature[i]->val == KLT_TRACKED;
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q0).count());
    }
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  preload.join();
  if (breakdown)
    fprintf(stderr, "drv_track_stream: per frame computePyramid %.3f ms (preload thread), wait for it %.3f ms, "
            "KLTTrackFeatures %.3f ms\n", t_pyr / (warmup + steps + 1), t_join / (warmup + steps), t_trk / (warmup + steps));
  std::sort(ms.begin(), ms.end());
  out[0] = secs; out[1] = tracked; out[2] = ms[ms.size() / 2]; out[3] = ms[std::min(ms.size() - 1, ms.size() * 99 / 100)];
  for (int i = 0; i < n; ++i) { x[i] = fl->feature[i]->x; y[i] = fl->feature[i]->y; val[i] = fl->feature[i]->val; }
  for (auto p : frame) free(p);
  free(fl);
  free(tc);
  return 0;
}

// which: 0 KLTSelectGoodFeatures, 1 KLTReplaceLostFeatures, 2 ...2, 3 Replace...2, 4 ...Uniform, 5 Replace...Uniform
int drv_select(int which, int W, int H, const float *sxx, const float *syy, const float *sxy, int win, int levels,
               int border, int min_eig, int mindist, int grid_w, int grid_h, int n_in, int cap, float *x, float *y,
               int *val, float *aff_x_out) {
  KLT_TrackingContext tc = make_tc(win, levels, border);
  tc->min_eigenvalue = min_eig;
  tc->mindist = mindist;
  gridSize = cv::Size(grid_w, grid_h);
  KLT_FeatureList fl = KLTCreateFeatureList(tc, n_in);
  for (int i = 0; i < n_in; ++i) {
    const KLT_Feature f = fl->feature[i];
    f->x = x[i]; f->y = y[i]; f->val = val[i];
    f->aff_x = 7.0f;  // a mark the flavours that reset the affine state overwrite with -1
  }
  cv::Mat a(H + 1, W + 1, CV_32F, (void *)sxx), b(H + 1, W + 1, CV_32F, (void *)syy), c(H + 1, W + 1, CV_32F, (void *)sxy);
  switch (which) {
    case 0: KLTSelectGoodFeatures(tc, a, b, c, fl); break;
    case 1: KLTReplaceLostFeatures(tc, a, b, c, fl); break;
    case 2: KLTSelectGoodFeatures2(tc, a, b, c, fl); break;
    case 3: KLTReplaceLostFeatures2(tc, a, b, c, fl); break;
    case 4: KLTSelectGoodFeaturesUniform(tc, a, b, c, fl); break;
    case 5: KLTReplaceLostFeaturesUniform(tc, a, b, c, fl); break;
    default: return -1;
  }
  const int n = fl->nFeatures;
  if (n > cap) return -2;
  for (int i = 0; i < n; ++i) {
    x[i] = fl->feature[i]->x; y[i] = fl->feature[i]->y; val[i] = fl->feature[i]->val;
    aff_x_out[i] = fl->feature[i]->aff_x;
  }
  free(fl);
  free(tc);
  return n;
}

int drv_smooth_grad(int w, int h, const float *img, float sigma, float *sm, float *gx, float *gy) {
  cv::Mat I(h, w, CV_32F, (void *)img), S, GX, GY;
  computeSmoothedImage(I, S, sigma);
  computeGradients(I, GX, GY, sigma);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      sm[r * w + c] = S.at<float>(r, c);
      gx[r * w + c] = GX.at<float>(r, c);
      gy[r * w + c] = GY.at<float>(r, c);
    }
  return 0;
}

}  // extern "C"
