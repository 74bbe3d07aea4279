// gklt_shim.cpp -- GSLAM's KLT host API on top of libgklt.so.
//
// One translation unit that a GSLAM maintainer adds to the build IN PLACE OF KLTUtil.cpp,
// pyramid.cpp and convolve.cpp, together with this directory's LK.hpp and selectGoodFeatures.hpp
// (declaration-only versions of the reference's two defining headers).  KLT.h, KLTUtil.h,
// pyramid.h, convolve.h, TrackUtil.h and error.{h,cpp} stay the reference's own files, so System,
// Frame, KeyFrame and Estimation compile unchanged: same signatures, same KLT_FeatureList layout
// (AoS, capacity 100 x n), same warn-and-fix / KLTError conventions.  Link with -lgklt.
//
// Citations are file:line under /root/reference/GSLAM/.  tests/test_shim.py compiles this file
// against small stand-ins for cv::Mat / Eigen (tests/shim_mock/) and runs it: host-only entry
// points on the CPU, computePyramid / KLTTrackFeatures / the selectors on the GPU against the oracle.
#include "KLTUtil.h"
#include "pyramid.h"
#include "error.h"
#include "LK.hpp"
#include "selectGoodFeatures.hpp"

#include <gklt.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

cv::Size gridSize(20, 20);  // selectGoodFeatures.hpp:28

namespace {

struct PyramidEntry {
  gklt_pyramid *py = nullptr;
  int w = 0, h = 0, levels = 0, ss = 0;
};

// computePyramid runs on the preload std::thread while KLTTrackFeatures runs on the main thread
// (System.cpp:508-518): the table is guarded, the GPU work itself is ordered inside libgklt.
std::mutex g_mu;
gklt_ctx *g_ctx = nullptr;
std::map<const void *, PyramidEntry> g_pyr;  // &std::vector<cv::Mat> -> device pyramid

void fatal() {  // error.cpp:21-31: print and exit(1)
  KLTError(const_cast<char *>("%s"), gklt_last_error());
  std::exit(1);
}
inline void check(int rc) {
  if (rc < 0) fatal();
}

gklt_ctx *context() {  // created on first GPU use; GKLT_DEVICE selects the GPU (default 0)
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx) {
    const char *dev = std::getenv("GKLT_DEVICE");
    check(gklt_ctx_create(dev ? std::atoi(dev) : 0, &g_ctx));
  }
  return g_ctx;
}

gklt_params to_params(KLT_TrackingContext tc) {  // the KLT.h:18-67 fields the path reads
  gklt_params p{};
  p.mindist = tc->mindist;
  p.window_width = tc->window_width;
  p.window_height = tc->window_height;
  p.min_eigenvalue = tc->min_eigenvalue;
  p.min_determinant = tc->min_determinant;
  p.min_displacement = tc->min_displacement;
  p.max_iterations = tc->max_iterations;
  p.max_residue = tc->max_residue;
  p.grad_sigma = tc->grad_sigma;
  p.smooth_sigma_fact = tc->smooth_sigma_fact;
  p.pyramid_sigma_fact = tc->pyramid_sigma_fact;
  p.step_factor = tc->step_factor;
  p.nSkippedPixels = tc->nSkippedPixels;
  p.borderx = tc->borderx;
  p.bordery = tc->bordery;
  p.nPyramidLevels = tc->nPyramidLevels;
  p.subsampling = tc->subsampling;
  return p;
}

void from_params(const gklt_params &p, KLT_TrackingContext tc) {
  tc->mindist = p.mindist;
  tc->window_width = p.window_width;
  tc->window_height = p.window_height;
  tc->min_eigenvalue = p.min_eigenvalue;
  tc->min_determinant = p.min_determinant;
  tc->min_displacement = p.min_displacement;
  tc->max_iterations = p.max_iterations;
  tc->max_residue = p.max_residue;
  tc->grad_sigma = p.grad_sigma;
  tc->smooth_sigma_fact = p.smooth_sigma_fact;
  tc->pyramid_sigma_fact = p.pyramid_sigma_fact;
  tc->step_factor = p.step_factor;
  tc->nSkippedPixels = p.nSkippedPixels;
  tc->borderx = p.borderx;
  tc->bordery = p.bordery;
  tc->nPyramidLevels = p.nPyramidLevels;
  tc->subsampling = p.subsampling;
}

// The reference corrects an even or too small window in place and prints a KLTWarning in four
// functions (KLTUtil.cpp:94-113,239-258; LK.hpp:982-1013; selectGoodFeatures.hpp:249-268).
void warn_and_fix_window(KLT_TrackingContext tc, const char *who) {
  auto fix = [&](int &v, const char *what) {
    if (v % 2 != 1) {
      v = v + 1;
      KLTWarning(const_cast<char *>("(%s) Window %s must be odd.  Changing to %d.\n"), who, what, v);
    }
    if (v < 3) {
      v = 3;
      KLTWarning(const_cast<char *>("(%s) Window %s must be at least three.  \nChanging to %d.\n"), who, what, v);
    }
  };
  fix(tc->window_width, "width");
  fix(tc->window_height, "height");
}

// System.cpp:326-344 rewrites tc every frame: the context follows it on every call
// (gklt_ctx_set_params writes nothing when the block is unchanged, so the preload thread and the
// main thread, which push the same values at the same time, do not race)
gklt_ctx *push(KLT_TrackingContext tc) {
  gklt_ctx *ctx = context();
  const gklt_params p = to_params(tc);
  check(gklt_ctx_set_params(ctx, &p));
  return ctx;
}

// dense row-major fp32 view of a CV_32F image (copies only when the rows are padded)
const float *dense_f32(const cv::Mat &m, std::vector<float> &tmp) {
  if ((size_t)m.step == sizeof(float) * (size_t)m.cols) return reinterpret_cast<const float *>(m.data);
  tmp.resize((size_t)m.rows * m.cols);
  for (int r = 0; r < m.rows; ++r)
    std::memcpy(&tmp[(size_t)r * m.cols], m.data + (size_t)r * m.step, sizeof(float) * m.cols);
  return tmp.data();
}

void store_f32(const std::vector<float> &src, cv::Mat &dst) {
  for (int r = 0; r < dst.rows; ++r)
    std::memcpy(dst.data + (size_t)r * dst.step, &src[(size_t)r * dst.cols], sizeof(float) * dst.cols);
}

gklt_pyramid *device_pyramid(const std::vector<cv::Mat> &pyramid) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_pyr.find(&pyramid);
  if (it == g_pyr.end() || !it->second.py) {
    KLTError(const_cast<char *>("(KLTTrackFeatures) pyramid %p was not produced by computePyramid"),
             (const void *)&pyramid);
    std::exit(1);
  }
  return it->second.py;
}

}  // namespace

// ------------------------------------------------------------------ KLTUtil.cpp

KLT_FeatureList KLTCreateFeatureList(KLT_TrackingContext, int nFeatures) {  // KLTUtil.cpp:49-83
  // One malloc: header | 100 n record pointers | 100 n records.  As in the reference the records'
  // non-trivial members (cv::Mat, Eigen) are not constructed and x, y, val stay uninitialised.
  const size_t cap = (size_t)gklt_feature_list_capacity(nFeatures);
  const size_t nbytes = sizeof(KLT_FeatureListRec) + cap * sizeof(KLT_Feature) + cap * sizeof(KLT_FeatureRec);
  KLT_FeatureList fl = static_cast<KLT_FeatureList>(std::malloc(nbytes));
  if (!fl) {
    KLTError(const_cast<char *>("(KLTCreateFeatureList) out of memory"));
    std::exit(1);
  }
  fl->nFeatures = nFeatures;
  fl->feature = reinterpret_cast<KLT_Feature *>(fl + 1);
  KLT_Feature first = reinterpret_cast<KLT_Feature>(fl->feature + cap);
  for (size_t i = 0; i < cap; ++i) {
    KLT_Feature f = fl->feature[i] = first + i;
    f->aff_x = -1.0f;
    f->aff_y = -1.0f;
    f->aff_img = nullptr;
    f->aff_gradx = nullptr;
    f->aff_grady = nullptr;
  }
  return fl;
}

void KLTChangeTCPyramid(KLT_TrackingContext tc, int search_range) {  // KLTUtil.cpp:87-134
  warn_and_fix_window(tc, "KLTChangeTCPyramid");
  gklt_params p = to_params(tc);
  check(gklt_change_tc_pyramid(&p, search_range, nullptr));
  tc->nPyramidLevels = p.nPyramidLevels;
  tc->subsampling = p.subsampling;
}

void KLTUpdateTCBorder(KLT_TrackingContext tc) {  // KLTUtil.cpp:225-295
  warn_and_fix_window(tc, "KLTUpdateTCBorder");
  gklt_params p = to_params(tc);
  check(gklt_update_tc_border(&p));
  tc->borderx = p.borderx;
  tc->bordery = p.bordery;
}

void computeKernels(float sigma, ConvolutionKernel *gauss, ConvolutionKernel *gaussderiv) {  // KLTUtil.cpp:136-199
  check(gklt_compute_kernels(sigma, gauss->data, &gauss->width, gaussderiv->data, &gaussderiv->width));
}

float _KLTComputeSmoothSigma(KLT_TrackingContext tc) {  // KLTUtil.cpp:215-217
  const gklt_params p = to_params(tc);
  return gklt_compute_smooth_sigma(&p);
}

KLT_TrackingContext KLTCreateTrackingContext() {  // KLTUtil.cpp:298-342, defaults :12-46
  KLT_TrackingContext tc = static_cast<KLT_TrackingContext>(std::calloc(1, sizeof(KLT_TrackingContextRec)));
  if (!tc) {
    KLTError(const_cast<char *>("(KLTCreateTrackingContext) out of memory"));
    std::exit(1);
  }
  gklt_params p;
  check(gklt_default_params(&p));  // includes KLTChangeTCPyramid(tc, 15) and KLTUpdateTCBorder(tc)
  from_params(p, tc);
  tc->sequentialMode = false;
  tc->smoothBeforeSelecting = true;
  tc->writeInternalImages = false;
  tc->lighting_insensitive = false;
  tc->affineConsistencyCheck = -1;  // KLTUtil.cpp:36-41
  tc->affine_window_width = tc->affine_window_height = 15;
  tc->affine_max_iterations = 10;
  tc->affine_max_residue = 10.0f;
  tc->affine_min_displacement = 0.02f;
  tc->affine_max_displacement_differ = 1.5f;
  return tc;
}

// ------------------------------------------------------------------ pyramid.cpp / convolve.cpp

// pyramid.cpp:45-84.  The planes stay on the GPU; `pyramid` gets its 3 * nPyramidLevels entries and
// the ONE plane the callers read on the host, the coarsest image pyramid[3 * (L - 1)]
// (System.cpp:407,453,468-471,672), is copied back (130 KB at 1080p / 4 levels).  Like the
// reference the call returns when the pyramid is complete; it is meant to run on the preload
// thread (System.cpp:370-373,508-511).  Define GKLT_SHIM_ALL_PLANES to materialise every plane.
void computePyramid(KLT_TrackingContext tc, const cv::Mat &img, std::vector<cv::Mat> &pyramid) {
  // GKLT_SHIM_BREAKDOWN=1: mean time of the call's three parts on stderr, every 64 calls
  static const bool breakdown = std::getenv("GKLT_SHIM_BREAKDOWN") != nullptr;
  static double t_parts[3] = {0, 0, 0};
  static int t_calls = 0;
  const auto q0 = std::chrono::steady_clock::now();
  gklt_ctx *ctx = push(tc);
  PyramidEntry e;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    e = g_pyr[&pyramid];
  }
  if (!e.py || e.w != img.cols || e.h != img.rows || e.levels != tc->nPyramidLevels || e.ss != tc->subsampling) {
    if (e.py) check(gklt_pyramid_destroy(e.py));
    e = PyramidEntry{nullptr, img.cols, img.rows, tc->nPyramidLevels, tc->subsampling};
    check(gklt_pyramid_create(ctx, img.cols, img.rows, &e.py));
    std::lock_guard<std::mutex> lk(g_mu);
    g_pyr[&pyramid] = e;
  }
  const auto q1 = std::chrono::steady_clock::now();
  check(gklt_pyramid_build(ctx, e.py, img.data, (int)img.step, /*img_on_device=*/0));
  const auto q2 = std::chrono::steady_clock::now();
  pyramid.resize(3 * (size_t)tc->nPyramidLevels);
  for (int i = 0; i < 3 * tc->nPyramidLevels; ++i) {
#ifndef GKLT_SHIM_ALL_PLANES
    if (i != 3 * (tc->nPyramidLevels - 1)) continue;
#endif
    int w = 0, h = 0;
    check(gklt_pyramid_level_size(e.py, i / 3, &w, &h));
    pyramid[i].create(h, w, CV_32F);
    check(gklt_pyramid_download(ctx, e.py, i / 3, i % 3, reinterpret_cast<float *>(pyramid[i].data),
                                (int)(pyramid[i].step / sizeof(float))));
  }
  if (breakdown) {
    const auto q3 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::micro>(b - a).count();
    };
    t_parts[0] += us(q0, q1); t_parts[1] += us(q1, q2); t_parts[2] += us(q2, q3);
    if (++t_calls % 64 == 0) {
      std::fprintf(stderr, "computePyramid: set-up %.1f us, gklt_pyramid_build (enqueue) %.1f us, plane download (waits for the build) %.1f us\n",
                   t_parts[0] / 64, t_parts[1] / 64, t_parts[2] / 64);
      t_parts[0] = t_parts[1] = t_parts[2] = 0;
    }
  }
}

void computeSmoothedImage(const cv::Mat &img, cv::Mat &dst, float sigma) {  // convolve.cpp:46-56
  std::vector<float> in, out((size_t)img.rows * img.cols);
  check(gklt_smooth(context(), dense_f32(img, in), img.cols, img.rows, sigma, out.data()));
  dst.create(img.rows, img.cols, CV_32F);
  store_f32(out, dst);
}

void computeGradients(const cv::Mat &img, cv::Mat &gradx, cv::Mat &grady, float sigma, std::thread *) {  // convolve.cpp:19-42
  std::vector<float> in, gx((size_t)img.rows * img.cols), gy(gx.size());
  check(gklt_gradients(context(), dense_f32(img, in), img.cols, img.rows, sigma, gx.data(), gy.data()));
  gradx.create(img.rows, img.cols, CV_32F);
  grady.create(img.rows, img.cols, CV_32F);
  store_f32(gx, gradx);
  store_f32(gy, grady);
}

// ------------------------------------------------------------------ LK.hpp

int KLTCountRemainingFeatures(KLT_FeatureList fl) {  // LK.hpp:27-38
  int count = 0;
  for (int i = 0; i < fl->nFeatures; ++i) count += fl->feature[i]->val >= 0;
  return count;
}

void KLTTrackFeatures(KLT_TrackingContext tc, std::vector<cv::Mat> &pyramid1, std::vector<cv::Mat> &pyramid2,
                      KLT_FeatureList fl, const Eigen::Matrix3d &invK) {  // LK.hpp:972-1046
  warn_and_fix_window(tc, "KLTTrackFeatures");  // LK.hpp:982-1013
  gklt_ctx *ctx = push(tc);
  const int n = fl->nFeatures;
  // SoA images of the list, kept from call to call (the list has the same size every frame)
  static thread_local std::vector<float> x, y;
  static thread_local std::vector<int> v;
  static thread_local std::vector<double> nrm, vec;
  if ((int)x.size() < n) {
    x.resize(n); y.resize(n); v.resize(n);
    nrm.resize(3 * (size_t)n); vec.resize(3 * (size_t)n);
  }
  for (int i = 0; i < n; ++i) {  // AoS -> SoA; index i is the feature's identity and never moves
    const KLT_Feature f = fl->feature[i];
    x[i] = f->x;
    y[i] = f->y;
    v[i] = f->val;
  }
  double K[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K[3 * r + c] = invK(r, c);
  check(gklt_track(ctx, device_pyramid(pyramid1), device_pyramid(pyramid2), n, x.data(), y.data(), v.data(), K,
                   nrm.data(), vec.data()));
  for (int i = 0; i < n; ++i) {
    const KLT_Feature f = fl->feature[i];
    f->x = x[i];
    f->y = y[i];
    f->val = v[i];
    if (v[i] == KLT_TRACKED) {  // LK.hpp:1019-1035
      f->norm = Eigen::Vector3d(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
      f->vec = Eigen::Vector3d(vec[3 * i], vec[3 * i + 1], vec[3 * i + 2]);
    }
  }
  fl->isOutlierRejected = false;  // LK.hpp:1036
}

// ------------------------------------------------------------------ selectGoodFeatures.hpp

namespace {
void select(KLT_TrackingContext tc, const cv::Mat &sxx, const cv::Mat &syy, const cv::Mat &sxy, KLT_FeatureList fl,
            int mode, int variant, const char *who) {
  warn_and_fix_window(tc, who);  // selectGoodFeatures.hpp:249-268
  gklt_ctx *ctx = push(tc);
  // the sums are (H+1) x (W+1) CV_32F images from cv::integral (System.cpp:389-392)
  const int W = sxx.cols - 1, H = sxx.rows - 1;
  int n = fl->nFeatures;
  // the Uniform selector may grow the list (selectGoodFeatures.hpp:1257); KLTCreateFeatureList
  // allocated 100 x the size it was created with, which the list itself does not record: the shim
  // lets it grow up to 100 x its current size and relies on the caller's allocation as the reference does
  const int cap = gklt_feature_list_capacity(std::max(n, 1));
  std::vector<float> x(cap, -1.0f), y(cap, -1.0f), ta, tb, tc2;
  std::vector<int> v(cap, KLT_NOT_FOUND);
  for (int i = 0; i < n; ++i) {
    x[i] = fl->feature[i]->x;
    y[i] = fl->feature[i]->y;
    v[i] = fl->feature[i]->val;
  }
  const int n0 = n;
  check(gklt_select(ctx, mode, variant, dense_f32(sxx, ta), dense_f32(syy, tb), dense_f32(sxy, tc2), W, H, &n, cap,
                    x.data(), y.data(), v.data(), gridSize.width, gridSize.height));
  fl->nFeatures = n;
  for (int i = 0; i < n; ++i) {
    const KLT_Feature f = fl->feature[i];
    // slots the selector writes: all of them when selecting, the lost ones when replacing, and
    // the ones the Uniform flavour appended
    const bool written = mode == GKLT_SELECT_ALL || i >= n0 || f->val < 0;
    f->x = x[i];
    f->y = y[i];
    f->val = v[i];
    // KLTReplaceLostFeatures2 (rounding body, :1020) and KLTSelectGoodFeaturesUniform reset the
    // affine state of every slot they write (selectGoodFeatures.hpp:562-567,602-607,1263-1268,
    // 1280-1285,1304-1309); the plain body, which KLTSelectGoodFeatures2 (:1005) and
    // KLTReplaceLostFeaturesUniform (:1353) also run, has these lines commented out (:140-148,182-190)
    const bool resets = (mode == GKLT_REPLACE_SOME && variant == GKLT_VARIANT_ROUND2) ||
                        (mode == GKLT_SELECT_ALL && variant == GKLT_VARIANT_UNIFORM);
    if (written && resets) {
      f->aff_x = -1.0f;
      f->aff_y = -1.0f;
      f->aff_Axx = 1.0f;
      f->aff_Ayx = 0.0f;
      f->aff_Axy = 0.0f;
      f->aff_Ayy = 1.0f;
    }
  }
}
}  // namespace

void KLTSelectGoodFeatures(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :424-433
  select(tc, a, b, c, fl, GKLT_SELECT_ALL, GKLT_VARIANT_PLAIN, "KLTSelectGoodFeatures");
}
void KLTReplaceLostFeatures(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :436-449
  if (fl->nFeatures - KLTCountRemainingFeatures(fl) > 0)
    select(tc, a, b, c, fl, GKLT_REPLACE_SOME, GKLT_VARIANT_PLAIN, "KLTReplaceLostFeatures");
}
void KLTSelectGoodFeatures2(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :997-1006
  select(tc, a, b, c, fl, GKLT_SELECT_ALL, GKLT_VARIANT_ROUND2, "KLTSelectGoodFeatures2");
}
void KLTReplaceLostFeatures2(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :1009-1022
  if (fl->nFeatures - KLTCountRemainingFeatures(fl) > 0)
    select(tc, a, b, c, fl, GKLT_REPLACE_SOME, GKLT_VARIANT_ROUND2, "KLTReplaceLostFeatures2");
}
void KLTSelectGoodFeaturesUniform(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :1318-1338
  select(tc, a, b, c, fl, GKLT_SELECT_ALL, GKLT_VARIANT_UNIFORM, "KLTSelectGoodFeaturesUniform");
}
void KLTReplaceLostFeaturesUniform(KLT_TrackingContext tc, const cv::Mat &a, const cv::Mat &b, const cv::Mat &c, KLT_FeatureList fl) {  // :1341-1355 (runs the plain replace)
  if (fl->nFeatures - KLTCountRemainingFeatures(fl) > 0)
    select(tc, a, b, c, fl, GKLT_REPLACE_SOME, GKLT_VARIANT_UNIFORM, "KLTReplaceLostFeaturesUniform");
}
