/*
 * klt_oracle.h -- CPU restatement of GSLAM's KLT front end (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle and the timed CPU baseline.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be built here (OpenCV, Eigen, TBB absent; LK.hpp
 * includes <avxintrin.h> directly).  The tracker / context / selector arithmetic below follows
 * the reference source line by line (citations at each function); the separable filter, the
 * 2x resize and the integral image restate the documented semantics of the OpenCV calls the
 * reference makes (cv::sepFilter2D, cv::resize, cv::integral), whose library is not vendored
 * and whose version is not pinned by the reference.
 */
#ifndef KLT_ORACLE_H
#define KLT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OK_MAX_KERNEL_WIDTH 71 /* KLT.h:107 */

/* KLT.h:69-74 */
#define OK_KLT_TRACKED 0
#define OK_KLT_NOT_FOUND -1
#define OK_KLT_SMALL_DET -2
#define OK_KLT_MAX_ITERATIONS -3
#define OK_KLT_OOB -4
#define OK_KLT_LARGE_RESIDUE -5

/* the KLT_TrackingContextRec fields the hot path reads (KLT.h:18-67) */
typedef struct {
  int mindist;
  int window_width, window_height;
  int min_eigenvalue;
  float min_determinant;
  float min_displacement;
  int max_iterations;
  float max_residue;
  float grad_sigma;
  float smooth_sigma_fact;
  float pyramid_sigma_fact;
  float step_factor;
  int nSkippedPixels;
  int borderx, bordery;
  int nPyramidLevels;
  int subsampling;
} ok_params;

enum { OK_SELECT_ALL = 0, OK_REPLACE_SOME = 1 };
enum { OK_VARIANT_PLAIN = 0, OK_VARIANT_ROUND2 = 1, OK_VARIANT_UNIFORM = 2 };

/* KLTUtil.cpp:12-46,298-342 (defaults, then KLTChangeTCPyramid(15) and KLTUpdateTCBorder) */
void ok_default_params(ok_params *p);
/* KLTUtil.cpp:87-134; returns number of warn-and-fix corrections applied to the window */
int ok_change_tc_pyramid(ok_params *p, int search_range);
/* KLTUtil.cpp:225-295; returns <0 if a kernel would be too wide */
int ok_update_tc_border(ok_params *p);
/* KLTUtil.cpp:136-199; returns 0, or -1 where the reference calls KLTError (width stays 71) */
int ok_compute_kernels(float sigma, float *gauss, int *gauss_width, float *deriv,
                       int *deriv_width);
/* KLTUtil.cpp:215-217 */
float ok_smooth_sigma(const ok_params *p);

/* cv::sepFilter2D restatement (convolve.cpp:16,36,37,55): correlation, centre anchor,
 * BORDER_REFLECT_101, fp32 intermediate.  sym = +1 symmetric taps, -1 antisymmetric. */
void ok_sep_filter(const float *src, int W, int H, const float *rowK, int rw, int rsym,
                   const float *colK, int cw, int csym, float *dst);
/* convolve.cpp:46-56 */
int ok_smooth(const float *src, int W, int H, float sigma, float *dst);
/* convolve.cpp:19-42 */
int ok_gradients(const float *src, int W, int H, float sigma, float *gx, float *gy);
/* cv::resize INTER_LINEAR restatement (pyramid.cpp:65) */
void ok_resize_linear(const float *src, int W, int H, float *dst, int W2, int H2);

/* level geometry: W_l = W_{l-1}/ss (pyramid.cpp:64) */
void ok_level_size(const ok_params *p, int W, int H, int level, int *Wl, int *Hl);
/* pyramid.cpp:45-84.  planes[3*l+{0,1,2}] = I, gx, gy of level l, dense W_l x H_l fp32,
 * caller-allocated. */
int ok_compute_pyramid(const ok_params *p, const uint8_t *img, int W, int H, int stride,
                       float **planes);

/* LK.hpp:972-1046 (+ everything under it).  x,y,val updated in place; norm3/vec3 (n*3
 * doubles each, may be NULL) written for TRACKED features only.  iters_out (may be NULL)
 * receives the total Newton iterations spent on each feature (measurement aid). */
int ok_track_features(ok_params *p, int W, int H, float *const *pyr1, float *const *pyr2, int n,
                      float *x, float *y, int *val, const double *invK, double *norm3,
                      double *vec3, int *iters_out);
/* LK.hpp:27-38 */
int ok_count_remaining(int n, const int *val);
/* ImageGrids::rotateAndNormalizePoint + NormalizeFrameInvoker (ImageGrids.hpp:73-106,
 * System.cpp:246-267): bearing vectors of the tracked features with per-row rotations. */
int ok_rotate_normalize(int n, const float *x, const float *y, const int *val, int grid_height,
                        int nrot, const double *rotations, const double *invK, double *norm3,
                        double *vec3);

/* System.cpp:385-392: gxx=gx*gx ... then cv::integral(.., CV_32F) -> (H+1)x(W+1) */
void ok_integral_products(const float *gx, const float *gy, int W, int H, float *sumxx,
                          float *sumyy, float *sumxy);

/* selectGoodFeatures.hpp:234-449 (plain), :620-806/:1009-1022 (ROUND2), :1064-1355 (UNIFORM).
 * sums are (H+1)x(W+1) dense fp32.  *nFeatures may grow (UNIFORM) up to capacity. */
int ok_select_features(ok_params *p, int mode, int variant, const float *sumxx,
                       const float *sumyy, const float *sumxy, int W, int H, int *nFeatures,
                       int capacity, float *x, float *y, int *val, int gridW, int gridH);

/* SURVEY section 7's selector fast path (no counterpart in the reference): window sums formed
 * directly.  ok_box_products: H x W planes of window sums in a defined order; ok_set_box_sums:
 * while set (non-NULL), ok_select_features scores from these planes and ignores its sum images. */
void ok_box_products(const float *gx, const float *gy, int W, int H, int hw, int hh, float *bxx,
                     float *byy, float *bxy);
void ok_set_box_sums(const float *bxx, const float *byy, const float *bxy);

void ok_set_threads(int n);
int ok_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
