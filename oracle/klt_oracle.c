/*
 * klt_oracle.c -- CPU restatement of GSLAM's KLT front end.  TEST INFRASTRUCTURE ONLY
 * (see klt_oracle.h: parity unpinned, may be loaded only by tests/, smoke() and bench.py's
 * cpu_baseline leg).  Build: oracle/Makefile (-O2 -ffp-contract=off: the reference's Release
 * build is AVX without FMA, project.pbxproj:492,496, so every mul+add is unfused fp32).
 *
 * All file:line citations are relative to /root/reference/GSLAM/.
 */
#include "klt_oracle.h"
#include "../include/gklt_order.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void ok_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int ok_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ context / kernels */

/* warn-and-fix of the window size, KLTUtil.cpp:94-113 (same block at :239-258, LK.hpp:982-1013,
 * selectGoodFeatures.hpp:249-268) */
static int fix_window(ok_params *p) {
  int fixes = 0;
  if (p->window_width % 2 != 1) { p->window_width++; fixes++; }
  if (p->window_height % 2 != 1) { p->window_height++; fixes++; }
  if (p->window_width < 3) { p->window_width = 3; fixes++; }
  if (p->window_height < 3) { p->window_height = 3; fixes++; }
  return fixes;
}

int ok_change_tc_pyramid(ok_params *p, int search_range) { /* KLTUtil.cpp:87-134 */
  int fixes = fix_window(p);
  int mn = p->window_width < p->window_height ? p->window_width : p->window_height;
  float window_halfwidth = mn / 2.0f;
  float subsampling = ((float)search_range) / window_halfwidth;
  if (subsampling < 1.0) {
    p->nPyramidLevels = 1;
  } else if (subsampling <= 3.0) {
    p->nPyramidLevels = 2; p->subsampling = 2;
  } else if (subsampling <= 5.0) {
    p->nPyramidLevels = 2; p->subsampling = 4;
  } else if (subsampling <= 9.0) {
    p->nPyramidLevels = 2; p->subsampling = 8;
  } else {
    float val = (float)(log(7.0 * subsampling + 1.0) / log(8.0));
    p->nPyramidLevels = (int)(val + 0.99);
    p->subsampling = 8;
  }
  return fixes;
}

int ok_compute_kernels(float sigma, float *gauss, int *gauss_width, float *deriv,
                       int *deriv_width) { /* KLTUtil.cpp:136-199 */
  const float factor = 0.01f;
  float g[OK_MAX_KERNEL_WIDTH], d[OK_MAX_KERNEL_WIDTH];
  int i, gw, dw;
  {
    const int hw = OK_MAX_KERNEL_WIDTH / 2;
    /* :149  exp(float) resolves to the float overload in C++ */
    float max_gauss = 1.0f, max_gaussderiv = (float)(sigma * expf(-0.5f));
    for (i = -hw; i <= hw; i++) { /* :152-155 */
      g[i + hw] = (float)expf((float)(-i * i) / (2 * sigma * sigma));
      d[i + hw] = -i * g[i + hw];
    }
    gw = OK_MAX_KERNEL_WIDTH; /* :158-163 */
    for (i = -hw; i <= hw && fabsf(g[i + hw] / max_gauss) < factor; i++, gw -= 2) {}
    dw = OK_MAX_KERNEL_WIDTH;
    for (i = -hw; i <= hw && fabsf(d[i + hw] / max_gaussderiv) < factor; i++, dw -= 2) {}
    if (gw == OK_MAX_KERNEL_WIDTH || dw == OK_MAX_KERNEL_WIDTH || gw < 1 || dw < 1)
      return -1; /* :164-167 KLTError */
  }
  for (i = 0; i < gw; i++) g[i] = g[i + (OK_MAX_KERNEL_WIDTH - gw) / 2]; /* :172-176 */
  for (i = 0; i < dw; i++) d[i] = d[i + (OK_MAX_KERNEL_WIDTH - dw) / 2];
  { /* :180-197 */
    const int hw = dw / 2;
    float den = 0.0f;
    for (i = 0; i < gw; i++) den += g[i];
    for (i = 0; i < gw; i++) g[i] /= den;
    den = 0.0f;
    for (i = -hw; i <= hw; i++) den += i * d[i + hw];
    for (i = -hw; i <= hw; i++) d[i + hw] /= den;
  }
  if (gauss) memcpy(gauss, g, gw * sizeof(float));
  if (deriv) memcpy(deriv, d, dw * sizeof(float));
  if (gauss_width) *gauss_width = gw;
  if (deriv_width) *deriv_width = dw;
  return 0;
}

float ok_smooth_sigma(const ok_params *p) { /* KLTUtil.cpp:215-217 */
  int mx = p->window_width > p->window_height ? p->window_width : p->window_height;
  return p->smooth_sigma_fact * mx;
}
static float pyramid_sigma(const ok_params *p) { /* KLTUtil.cpp:219-221, pyramid.cpp:56 */
  return p->pyramid_sigma_fact * p->subsampling;
}

int ok_update_tc_border(ok_params *p) { /* KLTUtil.cpp:225-295 */
  int gw, dw, i;
  fix_window(p);
  int window_hw = (p->window_width > p->window_height ? p->window_width : p->window_height) / 2;
  if (ok_compute_kernels(ok_smooth_sigma(p), NULL, &gw, NULL, &dw)) return -1;
  int smooth_gauss_hw = gw / 2;
  if (ok_compute_kernels(pyramid_sigma(p), NULL, &gw, NULL, &dw)) return -1;
  int pyramid_gauss_hw = gw / 2;
  int ss = p->subsampling;
  int n_invalid_pixels = smooth_gauss_hw;
  for (i = 1; i < p->nPyramidLevels; i++) {
    float val = ((float)n_invalid_pixels + pyramid_gauss_hw) / ss;
    n_invalid_pixels = (int)(val + 0.99);
  }
  int ss_power = 1;
  for (i = 1; i < p->nPyramidLevels; i++) ss_power *= ss;
  int border = (n_invalid_pixels + window_hw) * ss_power;
  p->borderx = border;
  p->bordery = border;
  return 0;
}

void ok_default_params(ok_params *p) { /* KLTUtil.cpp:12-46,298-342 */
  memset(p, 0, sizeof(*p));
  p->mindist = 10;
  p->window_width = 21;
  p->window_height = 21;
  p->min_eigenvalue = 200;
  p->min_determinant = 0.01f;
  p->min_displacement = 0.1f;
  p->max_iterations = 10;
  p->max_residue = 10.0f;
  p->grad_sigma = 1.0f;
  p->smooth_sigma_fact = 0.1f;
  p->pyramid_sigma_fact = 0.9f;
  p->step_factor = 1.0f;
  p->nSkippedPixels = 0;
  p->subsampling = 2; /* the struct is malloc'd and only the <=9 branches set it; see SURVEY F6 */
  ok_change_tc_pyramid(p, 15);
  ok_update_tc_border(p);
}

/* ------------------------------------------------------------------ separable filter */

static inline int reflect101(int i, int n) { /* cv::BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * n - 2 - i;
  }
  return i;
}

/* One 1-D correlation at the centre of a (2r+1)-tap window whose samples are s[-r..r].
 * Summation order (restated; unpinned against OpenCV): symmetric taps  k0*s0 + sum_j kj*(s[+j]+s[-j]),
 * antisymmetric taps  sum_j kj*(s[+j]-s[-j]),  j = 1..r, every product and add rounded to fp32. */
static inline float corr1d(const float *s, int stride, const float *k, int r, int sym) {
  float acc;
  int j;
  if (sym > 0) {
    acc = k[r] * s[0];
    for (j = 1; j <= r; j++) acc = acc + k[r + j] * (s[j * stride] + s[-j * stride]);
  } else {
    acc = 0.0f;
    for (j = 1; j <= r; j++) acc = acc + k[r + j] * (s[j * stride] - s[-j * stride]);
  }
  return acc;
}

/* One filtered row: out[x] = correlation of the taps with s_j[x], j = -r..r, where sp[j] / sm[j]
 * point at the samples at offset +j / -j.  The loops run tap-outer, pixel-inner, so that the
 * compiler can vectorise over x; every pixel still sees k0*s0, then + k1*(s1 +- s-1), ... in
 * this order, i.e. the same rounded operations as the tap-inner form. */
static void corr_row(float *restrict out, int n, const float *const *sp, const float *const *sm,
                     const float *restrict s0, const float *K, int r, int sym) {
  int x, j;
  if (sym > 0) {
    const float k0 = K[r];
    for (x = 0; x < n; x++) out[x] = k0 * s0[x];
    for (j = 1; j <= r; j++) {
      const float kj = K[r + j];
      const float *restrict a = sp[j], *restrict b = sm[j];
      for (x = 0; x < n; x++) out[x] = out[x] + kj * (a[x] + b[x]);
    }
  } else {
    for (x = 0; x < n; x++) out[x] = 0.0f;
    for (j = 1; j <= r; j++) {
      const float kj = K[r + j];
      const float *restrict a = sp[j], *restrict b = sm[j];
      for (x = 0; x < n; x++) out[x] = out[x] + kj * (a[x] - b[x]);
    }
  }
}

void ok_sep_filter(const float *src, int W, int H, const float *rowK, int rw, int rsym,
                   const float *colK, int cw, int csym, float *dst) {
  const int rr = rw / 2, cr = cw / 2;
  float *tmp = (float *)malloc((size_t)W * H * sizeof(float));
  int y;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (y = 0; y < H; y++) {
    float *ext = (float *)malloc((size_t)(W + 2 * rr) * sizeof(float));
    const float *srow = src + (size_t)y * W;
    const float *sp[35 + 1], *sm[35 + 1];
    int x, j;
    for (x = -rr; x < W + rr; x++) ext[x + rr] = srow[reflect101(x, W)];
    for (j = 1; j <= rr; j++) { sp[j] = ext + rr + j; sm[j] = ext + rr - j; }
    corr_row(tmp + (size_t)y * W, W, sp, sm, ext + rr, rowK, rr, rsym);
    free(ext);
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (y = 0; y < H; y++) {
    const float *sp[35 + 1], *sm[35 + 1];
    int j;
    for (j = 1; j <= cr; j++) {
      sp[j] = tmp + (size_t)reflect101(y + j, H) * W;
      sm[j] = tmp + (size_t)reflect101(y - j, H) * W;
    }
    corr_row(dst + (size_t)y * W, W, sp, sm, tmp + (size_t)y * W, colK, cr, csym);
  }
  free(tmp);
}

int ok_smooth(const float *src, int W, int H, float sigma, float *dst) { /* convolve.cpp:46-56 */
  float g[OK_MAX_KERNEL_WIDTH], d[OK_MAX_KERNEL_WIDTH];
  int gw, dw;
  if (ok_compute_kernels(sigma, g, &gw, d, &dw)) return -1;
  ok_sep_filter(src, W, H, g, gw, +1, g, gw, +1, dst);
  return 0;
}

int ok_gradients(const float *src, int W, int H, float sigma, float *gx, float *gy) {
  /* convolve.cpp:19-42 / pyramid.cpp:69-79: gx = rows G', cols G ; gy = rows G, cols G' */
  float g[OK_MAX_KERNEL_WIDTH], d[OK_MAX_KERNEL_WIDTH];
  int gw, dw;
  if (ok_compute_kernels(sigma, g, &gw, d, &dw)) return -1;
  ok_sep_filter(src, W, H, d, dw, -1, g, gw, +1, gx);
  ok_sep_filter(src, W, H, g, gw, +1, d, dw, -1, gy);
  return 0;
}

void ok_resize_linear(const float *src, int W, int H, float *dst, int W2, int H2) {
  /* cv::resize(..., INTER_LINEAR) as called at pyramid.cpp:65.  Exact 2x decimation takes
   * OpenCV's area-fast path = mean of the 2x2 block; otherwise bilinear at
   * sx = (dx+0.5)*(W/W2)-0.5 with the source index clamped. */
  int y;
  if (W == 2 * W2 && H == 2 * H2) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (y = 0; y < H2; y++) {
      const float *r0 = src + (size_t)(2 * y) * W, *r1 = r0 + W;
      int x;
      for (x = 0; x < W2; x++)
        dst[(size_t)y * W2 + x] = GKLT_MEAN2X2(r0[2 * x], r0[2 * x + 1], r1[2 * x], r1[2 * x + 1]);
    }
    return;
  }
  {
    double scale_x = (double)W / W2, scale_y = (double)H / H2;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (y = 0; y < H2; y++) {
      float fy = (float)((y + 0.5) * scale_y - 0.5);
      int sy = (int)floorf(fy);
      fy -= sy;
      if (sy < 0) { sy = 0; fy = 0.f; }
      if (sy >= H - 1) { sy = H - 1; fy = 0.f; }
      int sy1 = sy + 1 < H ? sy + 1 : H - 1;
      int x;
      for (x = 0; x < W2; x++) {
        float fx = (float)((x + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= W - 1) { sx = W - 1; fx = 0.f; }
        int sx1 = sx + 1 < W ? sx + 1 : W - 1;
        float top = src[(size_t)sy * W + sx] * (1.f - fx) + src[(size_t)sy * W + sx1] * fx;
        float bot = src[(size_t)sy1 * W + sx] * (1.f - fx) + src[(size_t)sy1 * W + sx1] * fx;
        dst[(size_t)y * W2 + x] = top * (1.f - fy) + bot * fy;
      }
    }
  }
}

void ok_level_size(const ok_params *p, int W, int H, int level, int *Wl, int *Hl) {
  int l;
  for (l = 0; l < level; l++) { W /= p->subsampling; H /= p->subsampling; } /* pyramid.cpp:64 */
  *Wl = W; *Hl = H;
}

int ok_compute_pyramid(const ok_params *p, const uint8_t *img, int W, int H, int stride,
                       float **planes) { /* pyramid.cpp:45-84 */
  int L = p->nPyramidLevels, l, y, x;
  float *tmp = (float *)malloc((size_t)W * H * sizeof(float));
  for (y = 0; y < H; y++) /* :59 convertTo(CV_32F) */
    for (x = 0; x < W; x++) tmp[(size_t)y * W + x] = (float)img[(size_t)y * stride + x];
  if (ok_smooth(tmp, W, H, ok_smooth_sigma(p), planes[0])) { free(tmp); return -1; } /* :61 */
  int w = W, h = H;
  for (l = 1; l < L; l++) { /* :62-66 */
    if (ok_smooth(planes[3 * l - 3], w, h, pyramid_sigma(p), tmp)) { free(tmp); return -1; }
    int w2 = w / p->subsampling, h2 = h / p->subsampling;
    ok_resize_linear(tmp, w, h, planes[3 * l], w2, h2);
    w = w2; h = h2;
  }
  w = W; h = H;
  for (l = 0; l < L; l++) { /* :69-83 */
    if (ok_gradients(planes[3 * l], w, h, p->grad_sigma, planes[3 * l + 1], planes[3 * l + 2])) {
      free(tmp); return -1;
    }
    w /= p->subsampling; h /= p->subsampling;
  }
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ tracker */

#define SIMD_STEP 8 /* LK.hpp:40 */

static void bilinear_weight(float x, float y, int *xt, int *yt, float w[4]) { /* LK.hpp:56-68 */
  *xt = (int)x;
  *yt = (int)y;
  float ax = x - *xt;
  float ay = y - *yt;
  w[0] = (float)((1.0 - ax) * (1.0 - ay));
  w[1] = (float)((ax) * (1.0 - ay));
  w[2] = (float)((1.0 - ax) * ay);
  w[3] = ax * ay; /* :67 is a float*float product, the other three are evaluated in double */
}

/* LK.hpp:71-108.  The 8-wide stores that overrun each row are overwritten by the next row and the
 * tail is zeroed, so the result is the w*h patch followed by zeros up to alignedPatchSize. */
static void bilinear_patch(float *patch, const float *img, int nc, int xt, int yt, int width,
                           int height, int aligned, const float w[4]) {
  int hw = width / 2, hh = height / 2, h, y, c;
  if (w[0] == 1.0) { /* :80-85 integer position: memcpy rows (tail keeps its zeros) */
    for (h = -hh, y = 0; h <= hh; h++, y++)
      memcpy(&patch[y * width], &img[(size_t)(yt + h) * nc + (xt - hw)], width * sizeof(float));
    return;
  }
  for (h = -hh, y = 0; h <= hh; h++, y++) {
    const float *p0 = img + (size_t)(h + yt) * nc, *p1 = img + (size_t)(h + yt + 1) * nc;
    for (c = 0; c < width; c++) { /* :98-103 */
      int xx = xt - hw + c;
      float a = p0[xx] * w[0], b = p0[xx + 1] * w[1], cc = p1[xx] * w[2], d = p1[xx + 1] * w[3];
      patch[y * width + c] = (a + b) + (cc + d);
    }
  }
  memset(&patch[width * height], 0, (aligned - width * height) * sizeof(float)); /* :107 */
}

static inline float lanes_sum(const float a[SIMD_STEP]) { /* LK.hpp:143,163,207 */
  return a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}

static int track_feature(float x1, float y1, float *x2, float *y2, const float *img1,
                         const float *gradx1, const float *grady1, const float *img2,
                         const float *gradx2, const float *grady2, int nc, int nr, int width,
                         int height, float step_factor, int max_iterations, float small, float th,
                         float max_residue, float *buffers, int *iters) { /* LK.hpp:413-669 */
  float gxx, gxy, gyy, ex, ey, dx = 0, dy = 0;
  int iteration = 0, status = OK_KLT_TRACKED, i, k;
  int hw = width / 2, hh = height / 2;
  float one_plus_eps = 1.001f;
  float weights1[4], weights2[4];
  int aligned = SIMD_STEP * ((width * height + SIMD_STEP) / SIMD_STEP); /* :448 */
  int nloop = ((width * height + SIMD_STEP - 1) / SIMD_STEP) * SIMD_STEP; /* i<w*h; i+=8 */
  memset(buffers, 0, 9 * aligned * sizeof(float)); /* :452 */
  float *patch1 = buffers, *gx1 = patch1 + aligned, *gy1 = gx1 + aligned;
  float *patch2 = gy1 + aligned, *gx2 = patch2 + aligned, *gy2 = gx2 + aligned;
  float *patchdiff = gy2 + aligned, *gx = patchdiff + aligned, *gy = gx + aligned;
  int xt1, yt1, xt2, yt2;
  int have1 = 0;

  do {
    if (x1 - hw < 0.0f || nc - (x1 + hw) < one_plus_eps || *x2 - hw < 0.0f ||
        nc - (*x2 + hw) < one_plus_eps || y1 - hh < 0.0f || nr - (y1 + hh) < one_plus_eps ||
        *y2 - hh < 0.0f || nr - (*y2 + hh) < one_plus_eps) { /* :558-564 */
      status = OK_KLT_OOB;
      break;
    }
    if (!have1) {
      /* :468-476 samples the template BEFORE the bounds test (reading outside the image when
       * x1 is out of bounds); the samples are only used after the test passed, so sampling
       * here gives the same result without the out-of-bounds read. */
      bilinear_weight(x1, y1, &xt1, &yt1, weights1);
      bilinear_patch(patch1, img1, nc, xt1, yt1, width, height, aligned, weights1);
      bilinear_patch(gx1, gradx1, nc, xt1, yt1, width, height, aligned, weights1);
      bilinear_patch(gy1, grady1, nc, xt1, yt1, width, height, aligned, weights1);
      have1 = 1;
    }
    bilinear_weight(*x2, *y2, &xt2, &yt2, weights2); /* :568-572 */
    bilinear_patch(patch2, img2, nc, xt2, yt2, width, height, aligned, weights2);
    bilinear_patch(gx2, gradx2, nc, xt2, yt2, width, height, aligned, weights2);
    bilinear_patch(gy2, grady2, nc, xt2, yt2, width, height, aligned, weights2);
    for (i = 0; i < nloop; i++) { /* :575-577 */
      patchdiff[i] = patch1[i] - patch2[i];
      gx[i] = gx1[i] + gx2[i];
      gy[i] = gy1[i] + gy2[i];
    }
    { /* :127-166: 8 lane partial sums, lanes then added left to right */
      float axx[SIMD_STEP] = {0}, ayy[SIMD_STEP] = {0}, axy[SIMD_STEP] = {0};
      float aex[SIMD_STEP] = {0}, aey[SIMD_STEP] = {0};
      for (i = 0; i < nloop; i += SIMD_STEP)
        for (k = 0; k < SIMD_STEP; k++) {
          axx[k] = axx[k] + gx[i + k] * gx[i + k];
          ayy[k] = ayy[k] + gy[i + k] * gy[i + k];
          axy[k] = axy[k] + gx[i + k] * gy[i + k];
        }
      for (i = 0; i < aligned; i += SIMD_STEP)
        for (k = 0; k < SIMD_STEP; k++) {
          aex[k] = aex[k] + gx[i + k] * patchdiff[i + k];
          aey[k] = aey[k] + gy[i + k] * patchdiff[i + k];
        }
      gxx = lanes_sum(axx); gyy = lanes_sum(ayy); gxy = lanes_sum(axy);
      ex = lanes_sum(aex) * step_factor;
      ey = lanes_sum(aey) * step_factor;
    }
    { /* :168-183 */
      float det = gxx * gyy - gxy * gxy;
      if (det < small) { status = OK_KLT_SMALL_DET; break; }
      dx = (gyy * ex - gxy * ey) / det;
      dy = (gxx * ey - gxy * ex) / det;
      status = OK_KLT_TRACKED;
    }
    *x2 += dx;
    *y2 += dy;
    iteration++;
  } while ((fabsf(dx) >= th || fabsf(dy) >= th) && iteration < max_iterations); /* :615 */

  if (*x2 - hw < 0.0f || nc - (*x2 + hw) < one_plus_eps || *y2 - hh < 0.0f ||
      nr - (*y2 + hh) < one_plus_eps) /* :620-623 */
    status = OK_KLT_OOB;

  if (status == OK_KLT_TRACKED) { /* :626-646 */
    float a[SIMD_STEP] = {0};
    bilinear_weight(*x2, *y2, &xt2, &yt2, weights2);
    bilinear_patch(patch2, img2, nc, xt2, yt2, width, height, aligned, weights2);
    for (i = 0; i < aligned; i++) patchdiff[i] = patch1[i] - patch2[i];
    for (i = 0; i < aligned; i += SIMD_STEP)
      for (k = 0; k < SIMD_STEP; k++) a[k] = a[k] + fabsf(patchdiff[i + k]);
    if (lanes_sum(a) / (width * height) > max_residue) status = OK_KLT_LARGE_RESIDUE;
  }
  if (iters) *iters += iteration;
  if (status == OK_KLT_SMALL_DET) return OK_KLT_SMALL_DET; /* :655-667 */
  else if (status == OK_KLT_OOB) return OK_KLT_OOB;
  else if (status == OK_KLT_LARGE_RESIDUE) return OK_KLT_LARGE_RESIDUE;
  else if (iteration >= max_iterations) return OK_KLT_MAX_ITERATIONS;
  else return OK_KLT_TRACKED;
}

static int out_of_bounds(float x, float y, int ncols, int nrows, int borderx, int bordery) {
  return (x < borderx || x > ncols - 1 - borderx || y < bordery || y > nrows - 1 - bordery);
} /* LK.hpp:185-191 */

int ok_track_features(ok_params *p, int W, int H, float *const *pyr1, float *const *pyr2, int n,
                      float *x, float *y, int *val, const double *invK, double *norm3,
                      double *vec3, int *iters_out) { /* LK.hpp:972-1046 */
  float subsampling = (float)p->subsampling;
  int L = p->nPyramidLevels;
  int indx;
  if (L < 1 || L > 16) return -1;
  fix_window(p); /* :982-1013 */
  int lw[16], lh[16];
  for (indx = 0; indx < L; indx++) ok_level_size(p, W, H, indx, &lw[indx], &lh[indx]);
  int aligned = SIMD_STEP * ((p->window_width * p->window_height + SIMD_STEP) / SIMD_STEP);

#pragma omp parallel num_threads(g_threads)
  {
    float *buffers = (float *)malloc((size_t)9 * aligned * sizeof(float));
#pragma omp for schedule(dynamic, 16)
    for (indx = 0; indx < n; indx++) { /* KLTInvoker::operator(), LK.hpp:825-970 */
      if (iters_out) iters_out[indx] = 0;
      if (val[indx] >= 0) {
        float xloc = x[indx], yloc = y[indx];
        int r, v = OK_KLT_TRACKED;
        for (r = L - 1; r >= 0; r--) { xloc /= subsampling; yloc /= subsampling; } /* :837-839 */
        float xlocout = xloc, ylocout = yloc;
        for (r = L - 1; r >= 0; r--) { /* :859-881 */
          xloc *= subsampling; yloc *= subsampling;
          xlocout *= subsampling; ylocout *= subsampling;
          v = track_feature(xloc, yloc, &xlocout, &ylocout, pyr1[3 * r], pyr1[3 * r + 1],
                            pyr1[3 * r + 2], pyr2[3 * r], pyr2[3 * r + 1], pyr2[3 * r + 2], lw[r],
                            lh[r], p->window_width, p->window_height, p->step_factor,
                            p->max_iterations, p->min_determinant, p->min_displacement,
                            p->max_residue, buffers, iters_out ? &iters_out[indx] : NULL);
          if (v == OK_KLT_SMALL_DET || v == OK_KLT_OOB) break;
        }
        if (v == OK_KLT_OOB) { /* :886-915 */
          x[indx] = -1.0f; y[indx] = -1.0f; val[indx] = OK_KLT_OOB;
        } else if (out_of_bounds(xlocout, ylocout, W, H, p->borderx, p->bordery)) {
          x[indx] = -1.0f; y[indx] = -1.0f; val[indx] = OK_KLT_OOB;
        } else if (v == OK_KLT_SMALL_DET) {
          x[indx] = -1.0f; y[indx] = -1.0f; val[indx] = OK_KLT_SMALL_DET;
        } else if (v == OK_KLT_LARGE_RESIDUE) {
          x[indx] = -1.0f; y[indx] = -1.0f; val[indx] = OK_KLT_LARGE_RESIDUE;
        } else if (v == OK_KLT_MAX_ITERATIONS) {
          x[indx] = -1.0f; y[indx] = -1.0f; val[indx] = OK_KLT_MAX_ITERATIONS;
        } else {
          x[indx] = xlocout; y[indx] = ylocout; val[indx] = OK_KLT_TRACKED;
        }
      }
    }
    free(buffers);
  }

  if (invK && norm3 && vec3) { /* LK.hpp:1019-1035, fp64 */
    for (indx = 0; indx < n; indx++) {
      if (val[indx] != OK_KLT_TRACKED) continue;
      double px = (double)x[indx], py = (double)y[indx], pz = 1.0, q[3], nn;
      int i;
      for (i = 0; i < 3; i++) q[i] = (invK[3 * i] * px + invK[3 * i + 1] * py) + invK[3 * i + 2] * pz;
      nn = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
      for (i = 0; i < 3; i++) norm3[3 * indx + i] = q[i] / nn;
      for (i = 0; i < 3; i++) vec3[3 * indx + i] = norm3[3 * indx + i] / norm3[3 * indx + 2];
    }
  }
  return 0;
}

int ok_count_remaining(int n, const int *val) { /* LK.hpp:27-38 */
  int c = 0, i;
  for (i = 0; i < n; i++) if (val[i] >= 0) c++;
  return c;
}

/* ImageGrids.hpp:73-106 for every feature with val == KLT_TRACKED (System.cpp:246-267).
 * The 3x3 arithmetic runs inside OpenCV in the reference (MatExpr a*A + b*B -> scaled add,
 * gemm with transposed first operand, cv::norm L2, `dst *= s`): un-vendored, un-pinned ->
 * PARITY UNPINNED for these steps.  Restated as plain fp64, unfused, sums left to right. */
int ok_rotate_normalize(int n, const float *x, const float *y, const int *val, int grid_height,
                        int nrot, const double *rotations, const double *invK, double *norm3,
                        double *vec3) {
  int i, k, r, c;
  if (grid_height < 1 || nrot < 2) return -1;
  for (i = 0; i < n; i++) {
    int h0, h1;
    double y0, y1, dist0, dist1, dist, M[9], T[9], d[3], nv, inv, nn[3];
    const double *R0, *R1;
    if (val[i] != 0) continue;                       /* System.cpp:250 */
    h0 = (int)(y[i] / (float)grid_height);           /* :81 float / int */
    h1 = h0 + 1;
    if (h0 < 0 || h1 >= nrot) continue;              /* the reference indexes unchecked */
    y0 = (double)(h0 * grid_height);                 /* originGridVertices[vIndex0].y */
    y1 = (double)(h1 * grid_height);
    dist0 = (double)y[i] - y0;                       /* :87-93 */
    dist1 = y1 - (double)y[i];
    dist = dist0 + dist1;
    dist0 = dist1 / dist;
    dist1 = 1.0 - dist0;
    R0 = rotations + 9 * (size_t)h0;
    R1 = rotations + 9 * (size_t)h1;
    for (k = 0; k < 9; k++) M[k] = dist0 * R0[k] + dist1 * R1[k];          /* :94 */
    for (r = 0; r < 3; r++)                                                /* :95 M^T * invK */
      for (c = 0; c < 3; c++)
        T[3 * r + c] = (M[r] * invK[c] + M[3 + r] * invK[3 + c]) + M[6 + r] * invK[6 + c];
    for (r = 0; r < 3; r++)                                                /* :99-101 */
      d[r] = (T[3 * r] * (double)x[i] + T[3 * r + 1] * (double)y[i]) + T[3 * r + 2];
    nv = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);                  /* :102 */
    inv = 1.0 / nv;                                                        /* :103 */
    for (r = 0; r < 3; r++) { nn[r] = d[r] * inv; norm3[3 * (size_t)i + r] = nn[r]; }
    for (r = 0; r < 3; r++) vec3[3 * (size_t)i + r] = nn[r] / nn[2];       /* System.cpp:260 */
  }
  return 0;
}

/* ------------------------------------------------------------------ selector */

void ok_integral_products(const float *gx, const float *gy, int W, int H, float *sumxx,
                          float *sumyy, float *sumxy) {
  /* System.cpp:385-392: cv::multiply then cv::integral(src, sum, CV_32F):
   * sum(y+1,x+1) = sum(y,x+1) + rowprefix_y(x), running row prefix in fp32 */
  int x, y, W1 = W + 1;
  for (x = 0; x <= W; x++) sumxx[x] = sumyy[x] = sumxy[x] = 0.0f;
  for (y = 0; y < H; y++) {
    float sxx = 0.f, syy = 0.f, sxy = 0.f;
    float *oxx = sumxx + (size_t)(y + 1) * W1, *oyy = sumyy + (size_t)(y + 1) * W1,
          *oxy = sumxy + (size_t)(y + 1) * W1;
    oxx[0] = oyy[0] = oxy[0] = 0.0f;
    for (x = 0; x < W; x++) {
      float a = gx[(size_t)y * W + x], b = gy[(size_t)y * W + x];
      sxx += a * a; syy += b * b; sxy += a * b;
      oxx[x + 1] = oxx[x + 1 - W1] + sxx;
      oyy[x + 1] = oyy[x + 1 - W1] + syy;
      oxy[x + 1] = oxy[x + 1 - W1] + sxy;
    }
  }
}

static void fill_featuremap(int x, int y, uint8_t *fm, int mindist, int ncols, int nrows) {
  int ix, iy; /* selectGoodFeatures.hpp:37-49 */
  for (iy = y - mindist; iy <= y + mindist; iy++)
    for (ix = x - mindist; ix <= x + mindist; ix++)
      if (ix >= 0 && ix < ncols && iy >= 0 && iy < nrows) fm[(size_t)iy * ncols + ix] = 1;
}

/* :70-84 sorts by val descending with an unstable qsort (tie order unspecified in the
 * reference).  The build defines the total order: val desc, then y asc, then x asc. */
static int cmp_points(const void *a, const void *b) {
  const int *p = (const int *)a, *q = (const int *)b;
  if (p[2] != q[2]) return p[2] > q[2] ? -1 : 1;
  if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
  if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
  return 0;
}

static float min_eigenvalue(float gxx, float gxy, float gyy) { /* :220-223 */
  return (float)((gxx + gyy - sqrtf((gxx - gyy) * (gxx - gyy) + 4 * gxy * gxy)) / 2.0f);
}

/* candidate score at (x,y); returns 0 if not emitted.  :341-364 (plain), :725-749 (ROUND2).
 * A score above INT_MAX is clamped to INT_MAX (the reference converts 2^31 to int, which is
 * undefined; unreachable for 8-bit imagery). */
/* Direct window sums ("compute window sums directly ... separable box filter on
 * gx^2, gy^2, gx gy"; not in the reference, whose selector takes integral images).  When set, the
 * candidate score is formed from these H x W planes of window sums instead of four integral-image
 * taps: no cancellation between sums ~1e9.  ok_box_products defines the summation order. */
static const float *g_box[3] = {0, 0, 0};
void ok_set_box_sums(const float *bxx, const float *byy, const float *bxy) {
  g_box[0] = bxx; g_box[1] = byy; g_box[2] = bxy;
}
/* window sums of gx*gx, gy*gy, gx*gy over (2hw+1) x (2hh+1): each row of the window summed left to
 * right, then the row sums top to bottom; windows that leave the image are not candidates (0) */
void ok_box_products(const float *gx, const float *gy, int W, int H, int hw, int hh, float *bxx,
                     float *byy, float *bxy) {
  int x, y, q, dx, dy;
  float *out[3];
  out[0] = bxx; out[1] = byy; out[2] = bxy;
  for (q = 0; q < 3; q++) {
    float *rs = (float *)malloc(sizeof(float) * (size_t)W * H);
    for (y = 0; y < H; y++)
      for (x = 0; x < W; x++) {
        float acc = 0.0f;
        if (x - hw >= 0 && x + hw < W)
          for (dx = -hw; dx <= hw; dx++) {
            const float a = gx[(size_t)y * W + x + dx], b = gy[(size_t)y * W + x + dx];
            const float p = q == 0 ? a * a : (q == 1 ? b * b : a * b);
            acc = dx == -hw ? p : acc + p;
          }
        rs[(size_t)y * W + x] = acc;
      }
    for (y = 0; y < H; y++)
      for (x = 0; x < W; x++) {
        float acc = 0.0f;
        if (y - hh >= 0 && y + hh < H && x - hw >= 0 && x + hw < W)
          for (dy = -hh; dy <= hh; dy++) acc = dy == -hh ? rs[(size_t)(y + dy) * W + x] : acc + rs[(size_t)(y + dy) * W + x];
        out[q][(size_t)y * W + x] = acc;
      }
    free(rs);
  }
}

static int score_at(const float *sxx, const float *syy, const float *sxy, int W1, int x, int y,
                    int hw, int hh, int min_eig, int round2) {
  const size_t r0 = (size_t)(y - hh) * W1, r1 = (size_t)(y + hh + 1) * W1;
  const int x1 = x - hw, x2 = x + hw + 1;
  float gxy, gxx, gyy;
  if (g_box[0]) { /* direct sums: H x W planes, row length W1 - 1 */
    const size_t i = (size_t)y * (W1 - 1) + x;
    gxx = g_box[0][i]; gyy = g_box[1][i]; gxy = g_box[2][i];
  } else {
    gxy = sxy[r0 + x1] + sxy[r1 + x2] - sxy[r1 + x1] - sxy[r0 + x2];
    gxx = sxx[r0 + x1] + sxx[r1 + x2] - sxx[r1 + x1] - sxx[r0 + x2];
    gyy = syy[r0 + x1] + syy[r1 + x2] - syy[r1 + x1] - syy[r0 + x2];
  }
  float v = min_eigenvalue(gxx, gxy, gyy);
  int iv;
  if (round2) {
    if (v >= 2147483648.0f) iv = INT_MAX;
    else if (!(v > -2147483648.0f)) return 0;
    else iv = (int)(v + 0.5);
    return iv > min_eig ? iv : 0;
  }
  if (!(v > (float)min_eig)) return 0;
  if (v >= 2147483648.0f) return INT_MAX;
  iv = (int)v;
  return iv;
}

static int select_plain(ok_params *p, int mode, int round2, const float *sxx, const float *syy,
                        const float *sxy, int W, int H, int nFeatures, float *fx, float *fy,
                        int *fval) { /* :234-419 and :620-806 */
  const int overwrite = (mode == OK_SELECT_ALL);
  const int W1 = W + 1;
  fix_window(p);
  const int hw = p->window_width / 2, hh = p->window_height / 2;
  int *pointlist = (int *)malloc((size_t)W * H * 3 * sizeof(int));
  uint8_t *fm = (uint8_t *)calloc((size_t)W * H, 1);
  int npoints = 0, indx, x, y;
  if (!overwrite) /* :280-288 radius mindist */
    for (indx = 0; indx < nFeatures; indx++)
      if (fval[indx] >= 0) fill_featuremap((int)fx[indx], (int)fy[indx], fm, p->mindist, W, H);
  int bx = p->borderx < hw ? hw : p->borderx, by = p->bordery < hh ? hh : p->bordery;
  const int step = p->nSkippedPixels + 1;
  for (y = by; y < H - by; y += step) /* :324-366 */
    for (x = bx; x < W - bx; x += step) {
      if (fm[(size_t)y * W + x]) continue;
      int v = score_at(sxx, syy, sxy, W1, x, y, hw, hh, p->min_eigenvalue, round2);
      if (v) { pointlist[3 * npoints] = x; pointlist[3 * npoints + 1] = y; pointlist[3 * npoints + 2] = v; npoints++; }
    }
  qsort(pointlist, npoints, 3 * sizeof(int), cmp_points); /* :401 */
  { /* _enforceMinimumDistance :88-201 (radius mindist-1) / ...2 :507-618 (radius mindist) */
    int min_eig = p->min_eigenvalue < 1 ? 1 : p->min_eigenvalue;
    int r = round2 ? p->mindist : p->mindist - 1;
    int i = 0;
    indx = 0;
    while (1) {
      if (i >= npoints) {
        while (indx < nFeatures) {
          if (overwrite || fval[indx] < 0) { fx[indx] = -1; fy[indx] = -1; fval[indx] = OK_KLT_NOT_FOUND; }
          indx++;
        }
        break;
      }
      x = pointlist[3 * i]; y = pointlist[3 * i + 1];
      int v = pointlist[3 * i + 2];
      i++;
      while (!overwrite && indx < nFeatures && fval[indx] >= 0) indx++;
      if (indx >= nFeatures) break;
      if (!fm[(size_t)y * W + x] && v >= min_eig) {
        fx[indx] = (float)x; fy[indx] = (float)y; fval[indx] = v;
        indx++;
        fill_featuremap(x, y, fm, r, W, H);
      }
    }
  }
  free(fm);
  free(pointlist);
  return 0;
}

typedef struct { int val, y, x; } cell_entry;
static int cmp_cell(const void *a, const void *b) { /* :1026-1031 val desc; ties: y asc, x asc */
  const cell_entry *p = (const cell_entry *)a, *q = (const cell_entry *)b;
  if (p->val != q->val) return p->val > q->val ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  return 0;
}
typedef struct { float d; int idx; } seq_entry;
static int cmp_seq(const void *a, const void *b) { /* :1033-1050; ties: index asc */
  const seq_entry *p = (const seq_entry *)a, *q = (const seq_entry *)b;
  if (p->d != q->d) return p->d < q->d ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

static int select_uniform(ok_params *p, int mode, const float *sxx, const float *syy,
                          const float *sxy, int W, int H, int *nFeaturesIO, int capacity, float *fx,
                          float *fy, int *fval, int gridW, int gridH) { /* :1064-1338 */
  const int overwrite = (mode == OK_SELECT_ALL);
  const int W1 = W + 1;
  int nFeatures = *nFeaturesIO;
  if (gridW < 1 || gridH < 1) return -1;
  const int cw = W / gridW, ch = H / gridH; /* sizeByGrid :1330-1331 */
  if (cw < 1 || ch < 1) return -1;
  const int ncell = cw * ch;
  const int minFeatureInGrid = (int)((nFeatures / (float)ncell) + 0.5); /* :1333 */
  fix_window(p);
  const int hw = p->window_width / 2, hh = p->window_height / 2;
  int x, y, indx, i;

  seq_entry *seq = (seq_entry *)malloc(ncell * sizeof(seq_entry)); /* :1055-1061 */
  for (i = 0; i < ncell; i++) {
    float dY = (float)(i / cw - (float)ch / 2.0);
    float dX = (float)(i % cw - (float)cw / 2.0);
    seq[i].d = dX * dX + dY * dY;
    seq[i].idx = i;
  }
  qsort(seq, ncell, sizeof(seq_entry), cmp_seq);

  uint8_t *fm = (uint8_t *)calloc((size_t)W * H, 1);
  int *count = (int *)calloc(ncell + 1, sizeof(int));
  cell_entry *all = (cell_entry *)malloc(((size_t)W * H + nFeatures + 1) * sizeof(cell_entry));
  int *cellof = (int *)malloc(((size_t)W * H + nFeatures + 1) * sizeof(int));
  size_t nall = 0;
  /* The reference indexes grids[] with (y/gridH)*cw + x/gridW, which runs past the array when a
   * candidate lies in the remainder strip (W % gridW or H % gridH != 0).  The build clamps. */
#define CELL_OF(xx, yy) (((yy) / gridH >= ch ? ch - 1 : (yy) / gridH) * cw + ((xx) / gridW >= cw ? cw - 1 : (xx) / gridW))
  if (!overwrite) /* :1121-1139 */
    for (indx = 0; indx < nFeatures; indx++)
      if (fval[indx] >= 0) {
        x = (int)fx[indx]; y = (int)fy[indx];
        fill_featuremap(x, y, fm, p->mindist, W, H);
        if (x >= 0 && x < W && y >= 0 && y < H) {
          all[nall].val = INT_MAX; all[nall].x = x; all[nall].y = y;
          cellof[nall] = CELL_OF(x, y);
          nall++;
        }
      }
  int bx = p->borderx < hw ? hw : p->borderx, by = p->bordery < hh ? hh : p->bordery;
  const int step = p->nSkippedPixels + 1;
  for (y = by; y < H - by; y += step) /* :1172-1211 */
    for (x = bx; x < W - bx; x += step) {
      if (fm[(size_t)y * W + x]) continue;
      int v = score_at(sxx, syy, sxy, W1, x, y, hw, hh, p->min_eigenvalue, 0);
      if (v) {
        if (v == INT_MAX) v = INT_MAX - 1; /* INT_MAX is the "surviving feature" marker */
        all[nall].val = v; all[nall].x = x; all[nall].y = y;
        cellof[nall] = CELL_OF(x, y);
        nall++;
      }
    }
  /* bucket by cell, then sort each cell descending (:1215-1217) */
  size_t k;
  for (k = 0; k < nall; k++) count[cellof[k] + 1]++;
  for (i = 0; i < ncell; i++) count[i + 1] += count[i];
  cell_entry *bucket = (cell_entry *)malloc((nall + 1) * sizeof(cell_entry));
  int *fillpos = (int *)malloc((ncell + 1) * sizeof(int));
  memcpy(fillpos, count, (ncell + 1) * sizeof(int));
  for (k = 0; k < nall; k++) bucket[fillpos[cellof[k]]++] = all[k];
  for (i = 0; i < ncell; i++)
    qsort(bucket + count[i], count[i + 1] - count[i], sizeof(cell_entry), cmp_cell);

  int rc = 0;
  indx = 0;
  for (i = 0; i < ncell && rc == 0; i++) { /* :1222-1293 */
    int c = seq[i].idx, featureAdded = 0, j;
    for (j = count[c]; j < count[c + 1]; j++) {
      int v = bucket[j].val;
      if (v == INT_MAX) { featureAdded++; continue; }
      x = bucket[j].x; y = bucket[j].y;
      while (!overwrite && indx < nFeatures && fval[indx] >= 0) indx++;
      if (indx >= nFeatures) {
        if (featureAdded > minFeatureInGrid) break;
        if (nFeatures >= capacity) { rc = -2; break; } /* reference: capacity is 100x, unchecked */
        nFeatures++; /* :1257-1261 */
        fx[indx] = -1; fy[indx] = -1; fval[indx] = -1;
      }
      if (!fm[(size_t)y * W + x]) {
        fx[indx] = (float)x; fy[indx] = (float)y; fval[indx] = v;
        indx++;
        fill_featuremap(x, y, fm, p->mindist, W, H);
        featureAdded++;
      }
    }
  }
  while (indx < nFeatures) { /* :1296-1312 */
    if (overwrite || fval[indx] < 0) { fx[indx] = -1; fy[indx] = -1; fval[indx] = OK_KLT_NOT_FOUND; }
    indx++;
  }
  *nFeaturesIO = nFeatures;
  free(fillpos); free(bucket); free(cellof); free(all); free(count); free(fm); free(seq);
  return rc;
}

int ok_select_features(ok_params *p, int mode, int variant, const float *sumxx,
                       const float *sumyy, const float *sumxy, int W, int H, int *nFeatures,
                       int capacity, float *x, float *y, int *val, int gridW, int gridH) {
  if (mode == OK_REPLACE_SOME) { /* :436-449, :1009-1022, :1341-1355 */
    int lost = *nFeatures - ok_count_remaining(*nFeatures, val);
    if (lost <= 0) return 0;
    /* KLTReplaceLostFeaturesUniform falls back to the plain replace (:1353);
     * KLTReplaceLostFeatures2 calls the rounding overload (:1020) */
    if (variant == OK_VARIANT_UNIFORM) variant = OK_VARIANT_PLAIN;
  }
  (void)capacity;
  switch (variant) {
    case OK_VARIANT_PLAIN:
      return select_plain(p, mode, 0, sumxx, sumyy, sumxy, W, H, *nFeatures, x, y, val);
    case OK_VARIANT_ROUND2:
      /* KLTSelectGoodFeatures2 (:997-1006) runs _KLTSelectGoodFeatures2 = the plain body;
       * only KLTReplaceLostFeatures2 reaches the rounding overload */
      if (mode == OK_SELECT_ALL)
        return select_plain(p, mode, 0, sumxx, sumyy, sumxy, W, H, *nFeatures, x, y, val);
      return select_plain(p, mode, 1, sumxx, sumyy, sumxy, W, H, *nFeatures, x, y, val);
    case OK_VARIANT_UNIFORM:
      return select_uniform(p, mode, sumxx, sumyy, sumxy, W, H, nFeatures, capacity, x, y, val,
                            gridW, gridH);
  }
  return -1;
}
