// Host-side context arithmetic: tracking-context defaults, pyramid depth, default border and
// the Gaussian / Gaussian-derivative taps.  No GPU involved.
//
// Mirrors KLTUtil.cpp of the reference (file:line cited per function); the struct is a POD
// copy of the KLT_TrackingContextRec fields that the hot path reads (KLT.h:18-67).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "gklt_internal.h"

namespace gklt {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

// The reference prints a KLTWarning and silently fixes the window in four places
// (KLTUtil.cpp:94-113,239-258; LK.hpp:982-1013; selectGoodFeatures.hpp:249-268).
int fix_window(gklt_params *p) {
  int n = 0;
  auto fix = [&](int &w) {
    if (w % 2 != 1) { ++w; ++n; }
    if (w < 3) { w = 3; ++n; }
  };
  fix(p->window_width);
  fix(p->window_height);
  return n;
}

int half_taps(const float *full, int width, int sym, Taps *out) {
  if (width < 1 || width > GKLT_MAX_KERNEL_WIDTH || width % 2 == 0) return GKLT_EINVAL;
  out->r = width / 2;
  out->sym = sym;
  for (int j = 0; j <= out->r; ++j) out->k[j] = full[out->r + j];
  return 0;
}

}  // namespace gklt

using namespace gklt;

extern "C" const char *gklt_last_error(void) { return g_err.c_str(); }

extern "C" int gklt_compute_kernels(float sigma, float *gauss, int *gauss_width, float *deriv,
                                    int *deriv_width) {
  // KLTUtil.cpp:136-199.  fp32 throughout; exp() on a float argument is the float overload.
  constexpr int N = GKLT_MAX_KERNEL_WIDTH, C = N / 2;
  if (!(sigma >= 0.0f)) {  // the reference asserts sigma >= 0 (:144)
    set_error("computeKernels: sigma %f < 0", (double)sigma);
    return GKLT_EINVAL;
  }
  float g[N], d[N];
  const float peak_g = 1.0f;
  const float peak_d = sigma * std::exp(-0.5f);
  for (int i = -C; i <= C; ++i) {
    g[i + C] = std::exp(static_cast<float>(-i * i) / (2 * sigma * sigma));
    d[i + C] = -i * g[i + C];
  }
  // shrink from both ends while the outermost tap is below 1 % of the peak (:158-163)
  auto trimmed_width = [&](const float *v, float peak) {
    int w = N;
    for (int i = 0; i < N && std::fabs(v[i] / peak) < 0.01f; ++i) w -= 2;
    return w;
  };
  const int gw = trimmed_width(g, peak_g), dw = trimmed_width(d, peak_d);
  if (gw == N || dw == N || gw < 1 || dw < 1) {  // :164-167 KLTError
    set_error("(_computeKernels) MAX_KERNEL_WIDTH %d is too small for a sigma of %f", N,
              (double)sigma);
    return GKLT_EKERNEL;
  }
  std::memmove(g, g + (N - gw) / 2, gw * sizeof(float));  // :172-176
  std::memmove(d, d + (N - dw) / 2, dw * sizeof(float));
  float den = 0.0f;  // :180-197
  for (int i = 0; i < gw; ++i) den += g[i];
  for (int i = 0; i < gw; ++i) g[i] /= den;
  den = 0.0f;
  for (int i = -(dw / 2); i <= dw / 2; ++i) den += i * d[i + dw / 2];
  for (int i = 0; i < dw; ++i) d[i] /= den;
  if (gauss) std::memcpy(gauss, g, gw * sizeof(float));
  if (deriv) std::memcpy(deriv, d, dw * sizeof(float));
  if (gauss_width) *gauss_width = gw;
  if (deriv_width) *deriv_width = dw;
  return 0;
}

extern "C" float gklt_compute_smooth_sigma(const gklt_params *p) {  // KLTUtil.cpp:215-217
  return p->smooth_sigma_fact * std::max(p->window_width, p->window_height);
}

extern "C" int gklt_change_tc_pyramid(gklt_params *p, int search_range, int *nfixes) {
  // KLTUtil.cpp:87-134
  if (!p) return GKLT_EINVAL;
  const int n = fix_window(p);
  if (nfixes) *nfixes = n;
  const float half = std::min(p->window_width, p->window_height) / 2.0f;
  const float ratio = static_cast<float>(search_range) / half;
  if (ratio < 1.0) {
    p->nPyramidLevels = 1;
  } else if (ratio <= 9.0) {
    p->nPyramidLevels = 2;
    p->subsampling = ratio <= 3.0 ? 2 : (ratio <= 5.0 ? 4 : 8);
  } else {
    const float v = static_cast<float>(std::log(7.0 * ratio + 1.0) / std::log(8.0));
    p->nPyramidLevels = static_cast<int>(v + 0.99);
    p->subsampling = 8;
  }
  return 0;
}

extern "C" int gklt_update_tc_border(gklt_params *p) {  // KLTUtil.cpp:225-295
  if (!p) return GKLT_EINVAL;
  fix_window(p);
  int gw = 0, dw = 0;
  int rc = gklt_compute_kernels(gklt_compute_smooth_sigma(p), nullptr, &gw, nullptr, &dw);
  if (rc) return rc;
  const int smooth_hw = gw / 2;
  rc = gklt_compute_kernels(p->pyramid_sigma_fact * p->subsampling, nullptr, &gw, nullptr, &dw);
  if (rc) return rc;
  const int pyr_hw = gw / 2;
  int invalid = smooth_hw, scale = 1;
  for (int l = 1; l < p->nPyramidLevels; ++l) {
    const float v = (static_cast<float>(invalid) + pyr_hw) / p->subsampling;
    invalid = static_cast<int>(v + 0.99);
    scale *= p->subsampling;
  }
  const int border = (invalid + std::max(p->window_width, p->window_height) / 2) * scale;
  p->borderx = p->bordery = border;
  return 0;
}

extern "C" int gklt_default_params(gklt_params *p) {  // KLTUtil.cpp:12-46,298-342
  if (!p) return GKLT_EINVAL;
  std::memset(p, 0, sizeof *p);
  p->mindist = 10;
  p->window_width = p->window_height = 21;
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
  p->subsampling = 2;
  gklt_change_tc_pyramid(p, 15, nullptr);
  return gklt_update_tc_border(p);
}

extern "C" int gklt_feature_list_capacity(int nFeatures) {  // KLTUtil.cpp:53-55,68
  if (nFeatures < 0 || nFeatures > 0x7fffffff / 100) {
    set_error("KLTCreateFeatureList: %d features", nFeatures);
    return GKLT_EINVAL;
  }
  return 100 * nFeatures;
}

extern "C" int gklt_count_remaining(int n, const int *val) {  // LK.hpp:27-38
  if (n > 0 && !val) {  // (found by the UBSAN run of tests/test_sanitize.py)
    set_error("gklt_count_remaining: %d entries but no array", n);
    return GKLT_EINVAL;
  }
  int c = 0;
  for (int i = 0; i < n; ++i) c += val[i] >= 0;
  return c;
}
