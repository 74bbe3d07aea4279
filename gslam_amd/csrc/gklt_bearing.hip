// Bearing vectors with rolling-shutter rotation compensation: the step that System::Track runs on
// the feature list right after KLTTrackFeatures (System.cpp:539-540 -> NormalizeFrameInvoker,
// System.cpp:246-267 -> ImageGrids::rotateAndNormalizePoint, ImageGrids.hpp:73-106).  SURVEY 8(f)
// rank 1.  For every feature with val == KLT_TRACKED:
//     h0 = (int)(y / gridHeight)                       row band of the feature (float division)
//     a  = (y1 - y) / ((y - y0) + (y1 - y)),  y0 = h0 * gridHeight, y1 = y0 + gridHeight
//     M  = a * R[h0] + (1 - a) * R[h0 + 1]             per-row gyro rotations, interpolated
//     T  = M^T * invK
//     d  = T * (x, y, 1);  norm = d * (1 / |d|);  vec = norm / norm[2]
// all in fp64, every product and sum rounded separately (no FMA), sums left to right -- the order
// of the test oracle.  The matrix arithmetic runs inside OpenCV in the reference (MatExpr
// a*A + b*B, gemm with a transposed operand, cv::norm, scaled convertTo): un-vendored and
// un-pinned, so parity with the reference itself is UNPINNED for those steps.
// One thread per feature; untracked features are not touched.
#include "gklt_internal.h"

using namespace gklt;

namespace {

__global__ __launch_bounds__(256) void k_rotate_normalize(int n, const float *__restrict__ fx,
                                                          const float *__restrict__ fy,
                                                          const int *__restrict__ fval,
                                                          int grid_h, int nrot,
                                                          const double *__restrict__ rot,
                                                          const double *__restrict__ invk,
                                                          double *__restrict__ norm3,
                                                          double *__restrict__ vec3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || fval[i] != GKLT_TRACKED) return;
  const float x = fx[i], y = fy[i];
  const int h0 = (int)(y / (float)grid_h);  // ImageGrids.hpp:81
  const int h1 = h0 + 1;
  // the reference indexes rotations[h1] unchecked; a feature below the last vertex row (frame
  // height not a multiple of the grid height) is left untouched here
  if (h0 < 0 || h1 >= nrot) return;
  const double y0 = (double)(h0 * grid_h), y1 = (double)(h1 * grid_h);
  double dist0 = (double)y - y0;  // :87-93
  double dist1 = y1 - (double)y;
  const double dist = dist0 + dist1;
  dist0 = dist1 / dist;
  dist1 = 1.0 - dist0;
  const double *R0 = rot + 9 * (size_t)h0, *R1 = rot + 9 * (size_t)h1;
  double M[9], T[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = dist0 * R0[k] + dist1 * R1[k];  // :94
#pragma unroll
  for (int r = 0; r < 3; ++r)  // :95  T = M^T * invK
#pragma unroll
    for (int c = 0; c < 3; ++c)
      T[3 * r + c] = (M[r] * invk[c] + M[3 + r] * invk[3 + c]) + M[6 + r] * invk[6 + c];
  const double px = (double)x, py = (double)y;
  double d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = (T[3 * r] * px + T[3 * r + 1] * py) + T[3 * r + 2];  // :99-101
  const double nv = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);                      // :102
  const double inv = 1.0 / nv;                                                            // :103
  double nn[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    nn[r] = d[r] * inv;
    norm3[3 * (size_t)i + r] = nn[r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) vec3[3 * (size_t)i + r] = nn[r] / nn[2];  // System.cpp:260-261
}

int upload_rotations(gklt_ctx *ctx, hipStream_t s, int nrot, const double *rotations,
                     const double *invK) {
  const int need = 9 * nrot + 9;
  if (need > ctx->d_rot_capacity) {
    GKLT_HIP(hipStreamSynchronize(s));
    if (ctx->d_rot) (void)hipFree(ctx->d_rot);
    ctx->d_rot = nullptr;
    ctx->d_rot_capacity = 0;
    GKLT_HIP(hipMalloc(&ctx->d_rot, sizeof(double) * (size_t)need));
    ctx->d_rot_capacity = need;
  }
  GKLT_HIP(hipMemcpyAsync(ctx->d_rot, invK, 9 * sizeof(double), hipMemcpyHostToDevice, s));
  GKLT_HIP(hipMemcpyAsync(ctx->d_rot + 9, rotations, sizeof(double) * 9 * (size_t)nrot,
                          hipMemcpyHostToDevice, s));
  return 0;
}

// list on the device (dx, dy, dval); results into host norm3 / vec3
int rotate_normalize_device_list(gklt_ctx *ctx, int n, const float *dx, const float *dy,
                                 const int *dval, int grid_h, int nrot, const double *rotations,
                                 const double *invK, double *norm3, double *vec3) {
  hipStream_t s = ctx->s_trk;
  if (int rc = upload_rotations(ctx, s, nrot, rotations, invK)) return rc;
  // page-locked outputs are written by the kernel itself; pageable ones round-trip through a
  // device buffer so that entries of untracked features keep the caller's contents
  void *pn = nullptr, *pv = nullptr;
  const bool direct = hipHostGetDevicePointer(&pn, norm3, 0) == hipSuccess && pn &&
                      hipHostGetDevicePointer(&pv, vec3, 0) == hipSuccess && pv;
  if (!direct) {
    (void)hipGetLastError();
    if (int rc = ensure_bearing_buffers(ctx, n)) return rc;
    GKLT_HIP(hipMemcpyAsync(ctx->d_norm, norm3, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
    GKLT_HIP(hipMemcpyAsync(ctx->d_vec, vec3, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
    pn = ctx->d_norm;
    pv = ctx->d_vec;
  }
  {
    ProfScope ps(ctx, s, GKLT_K_MISC);
    hipLaunchKernelGGL(k_rotate_normalize, dim3((n + 255) / 256), dim3(256), 0, s, n, dx, dy, dval,
                       grid_h, nrot, ctx->d_rot + 9, ctx->d_rot, static_cast<double *>(pn),
                       static_cast<double *>(pv));
    GKLT_HIP(hipGetLastError());
  }
  if (!direct) {
    GKLT_HIP(hipMemcpyAsync(norm3, ctx->d_norm, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
    GKLT_HIP(hipMemcpyAsync(vec3, ctx->d_vec, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
  }
  GKLT_HIP(hipStreamSynchronize(s));
  return 0;
}

int check_args(gklt_ctx *ctx, int n, int grid_h, int nrot, const double *rotations,
               const double *invK, double *norm3, double *vec3) {
  if (!ctx || n < 0 || grid_h < 1 || nrot < 2 || !rotations || !invK || !norm3 || !vec3) {
    set_error("rotate_normalize: bad argument (grid height >= 1, at least 2 row rotations)");
    return GKLT_EINVAL;
  }
  return 0;
}

}  // namespace

extern "C" int gklt_flist_rotate_normalize(gklt_ctx *ctx, gklt_flist *fl, int grid_height, int nrot,
                                           const double *rotations, const double *invK,
                                           double *norm3, double *vec3) {
  if (!fl || (ctx && fl->ctx != ctx)) return GKLT_EINVAL;
  if (int rc = check_args(ctx, fl->n, grid_height, nrot, rotations, invK, norm3, vec3)) return rc;
  if (fl->n == 0) return 0;
  GKLT_HIP(hipSetDevice(ctx->device));
  return rotate_normalize_device_list(ctx, fl->n, fl->x, fl->y, fl->val, grid_height, nrot,
                                      rotations, invK, norm3, vec3);
}

extern "C" int gklt_rotate_normalize(gklt_ctx *ctx, int n, const float *x, const float *y,
                                     const int *val, int grid_height, int nrot,
                                     const double *rotations, const double *invK, double *norm3,
                                     double *vec3) {
  if (int rc = check_args(ctx, n, grid_height, nrot, rotations, invK, norm3, vec3)) return rc;
  if (n == 0) return 0;
  if (!x || !y || !val) return GKLT_EINVAL;
  GKLT_HIP(hipSetDevice(ctx->device));
  if (!ctx->scratch_list || ctx->scratch_list->capacity < n) {
    if (ctx->scratch_list) gklt_flist_destroy(ctx->scratch_list);
    ctx->scratch_list = nullptr;
    if (int rc = gklt_flist_create(ctx, n > 4096 ? n : 4096, &ctx->scratch_list)) return rc;
  }
  gklt_flist *fl = ctx->scratch_list;
  if (int rc = gklt_flist_upload(ctx, fl, n, x, y, val)) return rc;
  return rotate_normalize_device_list(ctx, n, fl->x, fl->y, fl->val, grid_height, nrot, rotations,
                                      invK, norm3, vec3);
}
