/*
 * gklt.h -- C ABI of the MI355X (gfx950) KLT front end that replaces GSLAM's
 * pyramid.cpp / convolve.cpp / KLTUtil.cpp / LK.hpp / selectGoodFeatures.hpp.
 *
 * Plain pointers and sizes only.  Every function returns 0 on success and a negative
 * GKLT_E* code on failure; gklt_last_error() returns the message of the calling thread's
 * last failure.  Citations are file:line under /root/reference/GSLAM/.
 *
 * Threading: gklt_pyramid_build() and gklt_track*() may be called concurrently from two host
 * threads on the same context (the reference builds pyramid t+1 on a std::thread while
 * tracking t-1 -> t, System.cpp:508-518).  They run on two HIP streams of the context and are
 * ordered against each other per pyramid by events, so a pyramid is never rebuilt while a
 * tracker launch still reads it and never tracked before its build finished.
 */
#ifndef GKLT_H
#define GKLT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKLT_MAX_KERNEL_WIDTH 71 /* KLT.h:107 MAX_KERNEL_WIDTH */
#define GKLT_MAX_LEVELS 12
#define GKLT_MAX_LOOKAHEAD 16 /* frames per batched build / multi-frame tracker launch */
#define GKLT_MAX_STREAMS 64   /* video streams of one gklt_multistream */

/* KLT.h:69-74 */
#define GKLT_TRACKED 0
#define GKLT_NOT_FOUND -1
#define GKLT_SMALL_DET -2
#define GKLT_MAX_ITERATIONS -3
#define GKLT_OOB -4
#define GKLT_LARGE_RESIDUE -5

#define GKLT_EINVAL -1   /* bad argument */
#define GKLT_EHIP -2     /* a HIP runtime call failed */
#define GKLT_EKERNEL -3  /* the reference would call KLTError (e.g. kernel wider than 71) */
#define GKLT_ENOMEM -4
#define GKLT_ECAPACITY -5 /* feature list capacity exceeded */
#define GKLT_EUNSUPPORTED -6 /* a reference entry point that is not built on the GPU yet */

/* The fields of KLT_TrackingContextRec (KLT.h:18-67) that the hot path reads. */
typedef struct gklt_params {
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
} gklt_params;

typedef struct gklt_ctx gklt_ctx;
typedef struct gklt_pyramid gklt_pyramid;
typedef struct gklt_flist gklt_flist;
typedef struct gklt_frame gklt_frame;

const char *gklt_last_error(void);

/* ---- K1-K4: context parameters and convolution kernels (host only, no GPU needed) ---- */

/* KLTCreateTrackingContext (KLTUtil.cpp:298-342): defaults of KLTUtil.cpp:12-46, then
 * KLTChangeTCPyramid(tc,15) and KLTUpdateTCBorder(tc). */
int gklt_default_params(gklt_params *p);
/* KLTChangeTCPyramid (KLTUtil.cpp:87-134).  *nfixes (may be NULL) = number of window
 * corrections applied where the reference prints a KLTWarning. */
int gklt_change_tc_pyramid(gklt_params *p, int search_range, int *nfixes);
/* KLTUpdateTCBorder (KLTUtil.cpp:225-295). */
int gklt_update_tc_border(gklt_params *p);
/* computeKernels (KLTUtil.cpp:136-199).  gauss/deriv hold >= 71 floats.  GKLT_EKERNEL where
 * the reference calls KLTError. */
int gklt_compute_kernels(float sigma, float *gauss, int *gauss_width, float *deriv,
                         int *deriv_width);
/* _KLTComputeSmoothSigma (KLTUtil.cpp:215-217). */
float gklt_compute_smooth_sigma(const gklt_params *p);
/* KLTCountRemainingFeatures (LK.hpp:27-38). */
int gklt_count_remaining(int n, const int *val);
/* KLTCreateFeatureList (KLTUtil.cpp:49-83) allocates 100 x nFeatures records so that
 * KLTSelectGoodFeaturesUniform can grow the list (selectGoodFeatures.hpp:1257): the number of
 * entries a host SoA list, a gklt_flist or gklt_select's `capacity` needs for a list created with
 * nFeatures.  The AoS list itself (KLT_FeatureRec holds cv::Mat / Eigen members) is built by the
 * C++ shim (shim/gklt_shim.cpp) with this capacity.  Negative nFeatures: GKLT_EINVAL. */
int gklt_feature_list_capacity(int nFeatures);

/* ---- context ---- */

int gklt_ctx_create(int device, gklt_ctx **out);
int gklt_ctx_destroy(gklt_ctx *ctx);
/* Copies *p, applies the window warn-and-fix of LK.hpp:982-1013 and validates. */
int gklt_ctx_set_params(gklt_ctx *ctx, const gklt_params *p);
int gklt_ctx_get_params(gklt_ctx *ctx, gklt_params *p);
/* Waits for both streams of the context. */
int gklt_sync(gklt_ctx *ctx);

/* ---- P1-P4: pyramid (pyramid.cpp:45-84, convolve.cpp:19-56, System.h:30-56) ---- */

/* Device-resident pyramid for W x H frames at the context's current levels/subsampling. */
int gklt_pyramid_create(gklt_ctx *ctx, int width, int height, gklt_pyramid **out);
int gklt_pyramid_destroy(gklt_pyramid *pyr);
int gklt_pyramid_levels(const gklt_pyramid *pyr);
int gklt_pyramid_level_size(const gklt_pyramid *pyr, int level, int *width, int *height);
/* computePyramid (pyramid.cpp:45): asynchronous on the context's pyramid stream.  img is an
 * 8-bit grayscale frame with row stride `stride` bytes, in host memory (img_on_device = 0; it
 * is staged H2D first) or in device memory (img_on_device = 1). */
int gklt_pyramid_build(gklt_ctx *ctx, gklt_pyramid *pyr, const uint8_t *img, int stride,
                       int img_on_device);
/* Copies plane (0 = image, 1 = gradx, 2 = grady) of `level` to host; dst_stride in floats.
 * This is pyramid[3*level+plane] of the reference's std::vector<cv::Mat>.  Synchronous. */
int gklt_pyramid_download(gklt_ctx *ctx, gklt_pyramid *pyr, int level, int plane, float *dst,
                          int dst_stride);
/* The image plane of `level` as 8-bit: pyramid[3*level].convertTo(ucharImage, CV_8UC1)
 * (System.cpp:403,672: the coarsest level handed to the ORB keyframe path), i.e.
 * saturate_cast<uchar>(cvRound(v)) with round-half-to-even.  dst_stride in bytes.  A quarter of the
 * bytes of the float download and no host conversion pass.  Synchronous. */
int gklt_pyramid_download_u8(gklt_ctx *ctx, gklt_pyramid *pyr, int level, uint8_t *dst,
                             int dst_stride);
/* Overwrites one plane from host memory (tests: track on externally produced pyramids). */
int gklt_pyramid_upload(gklt_ctx *ctx, gklt_pyramid *pyr, int level, int plane, const float *src,
                        int src_stride);

/* computeSmoothedImage (convolve.cpp:46-56) and computeGradients (convolve.cpp:19-42) on dense
 * host fp32 images; synchronous. */
int gklt_smooth(gklt_ctx *ctx, const float *src, int width, int height, float sigma, float *dst);
int gklt_gradients(gklt_ctx *ctx, const float *src, int width, int height, float sigma,
                   float *gradx, float *grady);

/* ---- T1-T9: tracker (LK.hpp:413-669, 825-1046) ---- */

/* KLTTrackFeatures (LK.hpp:972): host structure-of-arrays feature list, updated in place;
 * features with val < 0 are left untouched.  invK (row-major 3x3), norm3 and vec3 (n*3 doubles,
 * written for TRACKED features only) may all be NULL to skip the bearing-vector epilogue
 * (LK.hpp:1019-1035).  Synchronous: returns when x, y, val hold the results. */
int gklt_track(gklt_ctx *ctx, gklt_pyramid *pyr1, gklt_pyramid *pyr2, int n, float *x, float *y,
               int *val, const double *invK, double *norm3, double *vec3);

/* Device-resident feature list: lets consecutive frames chain without a host round trip. */
int gklt_flist_create(gklt_ctx *ctx, int capacity, gklt_flist **out);
int gklt_flist_destroy(gklt_flist *fl);
int gklt_flist_upload(gklt_ctx *ctx, gklt_flist *fl, int n, const float *x, const float *y,
                      const int *val);
/* Asynchronous on the tracker stream; x/y/val should be pinned for true overlap.  Follow with
 * gklt_sync() (or gklt_flist_download, which synchronises). */
int gklt_flist_download_async(gklt_ctx *ctx, gklt_flist *fl, float *x, float *y, int *val);
int gklt_flist_download(gklt_ctx *ctx, gklt_flist *fl, float *x, float *y, int *val);
/* One asynchronous copy of the whole list: dst (3 * capacity * 4 bytes, ideally pinned) receives
 * x[capacity] | y[capacity] | val[capacity] back to back. */
int gklt_flist_download_packed_async(gklt_ctx *ctx, gklt_flist *fl, void *dst);
int gklt_flist_size(const gklt_flist *fl);
/* Same arithmetic as gklt_track on the device-resident list; asynchronous. */
int gklt_track_flist(gklt_ctx *ctx, gklt_pyramid *pyr1, gklt_pyramid *pyr2, gklt_flist *fl);
/* Running totals accumulated by every gklt_track_flist call on this list since creation /
 * the last reset: features that ended TRACKED, and Newton iterations executed.  Synchronous. */
int gklt_flist_stats(gklt_ctx *ctx, gklt_flist *fl, uint64_t *tracked, uint64_t *iterations,
                     int reset);

/* No counterpart in the reference.  For callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc, code that runs between the capture regions of a HIP
 * graph).  Allocation-free but SYNCHRONOUS -- it waits on its stream and copies pageable arrays --
 * so it cannot itself run while that stream is being captured.  gklt_track() on host
 * or device x / y / val arrays of n <= GKLT_NOALLOC_MAX_FEATURES entries, with the list staging and
 * the tracker counters in a fixed array that libgklt.so holds on every GPU it runs on (388 KiB,
 * until the process ends) instead of hipMalloc'd buffers; the call makes no hipMalloc / hipFree
 * and creates no events (kernel profiling brackets are skipped).  Results are those of gklt_track
 * bit for bit; *tracked (may be NULL) = features that ended KLT_TRACKED.  More features:
 * GKLT_ECAPACITY.  Synchronous.  All calls on one GPU share that GPU's array: calls that overlap
 * on one GPU (two host threads, two contexts on one device) race -- serialise them. */
#define GKLT_NOALLOC_MAX_FEATURES 32768
int gklt_track_noalloc(gklt_ctx *ctx, gklt_pyramid *pyr1, gklt_pyramid *pyr2, int n, float *x,
                       float *y, int *val, uint64_t *tracked);

/* ---- S1-S6: Shi-Tomasi selection (selectGoodFeatures.hpp, System.cpp:385-392) ---- */

enum { GKLT_SELECT_ALL = 0, GKLT_REPLACE_SOME = 1 };                         /* selectionMode, :34 */
enum { GKLT_VARIANT_PLAIN = 0, GKLT_VARIANT_ROUND2 = 1, GKLT_VARIANT_UNIFORM = 2 };

/* The inputs the selector presupposes (System.cpp:385-392): gx*gx, gy*gy, gx*gy of pyramid
 * level 0 and their cv::integral(..., CV_32F) images, each (H+1) x (W+1) dense fp32, to host. */
int gklt_integral_products(gklt_ctx *ctx, gklt_pyramid *pyr, float *sumDxx, float *sumDyy,
                           float *sumDxy);
/* KLTSelectGoodFeatures / KLTReplaceLostFeatures (selectGoodFeatures.hpp:424-449), ...2
 * (:997-1022) and KLTReplaceLostFeaturesUniform (:1341-1355, which runs the plain replace) on host
 * integral images and a host SoA list of *nFeatures entries, updated in place.  Ties between
 * equal scores, unspecified in the reference (qsort / std::sort), are ordered y ascending then x
 * ascending; grid cells at equal distance from the centre by index.
 * KLTSelectGoodFeaturesUniform (mode ALL, variant UNIFORM, selectGoodFeatures.hpp:1318-1338)
 * takes the cell size in pixels in gridW/gridH (the reference's global gridSize, :28) and may
 * GROW the list: *nFeatures is updated and x/y/val must hold `capacity` entries (the reference
 * allocates 100x, KLTUtil.cpp:53-55); GKLT_ECAPACITY if it would grow beyond that. */
int gklt_select(gklt_ctx *ctx, int mode, int variant, const float *sumDxx, const float *sumDyy,
                const float *sumDxy, int width, int height, int *nFeatures, int capacity, float *x,
                float *y, int *val, int gridW, int gridH);
/* Same selection with the integral images computed on the device from pyramid level 0 and the
 * list resident on the device; asynchronous apart from the convergence polls. */
/* gklt_select_prepare (optional, no counterpart in the reference: System.cpp:385-392 computes the
 * integral images right before selecting): computes the integral images of `pyr` (built by
 * gklt_pyramid_build) AHEAD, on the build stream right behind the build, so that a later
 * gklt_select_from_pyramid on the same, unchanged pyramid finds them ready -- in a loop
 * build(t+1); prepare(t+1); select(t); track(t -> t+1) they are computed while select(t) runs.
 * Same bits either way; at most two preparations are kept. */
int gklt_select_prepare(gklt_ctx *ctx, gklt_pyramid *pyr);
int gklt_select_from_pyramid(gklt_ctx *ctx, gklt_pyramid *pyr, int mode, int variant,
                             gklt_flist *fl, int gridW, int gridH);

/* ---- keyframe seeding on the device ---- */

/* System::Track re-seeds the tracked list at every keyframe from keypoints detected on the COARSEST
 * pyramid image, scaled back by mScaleFactor = subsampling^(nPyramidLevels-1), all alive as
 * KLT_TRACKED (System.cpp:407-420,672-689); the reference downloads that plane, converts it to 8 bit
 * and runs the ORB extractor's FAST detector (ORBextractor.cc:807-900, OpenCV's cv::FAST, not in
 * the reference tree).  gklt_seed_from_level does the step without leaving the GPU, with the
 * detector this library has, the reference's own Shi-Tomasi selector (KLTSelectGoodFeatures,
 * selectGoodFeatures.hpp:234-433, the call System.cpp:385-394 keeps commented out): the integral
 * images of gx^2, gy^2, gx*gy of level `level`, plain selection of nFeatures corners there (window,
 * mindist, min_eigenvalue, nSkippedPixels of the context in that level's pixels; border =
 * ceil(border / scale) so that every seed lies inside the level-0 border the tracker tests), then
 * x *= scale, y *= scale, val = KLT_TRACKED with scale = subsampling^level.  The list gets
 * nFeatures entries; entries for which no corner was left stay (-1, -1, KLT_NOT_FOUND).
 * Asynchronous apart from the selector's convergence polls; ordered after the pyramid's build. */
int gklt_seed_from_level(gklt_ctx *ctx, gklt_pyramid *pyr, int level, gklt_flist *fl, int nFeatures);

/* ---- bearing vectors with per-row rotation ---- */

/* ImageGrids::rotateAndNormalizePoint (ImageGrids.hpp:73-106) applied as NormalizeFrameInvoker
 * does right after KLTTrackFeatures (System.cpp:246-267,539-540): for every feature with
 * val == KLT_TRACKED, interpolate the row rotations R[h0], R[h0+1] of its row band
 * (h0 = (int)(y / grid_height)), T = (a R[h0] + (1-a) R[h0+1])^T invK, d = T (x,y,1),
 * norm = d / |d|, vec = norm / norm[2]; fp64 throughout.  rotations = nrot row-major 3x3
 * matrices (System: nrot = frameHeight/20 + 1, grid_height = 20, System.cpp:351-356).  Entries
 * of other features are left as they are.  A feature whose band has no lower vertex row
 * (h0 + 1 >= nrot; the reference reads past `rotations` there) is left untouched as well.
 * gklt_rotate_normalize takes the list from host arrays, gklt_flist_rotate_normalize from a
 * device list (no list transfer); norm3 / vec3 are host arrays of 3*n doubles, written directly
 * by the kernel when both are page-locked. */
int gklt_rotate_normalize(gklt_ctx *ctx, int n, const float *x, const float *y, const int *val,
                          int grid_height, int nrot, const double *rotations, const double *invK,
                          double *norm3, double *vec3);
int gklt_flist_rotate_normalize(gklt_ctx *ctx, gklt_flist *fl, int grid_height, int nrot,
                                const double *rotations, const double *invK, double *norm3,
                                double *vec3);

/* ---- frame pipeline: PyramidBuffer (System.h:30-56) + the call pattern of System::Track ---- */

/* A 3-deep ring of device pyramids plus a device feature list.  gklt_pipeline_step() does what
 * one System::Track call does on the front end (System.cpp:501-518): rotate the ring, build the
 * pyramid of the NEXT frame while tracking previous -> current, and (optionally) deliver the list
 * to host_out.  Steps are asynchronous on the pipeline's own two streams (which alternate by
 * frame parity, one event wait per frame); gklt_pipeline_sync() waits for them. */
typedef struct gklt_pipeline gklt_pipeline;
int gklt_pipeline_create(gklt_ctx *ctx, int width, int height, int capacity, gklt_pipeline **out);
int gklt_pipeline_destroy(gklt_pipeline *p);
/* First frame (System.cpp:367-373): pyramid(img0) becomes "current", pyramid(img1) is preloaded. */
int gklt_pipeline_start(gklt_pipeline *p, const uint8_t *img0, const uint8_t *img1, int stride,
                        int img_on_device);
/* img_next may be NULL for the last frame (track only).  host_out (may be NULL) receives
 * x[capacity] | y[capacity] | val[capacity]; if it is page-locked (gklt_host_malloc_pinned,
 * hipHostMalloc) the tracker kernel writes it directly, otherwise a D2H copy is queued.  A host
 * frame (img_on_device = 0) must stay valid until the step's staging copy has run. */
int gklt_pipeline_step(gklt_pipeline *p, const uint8_t *img_next, int stride, int img_on_device,
                       void *host_out);
/* n consecutive gklt_pipeline_step() calls, frames[i] / host_out[i] being the arguments of call i
 * (host_out may be NULL): exactly the loop a C++ caller runs (System.cpp:501-518), for harnesses
 * whose foreign-function calls cost as much as a step's launches.  No counterpart in the
 * reference. */
int gklt_pipeline_run(gklt_pipeline *p, const uint8_t *const *frames, int n, int stride,
                      int img_on_device, void *const *host_out);
/* Replaces the tracked list (keyframe re-seeding, System.cpp:415-420); ordered with the steps.
 * x / y / val may be pageable and may be reused as soon as the call returns (they are copied
 * into a page-locked mirror of the list first); the same holds for gklt_flist_upload,
 * gklt_lookahead_upload_features and gklt_multistream_upload_features. */
int gklt_pipeline_upload_features(gklt_pipeline *p, int n, const float *x, const float *y,
                                  const int *val);
int gklt_pipeline_sync(gklt_pipeline *p);
/* Waits only for the list of the last step (device list and host_out complete); the pyramid build
 * of the next frame, which the step enqueued after the tracker, keeps running in the background.
 * This is the point at which System::Track can go on (it needs the list, not the next pyramid). */
int gklt_pipeline_wait_tracked(gklt_pipeline *p);
/* The list the pipeline tracks, and its pyramids (which = 0 previous, 1 current) for downloads;
 * call gklt_pipeline_sync() first. */
gklt_flist *gklt_pipeline_flist(gklt_pipeline *p);
gklt_pyramid *gklt_pipeline_pyramid(gklt_pipeline *p, int which);

/* ---- geometric outlier rejection on the tracks ---- */

/* RelativeOutlierRejection::relativeRansac (Estimation.cpp:621-683; System.cpp:550-564 calls it
 * after every KLTTrackFeatures with threshold 10 px and prob 0.98).  Host arrays: key_norm3 =
 * bearing of every feature in the keyframe (mvLocalMapPoints[i].norm), cur_norm3 = its bearing in
 * the current frame (feature[i]->norm, e.g. from gklt_rotate_normalize), n x 3 doubles each; val =
 * the list's status; is_valid = n flags, updated in place; R = rotation keyframe -> current frame
 * (row-major; its transpose is applied to cur_norm3), K = camera matrix (row-major); the view angles
 * are cosines as in SLAMSettings.  As the reference: candidates are the entries with val ==
 * KLT_TRACKED and is_valid; a candidate whose view angle p1.p2 is >= min_view_angle loses is_valid
 * and is skipped; if fewer than half of the candidates lie below med_view_angle *inliers = -1 and
 * nothing else happens; otherwise a fundamental matrix is fitted to the K-projected pixel pairs by
 * RANSAC and the pairs it rejects lose is_valid; *inliers = number of pairs it keeps (0 with fewer
 * than 7 pairs).  The RANSAC (cv::findFundamentalMat in the reference -- OpenCV's sampler is not
 * part of the reference tree, so its random choices cannot be reproduced) is this library's:
 * 7-point minimal solver, larger squared point-to-epipolar-line distance <= threshold^2, adaptive
 * iteration count for confidence `prob` capped at max_iters (OpenCV: 1000), counter-based sampler
 * seeded by `seed`; all max_iters hypotheses are evaluated in parallel and the sequential
 * bookkeeping is replayed on the host, so the result is that of the sequential algorithm
 * (oracle/ransac_oracle.c).  Parity with OpenCV is UNPINNED here in one more respect: OpenCV's
 * findFundamentalMat is believed to switch FM_RANSAC to LMedS below 15 point pairs (its source is
 * not in the reference tree); this library runs threshold RANSAC for every m >= 7.
 * F_out (may be NULL): the best matrix, pixels, x2^T F x1 = 0, row-major;
 * iters_out (may be NULL): hypotheses the adaptive loop examined.  Synchronous. */
int gklt_relative_ransac(gklt_ctx *ctx, int n, const double *key_norm3, const double *cur_norm3,
                         const int *val, uint8_t *is_valid, const double *R, const double *K,
                         double min_view_angle, double med_view_angle, double threshold, double prob,
                         int max_iters, uint64_t seed, int *inliers, double *F_out, int *iters_out);

/* ---- frame ingest (main.cpp:77-107, System.cpp:310-315) ---- */

/* A decoded video frame on the GPU: the 8-bit Y plane that computePyramid reads, and the
 * half-resolution RGB image System keeps for point colours.  The reference gets frames as NV12
 * pixel buffers (plane 0 = Y, W x H; plane 1 = interleaved chroma, W/2 x H/2 two-byte samples),
 * copies the Y plane (image.copyTo, main.cpp:99,101) and builds the colour image on the host
 * with cv::split / cv::resize / cv::merge / cv::cvtColor(CV_YCrCb2RGB) (main.cpp:84-93).
 * gklt_frame_ingest_nv12 moves both planes H2D once (1.5 bytes per pixel; on_device = 1: the planes
 * already are device memory, e.g. a hardware decoder's output, and are borrowed until consumed) and
 * forms the colour image with one kernel: Y' = 2x2 area mean (a+b+c+d+2)>>2, the two chroma bytes
 * in memory order as channels 1 and 2, OpenCV's 8-bit YCrCb->RGB fixed-point formula (channel 1
 * read as Cr, as the reference's merge order makes it).  gklt_frame_ingest_gray takes a Y plane
 * only (no colour image).  Asynchronous on the context's pyramid stream; a frame is re-ingested
 * only after the build that consumed it has finished reading (event-ordered), so two frames
 * alternate without host waits.  width and height must be even. */
int gklt_frame_create(gklt_ctx *ctx, int width, int height, gklt_frame **out);
int gklt_frame_destroy(gklt_frame *f);
int gklt_frame_ingest_nv12(gklt_ctx *ctx, gklt_frame *f, const uint8_t *y, int y_stride,
                           const uint8_t *uv, int uv_stride, int on_device);
int gklt_frame_ingest_gray(gklt_ctx *ctx, gklt_frame *f, const uint8_t *y, int y_stride,
                           int on_device);
/* computePyramid(tc, frame's Y plane, pyramid) (pyramid.cpp:45), ordered after the ingest. */
int gklt_pyramid_build_frame(gklt_ctx *ctx, gklt_pyramid *pyr, gklt_frame *f);
/* gklt_pipeline_step with the next frame taken from `next` (ordered after its ingest). */
int gklt_pipeline_step_frame(struct gklt_pipeline *p, gklt_frame *next, void *host_out);
/* Device address and row stride of the ingested Y plane (for the img_on_device = 1 entry points);
 * waits for the ingest. */
int gklt_frame_luma(gklt_ctx *ctx, gklt_frame *f, const uint8_t **device_ptr, int *stride);
/* The colour image, (H/2) rows of (W/2) RGB triples, to host; stride in bytes.  Synchronous. */
int gklt_frame_download_color(gklt_ctx *ctx, gklt_frame *f, uint8_t *rgb, int stride);
/* Point colours at a keyframe (NormalizeKeyFrameInvoker, System.cpp:310-315): for every entry of
 * the device list with val >= 0, rgb_out[3i..3i+2] = colour image at (round(y/2), round(x/2))
 * (indices clamped to the image; the reference does not test them); (0,0,0) for dead entries.
 * rgb_out: 3 * size bytes of host memory.  Synchronous. */
int gklt_flist_colors(gklt_ctx *ctx, gklt_frame *f, gklt_flist *fl, uint8_t *rgb_out);

/* ---- lookahead pipeline: the same frame loop with a deeper preload ---- */

/* The reference preloads ONE pyramid ahead on a std::thread (System.cpp:508-511,
 * PyramidBuffer System.h:30-56).  A caller that has the next frames at hand (recorded video, a
 * capture queue) can hand over up to max_batch (<= GKLT_MAX_LOOKAHEAD) frames per call:
 * gklt_lookahead_submit() builds their pyramids with one launch per level for the whole batch
 * and, overlapped with that, tracks the batch submitted by the PREVIOUS call with a single
 * launch in which each feature is followed through all frames of that batch.  The lists are those
 * of frame-at-a-time gklt_pipeline_step() calls, bit for bit; they arrive one call later.
 *   start(img0)                       pyramid of frame 0; the list is frame 0's (upload_features)
 *   submit(frames[0..n), host_out[])  frames built+1 .. built+n; host_out[i] (may be NULL, same
 *                                     layout and pinning rule as gklt_pipeline_step) receives the
 *                                     list after frame i of THIS batch, written when the batch is
 *                                     tracked: by the next submit()/flush(), so the buffers must
 *                                     stay valid until then
 *   flush()                           tracks what is still pending and synchronises
 *   upload_features()                 replaces the list before the next tracker launch, i.e. from
 *                                     the first frame that has been submitted but not yet tracked */
typedef struct gklt_lookahead gklt_lookahead;
int gklt_lookahead_create(gklt_ctx *ctx, int width, int height, int capacity, int max_batch,
                          gklt_lookahead **out);
int gklt_lookahead_destroy(gklt_lookahead *p);
int gklt_lookahead_start(gklt_lookahead *p, const uint8_t *img0, int stride, int img_on_device);
int gklt_lookahead_submit(gklt_lookahead *p, const uint8_t *const *frames, int n, int stride,
                          int img_on_device, void *const *host_out);
int gklt_lookahead_flush(gklt_lookahead *p);
int gklt_lookahead_upload_features(gklt_lookahead *p, int n, const float *x, const float *y,
                                   const int *val);
int gklt_lookahead_sync(gklt_lookahead *p);
gklt_flist *gklt_lookahead_flist(gklt_lookahead *p);
/* Index of the last frame the list has been tracked to (enqueued, not necessarily finished). */
long gklt_lookahead_frames_tracked(gklt_lookahead *p);

/* ---- several independent streams on one GPU, stepped together (BASELINE config 4) ---- */

/* nstreams (<= GKLT_MAX_STREAMS) independent video streams of the same size, each with the state
 * of a gklt_pipeline (a 3-deep pyramid ring as PyramidBuffer, System.h:30-56, and a device feature
 * list), advanced one frame per call as System::Track does (System.cpp:501-518): the pyramids of
 * the streams' NEXT frames are built while every list is tracked previous -> current.  The work of
 * a step is batched over the streams instead of over time -- one launch per pyramid level for up to
 * GKLT_MAX_LOOKAHEAD streams, one tracker launch for all lists -- so live streams, which cannot be
 * looked ahead, get the launch structure of gklt_lookahead without its one-batch delay.  Every
 * stream's lists are those of its own gklt_pipeline, bit for bit.  Streams are independent (the
 * multi-GPU partition of SURVEY 8e is whole streams -> GPUs; this object holds one GPU's share).
 * The work is enqueued on the context's own two HIP streams (a process has few hardware queues;
 * objects that bring streams of their own end up sharing them), so one context = one
 * gklt_multistream at a time.
 *   imgs / imgs_next   nstreams frame pointers (all host or all device), one row stride
 *   host_out           NULL, or x[S*capacity] | y[S*capacity] | val[S*capacity] with stream s at
 *                      offset s*capacity of each part; pinning rule as gklt_pipeline_step */
typedef struct gklt_multistream gklt_multistream;
int gklt_multistream_create(gklt_ctx *ctx, int nstreams, int width, int height, int capacity,
                            gklt_multistream **out);
int gklt_multistream_destroy(gklt_multistream *m);
int gklt_multistream_start(gklt_multistream *m, const uint8_t *const *imgs0,
                           const uint8_t *const *imgs1, int stride, int img_on_device);
/* imgs_next may be NULL for the last frame (track only). */
int gklt_multistream_step(gklt_multistream *m, const uint8_t *const *imgs_next, int stride,
                          int img_on_device, void *host_out);
/* Replaces the list of one stream (ordered with the steps). */
int gklt_multistream_upload_features(gklt_multistream *m, int stream, int n, const float *x,
                                     const float *y, const int *val);
/* Blocking download of one stream's list (n entries each). */
int gklt_multistream_download(gklt_multistream *m, int stream, float *x, float *y, int *val);
int gklt_multistream_size(const gklt_multistream *m, int stream);
int gklt_multistream_sync(gklt_multistream *m);
int gklt_multistream_wait_tracked(gklt_multistream *m);
/* Counters over all streams since the last reset, as gklt_flist_stats. */
int gklt_multistream_stats(gklt_multistream *m, uint64_t *tracked, uint64_t *iterations, int reset);
/* Pyramid of one stream (which = 0 previous, 1 current) for downloads; sync first. */
gklt_pyramid *gklt_multistream_pyramid(gklt_multistream *m, int stream, int which);

/* ---- memory helpers for callers that have no HIP of their own (no reference counterpart) ---- */

/* Device buffer on the context's GPU (e.g. decoded frames handed to gklt_pipeline_step with
 * img_on_device = 1), page-locked host buffer (frame staging, host_out), blocking H2D copy, and
 * a device-wide synchronise. */
int gklt_device_malloc(gklt_ctx *ctx, size_t bytes, void **out);
int gklt_device_free(gklt_ctx *ctx, void *ptr);
int gklt_host_malloc_pinned(gklt_ctx *ctx, size_t bytes, void **out);
int gklt_host_free_pinned(gklt_ctx *ctx, void *ptr);
int gklt_copy_to_device(gklt_ctx *ctx, void *dst, const void *src, size_t bytes);
int gklt_device_synchronize(gklt_ctx *ctx);
/* hipMemGetInfo of the context's GPU (tests: the allocation-free entry point allocates nothing). */
int gklt_device_mem_info(gklt_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* ---- measurement ---- */

/* Tracker mapping: 0 = one feature per wavefront; sampling chosen by window size: footprint in
 * registers (footprint <= 64 texels, e.g. 7x7), else the footprint staged in LDS (a lane per
 * window-row run for windows 15..21 wide) (default); 1 = one feature per 8-lane slice; 2 / 3 = per
 * wavefront with four gathers per element / with staging and a lane per element.
 * All produce identical results; selectable for A/B timing. */
int gklt_ctx_set_track_impl(gklt_ctx *ctx, int impl);
/* Wavefronts per wave-per-feature tracker launch; each takes features w, w + waves, ...  The
 * default (0) is one wavefront per feature for single-frame launches and, for the multi-frame
 * launches of the lookahead pipeline, a cap of 4096 wavefronts (4 per SIMD), so that the pyramid
 * build it overlaps with always finds room.
 * Results do not depend on it. */
int gklt_ctx_set_track_waves(gklt_ctx *ctx, int waves);
/* Pyramid build: 0 = one fused launch per level (u8 -> smooth -> texels + next level; 64 x 32
 * tiles) when the kernel radii are the reference's defaults, separate kernels otherwise
 * (default); 1 = always the separate smooth / decimate / gradient kernels; 2 = as 0; 3 = fused
 * launches take the streaming kernel (one wavefront per 112-column strip marching down the rows,
 * column windows in registers) where it applies (frames of at least 192 x 192, level-0 windows
 * 7x7 / 11x11 / 21x21).  Identical results for 0..3.  4 = the fused kernels of 0 with every
 * mul+add contracted into an FMA: NOT the reference's unfused AVX arithmetic any more -- planes
 * agree with it to 1e-4 relative (the tolerance this path is specified to), positions and status
 * flags of a few features per thousand differ in the last digits.  Never the default. */
int gklt_ctx_set_pyramid_impl(gklt_ctx *ctx, int impl);
/* A/B switch (default 1): a single-frame build runs levels 1 .. L-1 in ONE launch whose tiles wait
 * for the tiles of the level above through per-tile flags, instead of one launch per level
 * (0: one launch per level; 2: batched builds take the chained launch too -- measured slower).
 * Same results. */
int gklt_ctx_set_chain(gklt_ctx *ctx, int on);
/* Lookahead pipeline (gklt_lookahead_*): 1 = the coarse levels of a batch run on a second build
 * stream behind level 0, so that level 0 of the next batch overlaps them (they are short launches
 * that fill the chip less and less); 0 = the whole build on one stream.  Same results; no
 * counterpart in the reference (its preload thread builds one pyramid at a time, System.h:30-56). */
int gklt_ctx_set_coarse_split(gklt_ctx *ctx, int on);
/* Suppression rounds of the selectors (plain / ...2, and the uncapped phase of the Uniform one):
 * 0 (default) = rounds on box tops -- one candidate per (r+1) x (r+1) box is examined instead of
 * every pixel -- for radii 4..~45, dense rounds on 32-bit keys otherwise; 1 = dense rounds on
 * 64-bit keys; 2 = dense rounds on 32-bit keys.  Identical results. */
int gklt_ctx_set_select_impl(gklt_ctx *ctx, int impl);
/* Candidate scores of gklt_select_from_pyramid / gklt_seed_from_level: 0 (default) = from fp32
 * integral images in cv::integral's order, i.e. what the reference's selector is handed
 * (System.cpp:385-392); 1 = from window sums formed directly (separable box sum of gx^2, gy^2,
 * gx gy over the window: each window row left to right, the row sums top to bottom) -- no
 * counterpart in the reference: no integral images, no cancellation between sums ~1e9, windows up
 * to 33 px.  Scores then differ from mode 0 in the last bits, so selections are compared by set
 * overlap (bench.py --config C5 reports it); against the oracle's direct form they are identical. */
int gklt_ctx_set_select_scores(gklt_ctx *ctx, int mode);
/* Selector: fixpoint iterations a tile runs in LDS per suppression launch (1..64; default 3 for the
 * dense rounds, 12 for the rounds on box tops; 1 = one iteration per launch).  Results do not
 * depend on it. */
int gklt_ctx_set_select_iters(gklt_ctx *ctx, int iters);
/* Suppression launches the last selection on this context needed until no candidate was left
 * undecided: rounds[0] = plain / ...2 selector or phase A (no per-cell caps) of the Uniform
 * one, rounds[1] = phase B of the Uniform one (0 otherwise). */
int gklt_ctx_select_rounds(gklt_ctx *ctx, int rounds[2]);

enum {
  GKLT_K_SMOOTH0 = 0, /* u8 -> smoothed level 0 */
  GKLT_K_DOWN = 1,    /* pre-decimation smooth + 2x decimation (all levels) */
  GKLT_K_GRAD = 2,    /* gradients + texel packing (all levels) */
  GKLT_K_TRACK = 3,
  GKLT_K_MISC = 4,
  GKLT_K_INTEGRAL = 5, /* products + integral images */
  GKLT_K_SELECT = 6,   /* score, suppression rounds, rank/assign */
  GKLT_K_LEVEL0 = 7,   /* fused level 0: u8 -> smooth -> texels + level 1 */
  GKLT_K_LEVELN = 8,   /* fused level l > 0: I_l -> texels + level l+1 */
  GKLT_K_RANSAC = 9,   /* outlier rejection: prepare, hypotheses, scoring */
  GKLT_K_COUNT = 10
};
/* When enabled, every kernel launch is bracketed by HIP events on its own stream. */
int gklt_profile_enable(gklt_ctx *ctx, int on);
/* Synchronises, folds finished event pairs into per-kernel totals, returns them. */
int gklt_profile_get(gklt_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
int gklt_profile_reset(gklt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
