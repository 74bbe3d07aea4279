// Test stand-in for the part of GSLAM's own headers that shim/gklt_shim.cpp is written against
// (types and field names of KLT.h:18-120; the entry points declared by KLTUtil.h:12-16,
// convolve.h:12-13, pyramid.h:11, error.h), declared afresh in ONE file so that the shim can be
// compiled and run where the reference tree is absent (the GPU box).  The files KLT.h, KLTUtil.h,
// convolve.h, pyramid.h and error.h of this directory only forward to it, because those are the
// names the shim includes.  A GSLAM build uses the reference's own headers instead.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include <thread>
#include <vector>

#include "Eigen/Dense"
#include "opencv2/core/core.hpp"

struct KLT_TrackingContextRec {
  float lighting_alpha, lighting_beta;
  int mindist, window_width, window_height;
  bool sequentialMode, smoothBeforeSelecting, writeInternalImages, lighting_insensitive;
  int min_eigenvalue;
  float min_determinant, min_displacement;
  int max_iterations;
  float max_residue, grad_sigma, smooth_sigma_fact, pyramid_sigma_fact, step_factor;
  int nSkippedPixels, borderx, bordery, nPyramidLevels, featureSelectionLevel, subsampling;
  int affine_window_width, affine_window_height, affineConsistencyCheck, affine_max_iterations;
  float affine_max_residue, affine_min_displacement, affine_max_displacement_differ;
};
typedef KLT_TrackingContextRec *KLT_TrackingContext;

enum { KLT_TRACKED = 0, KLT_NOT_FOUND = -1, KLT_SMALL_DET = -2, KLT_MAX_ITERATIONS = -3, KLT_OOB = -4,
       KLT_LARGE_RESIDUE = -5 };

struct KLT_FeatureRec {
  float x, y;
  int val;
  cv::Point2f pt;
  Eigen::Vector3d norm, vec;
  float aff_x, aff_y, aff_Axx, aff_Ayx, aff_Axy, aff_Ayy;
  cv::Mat *aff_img, *aff_gradx, *aff_grady;
  cv::Mat coefficient;
};
typedef KLT_FeatureRec *KLT_Feature;

struct KLT_FeatureListRec {
  bool isOutlierRejected;
  int nFeatures;
  KLT_Feature *feature;
};
typedef KLT_FeatureListRec *KLT_FeatureList;

#define MAX_KERNEL_WIDTH 71
struct ConvolutionKernel {
  int width;
  float data[MAX_KERNEL_WIDTH];
};

// ---- entry points the shim defines (signatures are the reference's: the drop-in contract) ----
void KLTError(char *fmt, ...);    // prints and exits
void KLTWarning(char *fmt, ...);  // prints
KLT_TrackingContext KLTCreateTrackingContext();
KLT_FeatureList KLTCreateFeatureList(KLT_TrackingContext tc, int nFeatures);
float _KLTComputeSmoothSigma(KLT_TrackingContext tc);
void computeKernels(float sigma, ConvolutionKernel *gauss, ConvolutionKernel *gaussderiv);
void computeSmoothedImage(const cv::Mat &img, cv::Mat &dst, float sigma);
void computeGradients(const cv::Mat &img, cv::Mat &gradx, cv::Mat &grady, float sigma, std::thread *threads = NULL);
void computePyramid(KLT_TrackingContext tc, const cv::Mat &img, std::vector<cv::Mat> &pyramid);
