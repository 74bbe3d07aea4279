// Test stand-in for GSLAM's KLT.h (/root/reference/GSLAM/KLT.h:18-120): the type and field NAMES
// that gklt_shim.cpp is written against, declared afresh so the shim can be compiled and run where
// the reference tree is absent (the GPU box).  A GSLAM build uses the reference's own KLT.h.
#pragma once
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
