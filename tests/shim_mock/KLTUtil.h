// stand-in for KLTUtil.h:12-16 (see KLT.h in this directory)
#pragma once
#include "KLT.h"
void computeKernels(float sigma, ConvolutionKernel *gauss, ConvolutionKernel *gaussderiv);
float _KLTComputeSmoothSigma(KLT_TrackingContext tc);
KLT_TrackingContext KLTCreateTrackingContext();
KLT_FeatureList KLTCreateFeatureList(KLT_TrackingContext tc, int nFeatures);
