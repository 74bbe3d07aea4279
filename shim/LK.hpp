// LK.hpp of the gklt shim: declarations only.  The reference's LK.hpp is a header that DEFINES the
// tracker (includable from one translation unit, System.cpp:13); with libgklt the bodies live in
// gklt_shim.cpp and this header keeps System.cpp's #include "LK.hpp" compiling.
// Signatures: /root/reference/GSLAM/LK.hpp:27 and :972-975.
#pragma once
#include "KLT.h"

int KLTCountRemainingFeatures(KLT_FeatureList fl);
void KLTTrackFeatures(KLT_TrackingContext tc, std::vector<cv::Mat> &pyramid1,
                      std::vector<cv::Mat> &pyramid2, KLT_FeatureList featurelist,
                      const Eigen::Matrix3d &invK);
