// selectGoodFeatures.hpp of the gklt shim: declarations only (the reference's header defines the
// selectors and their file-scope globals, selectGoodFeatures.hpp:28-30; it is compiled into
// System.cpp:14).  Signatures: /root/reference/GSLAM/selectGoodFeatures.hpp:424,436,997,1009,
// 1318,1341.  gridSize is the Uniform selector's cell size in pixels (:28, default 20 x 20).
#pragma once
#include "KLT.h"

extern cv::Size gridSize;

void KLTSelectGoodFeatures(KLT_TrackingContext tc, const cv::Mat &sumDxx, const cv::Mat &sumDyy,
                           const cv::Mat &sumDxy, KLT_FeatureList fl);
void KLTReplaceLostFeatures(KLT_TrackingContext tc, const cv::Mat &sumDxx, const cv::Mat &sumDyy,
                            const cv::Mat &sumDxy, KLT_FeatureList fl);
void KLTSelectGoodFeatures2(KLT_TrackingContext tc, const cv::Mat &sumDxx, const cv::Mat &sumDyy,
                            const cv::Mat &sumDxy, KLT_FeatureList fl);
void KLTReplaceLostFeatures2(KLT_TrackingContext tc, const cv::Mat &sumDxx, const cv::Mat &sumDyy,
                             const cv::Mat &sumDxy, KLT_FeatureList fl);
void KLTSelectGoodFeaturesUniform(KLT_TrackingContext tc, const cv::Mat &sumDxx,
                                  const cv::Mat &sumDyy, const cv::Mat &sumDxy, KLT_FeatureList fl);
void KLTReplaceLostFeaturesUniform(KLT_TrackingContext tc, const cv::Mat &sumDxx,
                                   const cv::Mat &sumDyy, const cv::Mat &sumDxy, KLT_FeatureList fl);
