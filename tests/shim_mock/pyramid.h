// stand-in for pyramid.h:11
#pragma once
#include "convolve.h"
#include "opencv2/imgproc/imgproc.hpp"
void computePyramid(KLT_TrackingContext tc, const cv::Mat &img, std::vector<cv::Mat> &pyramid);
