// stand-in for convolve.h:12-13
#pragma once
#include <thread>
#include "KLT.h"
void computeGradients(const cv::Mat &img, cv::Mat &gradx, cv::Mat &grady, float sigma, std::thread *threads = NULL);
void computeSmoothedImage(const cv::Mat &img, cv::Mat &dst, float sigma);
