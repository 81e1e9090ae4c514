// stand-in for include/Converter.h (its Eigen / g2o conversions are not on the front-end path): the one function src/Frame.cc
// uses.  Only for the oracle/_ref build of the reference sources.
#pragma once
#include <vector>
#include "opencv2/core/core.hpp"
namespace ORB_SLAM2 { class Converter { public: static std::vector<cv::Mat> toDescriptorVector(const cv::Mat& Descriptors) { std::vector<cv::Mat> v; v.reserve(Descriptors.rows); for (int j = 0; j < Descriptors.rows; j++) v.push_back(Descriptors.row(j)); return v; } }; }
