// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <opencv2/core/core.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
struct CvImage { cv::Mat image; };
typedef std::shared_ptr<CvImage> CvImagePtr;
CvImagePtr toCvCopy(const sensor_msgs::Image &source, const std::string &encoding = std::string());
}  // namespace cv_bridge
