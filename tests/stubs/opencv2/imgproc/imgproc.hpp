// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
void medianBlur(const Mat &src, Mat &dst, int ksize);
}
