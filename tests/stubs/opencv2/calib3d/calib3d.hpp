// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
void stereoRectify(const Mat &K1, const Mat &D1, const Mat &K2, const Mat &D2, Size imageSize, const Mat &R, const Mat &T,
                   Mat &R1, Mat &R2, Mat &P1, Mat &P2, Mat &Q);
}
