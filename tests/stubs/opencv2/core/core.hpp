// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <cstdint>
#define CV_8U 0
#define CV_MAJOR_VERSION 4
#define CV_64F 6
#define CV_64FC1 6
namespace cv {
struct Size { int width = 0, height = 0; Size(); Size(int w, int h); };
struct MatExpr;
struct Mat {
  int rows = 0, cols = 0;
  const uint8_t *datastart = nullptr, *dataend = nullptr;
  Mat();
  Mat(Size s, int type);
  Mat(const MatExpr &e);
  Size size() const;
  Mat clone() const;
  void convertTo(Mat &dst, int type) const;
  template <class T> T *ptr(int row = 0);
  static MatExpr zeros(int rows, int cols, int type);
  static MatExpr eye(int rows, int cols, int type);
};
struct MatExpr { operator Mat() const; };
template <class T> struct MatCommaInitializer_ {
  MatCommaInitializer_ &operator,(T v);
  operator Mat() const;
};
template <class T> struct Mat_ : Mat {
  Mat_(int rows, int cols);
};
template <class T> MatCommaInitializer_<T> operator<<(const Mat_<T> &m, T v);
}  // namespace cv
