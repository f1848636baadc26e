// Test stub, build container only: just enough of OpenCV's names for a -fsyntax-only compile of
// mono_slam_framework_amd/csrc/hip_feature_matcher.h against the reference's own
// slam_pipeline/include/FeatureMatcher.h (tests/test_adapter_syntax.py).  Nothing here is linked or run.
#pragma once
#include <cstddef>
#define CV_8UC1 0
namespace cv {
template <class T> struct Point_ {
  T x{}, y{};
  Point_() = default;
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
struct Mat {
  unsigned char* data = nullptr;
  int cols = 0, rows = 0;
  size_t step = 0;
  int type() const { return CV_8UC1; }
};
struct SparseMat {};
}  // namespace cv
