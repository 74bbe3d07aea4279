// Test stand-in for the handful of OpenCV core types gklt_shim.cpp touches (cv::Mat as
// {rows, cols, data, step, create, ptr, at}, cv::Size, cv::Point2f).  OpenCV is not installed in the
// build image; this header exists only so that tests/test_shim.py can compile AND run the shim.
// It is not part of the product and implements no image processing.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F

typedef unsigned char uchar;

namespace cv {

struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};

struct Point2f {
  float x, y;
};

class Mat {
 public:
  int rows = 0, cols = 0;
  uchar *data = nullptr;
  size_t step = 0;  // bytes per row (a MatStep in OpenCV, convertible to size_t)

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void *ext, size_t ext_step = 0)  // external data, not owned
      : rows(r), cols(c), data(static_cast<uchar *>(ext)), step(ext_step ? ext_step : c * esz(type)), type_(type) {}

  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && own_) return;
    rows = r;
    cols = c;
    type_ = type;
    step = (size_t)c * esz(type);
    own_.reset(static_cast<uchar *>(std::malloc(step * (size_t)r + 64)), std::free);
    data = own_.get();
  }
  void create(Size s, int type) { create(s.height, s.width, type); }
  Size size() const { return Size(cols, rows); }
  int type() const { return type_; }
  bool empty() const { return rows == 0 || cols == 0; }
  template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
  template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
  template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }

 private:
  static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<uchar> own_;
};

}  // namespace cv
