// cv_shim.hpp — the slice of the OpenCV C++ API that the reference's in-tree sources touch
// (lib/libbackscrub.cc, app/deepseg.cc:87-134), so that those sources compile UNMODIFIED in an image
// without OpenCV.  TEST INFRASTRUCTURE (part of oracle/): the image operations declared here
// (resize, cvtColor, bilateralFilter, blur, convertTo) are implemented in ref_glue.cpp by calling the
// oracle's restatement of OpenCV (oracle/bs_oracle.cpp §3) — what this build pins is the REFERENCE'S
// OWN code around them (geometry, state, decode + IIR, blend, YUYV packing, call order), not OpenCV.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC2 CV_MAKETYPE(CV_8U, 2)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC(n) CV_MAKETYPE(CV_32F, (n))

namespace cv {

struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} bool operator==(const Size& o) const { return width == o.width && height == o.height; } };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() = default; Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {} Size size() const { return Size(width, height); } };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };

enum { COLOR_BGR2RGB = 4, COLOR_RGB2YUV = 83 };

class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;          // bytes per row

  Mat() = default;
  Mat(int r, int c, int t) { create(r, c, t); }                       // zero-filled here (OpenCV leaves it uninitialised)
  Mat(int r, int c, int t, const Scalar& s) { create(r, c, t); fill(s); }
  Mat(int r, int c, int t, void* d) : rows(r), cols(c), data((uint8_t*)d), type_(t) { step = (size_t)c * elem(); }   // user memory, not owned
  static Mat ones(int r, int c, int t) { Mat m(r, c, t); m.fill(Scalar(1, 1, 1, 1)); return m; }
  static Mat zeros(Size s, int t) { return Mat(s.height, s.width, t); }

  void create(int r, int c, int t) {
    if (data && r == rows && c == cols && t == type_) return;           // cv::Mat::create keeps a matching buffer
    rows = r; cols = c; type_ = t; step = (size_t)c * elem();
    own_.reset((uint8_t*)calloc((size_t)r * step + 16, 1), free);
    data = own_.get();
  }
  Mat operator()(const Rect& r) const {                                // view sharing the buffer
    assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
    Mat v; v.rows = r.height; v.cols = r.width; v.type_ = type_; v.step = step; v.own_ = own_;
    v.data = data + (size_t)r.y * step + (size_t)r.x * elem();
    return v;
  }
  int type() const { return type_; }
  int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
  int depth() const { return type_ & 7; }
  size_t elem() const { return (size_t)channels() * (depth() == CV_32F ? 4 : 1); }
  Size size() const { return Size(cols, rows); }
  size_t total() const { return (size_t)rows * cols; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step == (size_t)cols * elem(); }
  void deallocate() { own_.reset(); data = nullptr; rows = cols = 0; }
  Mat clone() const { Mat m(rows, cols, type_); for (int y = 0; y < rows; y++) memcpy(m.data + y * m.step, data + y * step, (size_t)cols * elem()); return m; }
  void convertTo(Mat& dst, int rtype, double alpha = 1, double beta = 0) const;   // ref_glue.cpp
  void fill(const Scalar& s) {
    for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) for (int c = 0; c < channels(); c++) {
      if (depth() == CV_32F) ((float*)(data + y * step))[x * channels() + c] = (float)s.v[c];
      else (data + y * step)[x * channels() + c] = (uint8_t)s.v[c];
    }
  }

 private:
  int type_ = 0;
  std::shared_ptr<uint8_t> own_;
};

// `cv::Mat::ones(h, w, CV_8UC1) * 255` (lib/libbackscrub.cc:248)
inline Mat operator*(const Mat& m, double k) {
  Mat o = m.clone();
  for (int y = 0; y < o.rows; y++) for (size_t i = 0; i < (size_t)o.cols * o.elem(); i++) { double v = (o.data + y * o.step)[i] * k; (o.data + y * o.step)[i] = (uint8_t)std::min(255.0, std::max(0.0, v)); }
  return o;
}

// imgproc — implemented in ref_glue.cpp on top of the oracle's OpenCV restatement
void resize(const Mat& src, Mat& dst, Size dsize);                       // INTER_LINEAR
void cvtColor(const Mat& src, Mat& dst, int code);
void bilateralFilter(const Mat& src, Mat& dst, int d, double sigmaColor, double sigmaSpace);
void blur(const Mat& src, Mat& dst, Size ksize);
void split(const Mat& src, std::vector<Mat>& planes);

}  // namespace cv
