// Minimal stand-in for <opencv2/core/core.hpp> so that csrc/bs_maskgen_shim.cpp can be compiled
// and exercised where OpenCV is not installed (this image).  Only the public cv::Mat surface the
// shim touches is provided: rows, cols, data, step[0], type(), empty(), the (rows, cols, type)
// and (rows, cols, type, data, step) constructors.  With a real OpenCV on the include path this
// directory is simply not used.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

#define CV_8U 0
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)

namespace cv {
class Mat {
 public:
  struct Step { size_t v[2] = {0, 0}; size_t& operator[](int i) { return v[i]; } const size_t& operator[](int i) const { return v[i]; } };
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  Step step;
  Mat() = default;
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t) {
    int cn = (t >> CV_CN_SHIFT) + 1;
    own_.reset(new uint8_t[(size_t)r * c * cn]);
    data = own_.get(); step[0] = (size_t)c * cn; step[1] = (size_t)cn;
  }
  Mat(int r, int c, int t, void* d, size_t s = 0) : rows(r), cols(c), data((uint8_t*)d), type_(t) {
    int cn = (t >> CV_CN_SHIFT) + 1;
    step[0] = s ? s : (size_t)c * cn; step[1] = (size_t)cn;
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
 private:
  int type_ = 0;
  std::shared_ptr<uint8_t[]> own_;
};
}  // namespace cv
