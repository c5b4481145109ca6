// tflite_model.hpp — product-side .tflite (schema v3) graph loader.
//
// Replaces tflite::FlatBufferModel::BuildFromFile + InterpreterBuilder
// (/root/reference/lib/libbackscrub.cc:190,205-217).  Produces a flat list of graph
// nodes with every constant already materialised as f32 (the f16→f32 DEQUANTIZE ops of
// the Google models are folded at load), ready for the fusing planner in plan.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bsx {

enum class OpType : int {
  Add = 0, AvgPool = 1, Concat = 2, Conv = 3, DwConv = 4, Dequantize = 6, FullyConnected = 9, Logistic = 14,
  Mul = 18, Relu = 19, Relu6 = 21, ResizeBilinear = 23, Custom = 32, HardSwish = 117,
  TransposeConvBias = 1000,  // CUSTOM "Convolution2DTransposeBias"
};

enum Activation : int { kActNone = 0, kActRelu = 1, kActRelu6 = 3, kActHswish = 100, kActSigmoid = 101 };

struct TensorInfo {
  std::vector<int> shape;          // as stored
  int dims[4] = {1, 1, 1, 1};      // NHWC, left-padded with 1
  bool is_const = false;
  std::vector<float> f32;          // constant payload (f32 or dequantised f16)
  std::vector<int32_t> i32;        // constant payload (int32, e.g. resize sizes)
  std::string name;
  size_t elems() const { return (size_t)dims[0] * dims[1] * dims[2] * dims[3]; }
};

struct Node {
  OpType type;
  std::vector<int> inputs;         // tensor ids (-1 = absent optional)
  int output = -1;
  int index = -1;                  // operator index in the file (for messages)
  // options (superset)
  bool same_padding = true;
  int stride_h = 1, stride_w = 1, dil_h = 1, dil_w = 1, depth_mult = 1;
  int filter_h = 0, filter_w = 0;
  int act = kActNone;
  int axis = 3;
  bool align_corners = false, half_pixel = false;
  int tconv_padding_same = 1, tconv_stride_h = 2, tconv_stride_w = 2;
};

struct Graph {
  std::vector<TensorInfo> tensors;
  std::vector<Node> nodes;         // executable nodes only (constant-only ops are folded away)
  int input = -1, output = -1;
  int n_file_ops = 0;
  std::string description;
};

// Returns false and fills `err` on any malformed / unsupported content.
bool load_tflite(const std::string& path, Graph* g, std::string* err);

}  // namespace bsx
