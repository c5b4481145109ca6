// tflite_shim.hpp — the slice of the TensorFlow-Lite C/C++ API that the reference's in-tree sources touch
// (lib/libbackscrub.cc, lib/transpose_conv_bias.{h,cc}), so that those sources compile UNMODIFIED in an image
// without TensorFlow.  TEST INFRASTRUCTURE (part of oracle/).  The Interpreter declared here executes the graph
// with the oracle's restatement of the TFLite builtin kernels (oracle/bs_oracle.cpp §2) and dispatches every CUSTOM
// operator through the TfLiteRegistration the reference registered (AddCustom) — i.e. Convolution2DTransposeBias
// runs the reference's own compiled Prepare/Eval/TransposeConvBias.  Names and field meanings follow the public
// TFLite headers (tensorflow/lite/c/common.h, kernels/kernel_util.h, kernels/internal/types.h, v2.8.0).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#define TFLITE_VERSION_STRING "2.8.0 (ref_shim: builtin kernels = oracle restatement)"

extern "C" {
typedef enum { kTfLiteOk = 0, kTfLiteError = 1 } TfLiteStatus;
typedef enum { kTfLiteNoType = 0, kTfLiteFloat32 = 1, kTfLiteInt32 = 2, kTfLiteUInt8 = 3, kTfLiteFloat16 = 10 } TfLiteType;
typedef enum { kTfLitePaddingUnknown = 0, kTfLitePaddingSame, kTfLitePaddingValid } TfLitePadding;
typedef struct { int size; int data[8]; } TfLiteIntArray;
typedef struct { int width, height, width_offset, height_offset; } TfLitePaddingValues;
typedef struct { TfLitePadding padding; int stride_width; int stride_height; } TfLiteTransposeConvParams;
typedef union { float* f; void* raw; } TfLitePtrUnion;
typedef struct TfLiteTensor { TfLiteType type; TfLitePtrUnion data; TfLiteIntArray* dims; } TfLiteTensor;
typedef struct TfLiteNode { TfLiteIntArray* inputs; TfLiteIntArray* outputs; const void* custom_initial_data; int custom_initial_data_size; } TfLiteNode;
typedef struct TfLiteContext {
  TfLiteTensor* tensors;
  TfLiteStatus (*ResizeTensor)(struct TfLiteContext*, TfLiteTensor* tensor, TfLiteIntArray* new_size);   // takes ownership of new_size
  void (*ReportError)(struct TfLiteContext*, const char* msg, ...);
} TfLiteContext;
typedef struct TfLiteRegistration {
  void* (*init)(TfLiteContext*, const char*, size_t);
  void (*free)(TfLiteContext*, void*);
  TfLiteStatus (*prepare)(TfLiteContext*, TfLiteNode*);
  TfLiteStatus (*invoke)(TfLiteContext*, TfLiteNode*);
} TfLiteRegistration;
}
inline TfLiteIntArray* TfLiteIntArrayCreate(int size) { TfLiteIntArray* a = (TfLiteIntArray*)calloc(1, sizeof(TfLiteIntArray)); a->size = size; return a; }
inline void TfLiteIntArrayFree(TfLiteIntArray* a) { free(a); }

#define TF_LITE_ENSURE(context, a) do { if (!(a)) { (context)->ReportError((context), "%s:%d %s was not true.", __FILE__, __LINE__, #a); return kTfLiteError; } } while (0)
#define TF_LITE_ENSURE_EQ(context, a, b) do { if ((a) != (b)) { (context)->ReportError((context), "%s:%d %s != %s", __FILE__, __LINE__, #a, #b); return kTfLiteError; } } while (0)
#define TF_LITE_ENSURE_OK(context, status) do { const TfLiteStatus s_ = (status); if (s_ != kTfLiteOk) return s_; } while (0)
#define TFLITE_DCHECK_EQ(a, b) do { if ((a) != (b)) { fprintf(stderr, "DCHECK failed %s:%d %s == %s\n", __FILE__, __LINE__, #a, #b); abort(); } } while (0)

namespace tflite {

inline int NumInputs(const TfLiteNode* n) { return n->inputs->size; }
inline int NumOutputs(const TfLiteNode* n) { return n->outputs->size; }
inline const TfLiteTensor* GetInput(const TfLiteContext* c, const TfLiteNode* n, int i) { return &c->tensors[n->inputs->data[i]]; }
inline TfLiteTensor* GetOutput(TfLiteContext* c, const TfLiteNode* n, int i) { return &c->tensors[n->outputs->data[i]]; }
inline int NumDimensions(const TfLiteTensor* t) { return t->dims->size; }
inline int SizeOfDimension(const TfLiteTensor* t, int d) { return t->dims->data[d]; }

class RuntimeShape {
 public:
  RuntimeShape() = default;
  RuntimeShape(int n, const int* d) : n_(n) { for (int i = 0; i < n; i++) d_[i] = d[i]; }
  int DimensionsCount() const { return n_; }
  int Dims(int i) const { return d_[i]; }
 private:
  int n_ = 0, d_[8] = {0};
};
inline RuntimeShape GetTensorShape(const TfLiteTensor* t) { return t ? RuntimeShape(t->dims->size, t->dims->data) : RuntimeShape(); }
template <typename T> inline T* GetTensorData(TfLiteTensor* t) { return t ? (T*)t->data.raw : nullptr; }
template <typename T> inline const T* GetTensorData(const TfLiteTensor* t) { return t ? (const T*)t->data.raw : nullptr; }
inline int Offset(const RuntimeShape& s, int i0, int i1, int i2, int i3) { return ((i0 * s.Dims(1) + i1) * s.Dims(2) + i2) * s.Dims(3) + i3; }
inline int MatchingDim(const RuntimeShape& a, int ia, const RuntimeShape& b, int ib) { TFLITE_DCHECK_EQ(a.Dims(ia), b.Dims(ib)); return a.Dims(ia); }

enum class PaddingType : uint8_t { kNone, kSame, kValid };
struct PaddingValues { int16_t width = 0, height = 0, width_offset = 0, height_offset = 0; };
struct ConvParams { PaddingType padding_type = PaddingType::kNone; PaddingValues padding_values; int16_t stride_width = 0, stride_height = 0; };

class FlatBufferModel {
 public:
  static std::unique_ptr<FlatBufferModel> BuildFromFile(const char* filename);
  ~FlatBufferModel();
  void* oracle_model = nullptr;     // OModel handle (oracle/bs_oracle.cpp)
};

namespace ops { namespace builtin {
class BuiltinOpResolver {
 public:
  void AddCustom(const char* name, const TfLiteRegistration* reg) { custom[name] = reg; }
  std::map<std::string, const TfLiteRegistration*> custom;
};
} }

class Interpreter {
 public:
  ~Interpreter();
  TfLiteStatus AllocateTensors();
  void SetNumThreads(int) {}
  void SetAllowFp16PrecisionForFp32(bool) {}
  TfLiteTensor* tensor(int i) { return &tensors_[i]; }
  template <typename T> T* typed_tensor(int i) { return (T*)tensors_[i].data.raw; }
  const std::vector<int>& inputs() const { return inputs_; }
  const std::vector<int>& outputs() const { return outputs_; }
  TfLiteStatus Invoke();
  // shim state
  void* oracle_model = nullptr;
  std::map<std::string, const TfLiteRegistration*> custom;
  std::vector<TfLiteTensor> tensors_;
  std::vector<TfLiteIntArray> dims_;
  std::vector<int> inputs_, outputs_;
  std::vector<float> in_buf_, out_buf_;
};

class InterpreterBuilder {
 public:
  InterpreterBuilder(const FlatBufferModel& m, const ops::builtin::BuiltinOpResolver& r) : m_(m), r_(r) {}
  TfLiteStatus operator()(std::unique_ptr<Interpreter>* out);
 private:
  const FlatBufferModel& m_;
  const ops::builtin::BuiltinOpResolver& r_;
};

}  // namespace tflite
