// ref_glue.cpp — builds oracle/_ref/libbs_ref.so: the REFERENCE'S OWN in-tree sources, compiled unmodified from
// /root/reference (lib/libbackscrub.cc, lib/transpose_conv_bias.cc, and the two file-static helpers of
// app/deepseg.cc:87-134 extracted at build time), on top of cv_shim.hpp / tflite_shim.hpp.
//
// TEST INFRASTRUCTURE (oracle/).  What the resulting library pins, through reference object code:
//   * bs_maskgen_new / bs_maskgen_process / bs_maskgen_delete glue: model-type sniffing, normalisation constants, the
//     roidim / in_roidim float-to-int geometry, persistent mask / in_u8_bgr state, callback order, mask aliasing
//     (lib/libbackscrub.cc:116-148,161-259,279-376)
//   * the three decode loops + temporal IIR (lib/libbackscrub.cc:317-357)
//   * Convolution2DTransposeBias Prepare/Eval incl. the SAME-padding arithmetic (lib/transpose_conv_bias.cc:37-256)
//   * alpha_blend and convert_rgb_to_yuyv's 4:2:2 packing (app/deepseg.cc:87-134)
// What it does NOT pin (third-party code absent from the checkout, restated by the oracle and called from here):
// cv::resize / cvtColor / bilateralFilter / blur / convertTo and every TFLite builtin kernel.
#include <cstring>
#include <string>
#include <vector>

#include "cv_shim.hpp"
#include "tflite_shim.hpp"
#include "libbackscrub.h"          // /root/reference/lib (include path set by oracle/Makefile)

// ---- the oracle's C API (oracle/libbs_oracle.so) ------------------------------------------------------------------
extern "C" {
void* bso_model_load(const char* path);
void bso_model_free(void* h);
int bso_model_input(void* h);
int bso_model_output(void* h);
int bso_model_num_tensors(void* h);
int bso_model_tensor_shape(void* h, int i, int* out4);
int bso_model_invoke(void* h, const float* in, float* out);
void bso_resize_linear_u8(const uint8_t* src, int sw, int sh, long sstride, int cn, uint8_t* dst, int dw, int dh, long dstride);
void bso_bilateral_c3(const uint8_t* src, int w, int h, uint8_t* dst, int d, double sc, double ss);
void bso_blur5_u8(const uint8_t* src, int w, int h, long sstride, uint8_t* dst, long dstride);
void bso_rgb2yuv_u8(const uint8_t* in, long npix, uint8_t* yuv);
void bso_convert_f32(const uint8_t* in, long n, float scale, float off, float* out);
typedef int (*bso_custom_fn)(const char* name, const unsigned char* opts, int n_opts, const float* x, const int* xs4, const float* w,
                             const int* ws4, const float* b, int nb, float** y_out, int* ys4);
void bso_set_custom_op_hook(bso_custom_fn fn);
}

// ---- cv:: implementations -----------------------------------------------------------------------------------------
namespace cv {

static std::vector<uint8_t> g_last_c1_resize_src;    // tap for the tests: the last 1-channel resize source = ctx.ofinal(in_roidim)
static int g_last_c1_w = 0, g_last_c1_h = 0;

static Mat packed(const Mat& m) { return m.isContinuous() ? m : m.clone(); }

void resize(const Mat& src, Mat& dst, Size dsize) {
  assert(src.depth() == CV_8U);
  dst.create(dsize.height, dsize.width, src.type());     // keeps a matching pre-allocated view (libbackscrub.cc:289 writes into in_u8_bgr)
  if (src.channels() == 1) {
    g_last_c1_w = src.cols; g_last_c1_h = src.rows; g_last_c1_resize_src.resize(src.total());
    for (int y = 0; y < src.rows; y++) memcpy(g_last_c1_resize_src.data() + (size_t)y * src.cols, src.data + y * src.step, src.cols);
  }
  bso_resize_linear_u8(src.data, src.cols, src.rows, (long)src.step, src.channels(), dst.data, dst.cols, dst.rows, (long)dst.step);
}

void cvtColor(const Mat& src, Mat& dst, int code) {
  assert(src.type() == CV_8UC3);
  Mat s = packed(src);
  Mat out(src.rows, src.cols, CV_8UC3);
  if (code == COLOR_BGR2RGB) {
    for (size_t i = 0; i < s.total(); i++) { out.data[3 * i] = s.data[3 * i + 2]; out.data[3 * i + 1] = s.data[3 * i + 1]; out.data[3 * i + 2] = s.data[3 * i]; }
  } else if (code == COLOR_RGB2YUV) {
    bso_rgb2yuv_u8(s.data, (long)s.total(), out.data);
  } else { fprintf(stderr, "cv_shim: cvtColor code %d not provided\n", code); abort(); }
  dst = out;
}

void bilateralFilter(const Mat& src, Mat& dst, int d, double sigmaColor, double sigmaSpace) {
  assert(src.type() == CV_8UC3);
  Mat s = packed(src);
  Mat out(src.rows, src.cols, CV_8UC3);
  bso_bilateral_c3(s.data, s.cols, s.rows, out.data, d, sigmaColor, sigmaSpace);
  dst = out;
}

void blur(const Mat& src, Mat& dst, Size ksize) {
  assert(src.type() == CV_8UC1 && ksize.width == 5 && ksize.height == 5);
  dst.create(src.rows, src.cols, src.type());             // keeps ctx.mroi's view into the persistent mask (libbackscrub.cc:371)
  bso_blur5_u8(src.data, src.cols, src.rows, (long)src.step, dst.data, (long)dst.step);
}

void split(const Mat& src, std::vector<Mat>& planes) {
  Mat s = packed(src);
  const int cn = s.channels();
  planes.clear();
  for (int c = 0; c < cn; c++) {
    Mat p(s.rows, s.cols, CV_8UC1);
    for (size_t i = 0; i < s.total(); i++) p.data[i] = s.data[cn * i + c];
    planes.push_back(p);
  }
}

void Mat::convertTo(Mat& dst, int rtype, double alpha, double beta) const {
  assert(depth() == CV_8U && (rtype & 7) == CV_32F);
  dst.create(rows, cols, CV_MAKETYPE(CV_32F, channels()));  // matching user-memory Mat (the TFLite input tensor) is kept
  Mat s = packed(*this);
  bso_convert_f32(s.data, (long)(s.total() * channels()), (float)alpha, (float)beta, (float*)dst.data);
}

}  // namespace cv

// ---- tflite:: implementations -------------------------------------------------------------------------------------
namespace tflite {

static Interpreter* g_active = nullptr;
static const float* g_forced_output = nullptr;   // tests: bypass the network and hand these logits to the reference's decode
static long g_forced_n = 0;

std::unique_ptr<FlatBufferModel> FlatBufferModel::BuildFromFile(const char* filename) {
  void* h = bso_model_load(filename);
  if (!h) return nullptr;
  std::unique_ptr<FlatBufferModel> m(new FlatBufferModel);
  m->oracle_model = h;
  return m;
}
FlatBufferModel::~FlatBufferModel() { if (oracle_model) bso_model_free(oracle_model); }

TfLiteStatus InterpreterBuilder::operator()(std::unique_ptr<Interpreter>* out) {
  std::unique_ptr<Interpreter> it(new Interpreter);
  it->oracle_model = m_.oracle_model;
  it->custom = r_.custom;
  const int nt = bso_model_num_tensors(it->oracle_model);
  it->tensors_.resize(nt); it->dims_.resize(nt);
  for (int i = 0; i < nt; i++) {
    int s4[4];
    const int rank = bso_model_tensor_shape(it->oracle_model, i, s4);
    it->dims_[i].size = rank > 0 ? 4 : 0;
    for (int k = 0; k < 4; k++) it->dims_[i].data[k] = s4[k];
    it->tensors_[i].type = kTfLiteFloat32; it->tensors_[i].dims = &it->dims_[i]; it->tensors_[i].data.raw = nullptr;
  }
  it->inputs_ = {bso_model_input(it->oracle_model)};
  it->outputs_ = {bso_model_output(it->oracle_model)};
  *out = std::move(it);
  return kTfLiteOk;
}

Interpreter::~Interpreter() { if (g_active == this) g_active = nullptr; }

TfLiteStatus Interpreter::AllocateTensors() {
  auto count = [&](int t) { size_t n = 1; for (int k = 0; k < dims_[t].size; k++) n *= (size_t)dims_[t].data[k]; return n; };
  in_buf_.assign(count(inputs_[0]), 0.f); out_buf_.assign(count(outputs_[0]), 0.f);
  tensors_[inputs_[0]].data.f = in_buf_.data();
  tensors_[outputs_[0]].data.f = out_buf_.data();
  return kTfLiteOk;
}

// CUSTOM operators → the registration the reference added with AddCustom (lib/libbackscrub.cc:207)
static std::vector<float> g_custom_out;
static TfLiteStatus resize_cb(TfLiteContext*, TfLiteTensor* t, TfLiteIntArray* new_size) {
  size_t n = 1;
  for (int k = 0; k < new_size->size; k++) n *= (size_t)new_size->data[k];
  g_custom_out.assign(n, 0.f);
  if (t->dims) TfLiteIntArrayFree(t->dims);
  t->dims = new_size; t->data.f = g_custom_out.data();
  return kTfLiteOk;
}
static void report_cb(TfLiteContext*, const char* msg, ...) { va_list ap; va_start(ap, msg); vfprintf(stderr, msg, ap); va_end(ap); fputc('\n', stderr); }

static int run_custom(const TfLiteRegistration* reg, const unsigned char* opts, int n_opts, const float* x, const int* xs4, const float* w, const int* ws4,
                      const float* b, int nb, float** y_out, int* ys4) {
  TfLiteIntArray d0{4, {xs4[0], xs4[1], xs4[2], xs4[3]}}, d1{4, {ws4[0], ws4[1], ws4[2], ws4[3]}}, d2{1, {nb}};
  TfLiteTensor t[4];
  t[0].type = kTfLiteFloat32; t[0].data.raw = (void*)x; t[0].dims = &d0;
  t[1].type = kTfLiteFloat32; t[1].data.raw = (void*)w; t[1].dims = &d1;
  t[2].type = kTfLiteFloat32; t[2].data.raw = (void*)b; t[2].dims = &d2;
  t[3].type = kTfLiteFloat32; t[3].data.raw = nullptr; t[3].dims = nullptr;
  TfLiteContext ctx{t, resize_cb, report_cb};
  TfLiteIntArray ins{3, {0, 1, 2}}, outs{1, {3}};
  TfLiteNode node{&ins, &outs, opts, n_opts};
  if (reg->prepare(&ctx, &node) != kTfLiteOk) return -1;
  if (reg->invoke(&ctx, &node) != kTfLiteOk) return -1;
  if (!t[3].dims || t[3].dims->size != 4) return -1;
  for (int k = 0; k < 4; k++) ys4[k] = t[3].dims->data[k];
  TfLiteIntArrayFree(t[3].dims);
  *y_out = g_custom_out.data();
  return 0;
}

static int custom_hook(const char* name, const unsigned char* opts, int n_opts, const float* x, const int* xs4, const float* w, const int* ws4,
                       const float* b, int nb, float** y_out, int* ys4) {
  if (!g_active) return -1;
  auto it = g_active->custom.find(name);
  if (it == g_active->custom.end()) return -1;
  return run_custom(it->second, opts, n_opts, x, xs4, w, ws4, b, nb, y_out, ys4);
}

TfLiteStatus Interpreter::Invoke() {
  if (g_forced_output) {
    if ((size_t)g_forced_n != out_buf_.size()) return kTfLiteError;
    memcpy(out_buf_.data(), g_forced_output, out_buf_.size() * sizeof(float));
    return kTfLiteOk;
  }
  g_active = this;
  bso_set_custom_op_hook(custom_hook);
  const int rc = bso_model_invoke(oracle_model, in_buf_.data(), out_buf_.data());
  bso_set_custom_op_hook(nullptr);
  g_active = nullptr;
  return rc == 0 ? kTfLiteOk : kTfLiteError;
}

}  // namespace tflite

// ---- app/deepseg.cc:87-134, extracted verbatim at build time into oracle/_ref/ (never committed) -------------------
#include "deepseg_fragment.inc"

// ---- C face for the ctypes tests ----------------------------------------------------------------------------------
namespace mediapipe { namespace tflite_operations { TfLiteRegistration* RegisterConvolution2DTransposeBias(); } }

extern "C" {

const char* ref_tensorflow_version(void) { return bs_tensorflow_version(); }

void* ref_maskgen_new(const char* model, long threads, long width, long height, void (*ondebug)(void*, const char*), void (*onprep)(void*),
                      void (*oninfer)(void*), void (*onmask)(void*), void* user) {
  return bs_maskgen_new(std::string(model), (size_t)threads, (size_t)width, (size_t)height, ondebug, onprep, oninfer, onmask, user);
}
void ref_maskgen_delete(void* ctx) { bs_maskgen_delete(ctx); }

// frame: packed BGR [h][w][3]; mask_out: [h][w].  Returns 1/0 like the bool of the reference; *aliases (optional) receives 1 when two
// consecutive calls hand back the same lib-owned buffer (mask = ctx.mask, :374).
int ref_maskgen_process(void* ctx, const uint8_t* bgr, int w, int h, uint8_t* mask_out, const uint8_t** mask_ptr) {
  cv::Mat frame(h, w, CV_8UC3, (void*)bgr), mask;
  if (!bs_maskgen_process(ctx, frame, mask)) return 0;
  if (mask.rows != h || mask.cols != w || mask.type() != CV_8UC1) return 0;
  for (int y = 0; y < h; y++) memcpy(mask_out + (size_t)y * w, mask.data + y * mask.step, w);
  if (mask_ptr) *mask_ptr = mask.data;
  return 1;
}

// decode tap: the model-resolution IIR state the reference fed to cv::resize on the last process call (= ofinal(in_roidim))
int ref_last_ofinal(uint8_t* out, long cap, int* w, int* h) {
  if (w) *w = cv::g_last_c1_w;
  if (h) *h = cv::g_last_c1_h;
  const long n = (long)cv::g_last_c1_resize_src.size();
  if (out && cap >= n) memcpy(out, cv::g_last_c1_resize_src.data(), n);
  return (int)n;
}
// hand these logits to the reference's decode instead of running the network (NULL → normal operation)
void ref_force_output(const float* logits, long n) { tflite::g_forced_output = logits; tflite::g_forced_n = n; }

void ref_alpha_blend(const uint8_t* bg, const uint8_t* frame, const uint8_t* mask, uint8_t* out, int w, int h) {
  cv::Mat a(h, w, CV_8UC3, (void*)bg), b(h, w, CV_8UC3, (void*)frame), m(h, w, CV_8UC1, (void*)mask);
  cv::Mat o = alpha_blend(a, b, m);
  memcpy(out, o.data, (size_t)w * h * 3);
}
void ref_convert_rgb_to_yuyv(const uint8_t* in, int w, int h, uint8_t* out) {
  cv::Mat i(h, w, CV_8UC3, (void*)in);
  cv::Mat o = convert_rgb_to_yuyv(i);
  memcpy(out, o.data, (size_t)w * h * 2);
}

// Convolution2DTransposeBias through the reference's registration (Prepare + Eval): custom options = {padding, stride_w, stride_h}
// as stored in the .tflite files.  y must hold ys4[0..3] product floats; call with y == NULL to query the shape.
int ref_tconv_bias(const float* x, const int* xs4, const float* w, const int* ws4, const float* b, int padding, int stride_w, int stride_h,
                   float* y, int* ys4) {
  const int32_t opts[3] = {padding, stride_w, stride_h};
  float* yp = nullptr;
  if (tflite::run_custom(mediapipe::tflite_operations::RegisterConvolution2DTransposeBias(), (const unsigned char*)opts, 12, x, xs4, w, ws4, b, ws4[0],
                         &yp, ys4) != 0) return -1;
  if (y) memcpy(y, yp, sizeof(float) * (size_t)ys4[0] * ys4[1] * ys4[2] * ys4[3]);
  return 0;
}

}  // extern "C"
