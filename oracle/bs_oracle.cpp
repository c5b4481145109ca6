// bs_oracle.cpp — CPU ORACLE for the backscrub per-frame hot path.
//
// ***************************************************************************************
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load it.  The product (libbsx.so) never links,
// loads or calls anything in oracle/.
//
// PARITY PINNING: every piece of this path whose source IS in the reference tree is pinned to the
// reference's OWN OBJECT CODE — oracle/_ref/libbs_ref.so compiles lib/libbackscrub.cc, lib/transpose_conv_bias.cc
// and app/deepseg.cc:87-134 unmodified (oracle/Makefile `ref-lib`) and tests/test_ref_pin.py compares them with the
// functions below bit for bit (glue + geometry + state, decode + IIR, Convolution2DTransposeBias, alpha_blend, YUYV
// packing).  PARITY UNPINNED for the rest: the reference ships no tests, golden vectors or fixtures, and the remaining
// arithmetic lives in two dependencies that are absent from the checkout (TensorFlow-Lite v2.8.0, git submodule
// `tensorflow` — empty dir, .gitmodules:1-4; and the system OpenCV, 4.2.0 per README.md:63).  Sections 2 and 3 of this
// file restate their *published* algorithms (TFLite reference kernels, OpenCV imgproc 8-bit paths); what pins those
// is listed in DESIGN.md §2 (PyTorch cross-checks, exhaustive integer identities, the photo fixture).
// ***************************************************************************************
//
// Every function cites the reference file:line (into /root/reference) it follows.
// Build: see oracle/Makefile (parity build: -O2 -ffp-contract=off, no fast-math).
//
// Determinism rules we add: `ofinal` starts at 0 (the reference leaves
// it uninitialised, lib/libbackscrub.cc:257); accumulation order is the TFLite reference
// kernel order (fy, fx, ic ascending) with bias added last; expf/exp from libm.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// =======================================================================================
// 1. Minimal TFLite flatbuffer reader (schema v3).  Reference call site:
//    lib/libbackscrub.cc:190 FlatBufferModel::BuildFromFile + :205-217 interpreter build.
// =======================================================================================
struct FB {
  const uint8_t* b = nullptr;
  size_t n = 0;
  template <class T> T rd(size_t o) const { T v; if (o + sizeof(T) > n) { return T(0); } memcpy(&v, b + o, sizeof(T)); return v; }
  size_t field(size_t tbl, int idx) const {
    size_t vt = tbl - rd<int32_t>(tbl);
    uint16_t vsz = rd<uint16_t>(vt);
    size_t slot = 4 + 2 * idx;
    if (slot >= vsz) return 0;
    uint16_t off = rd<uint16_t>(vt + slot);
    return off ? tbl + off : 0;
  }
  size_t indirect(size_t o) const { return o + rd<uint32_t>(o); }
  template <class T> T scalar(size_t tbl, int idx, T def) const { size_t o = field(tbl, idx); return o ? rd<T>(o) : def; }
  size_t table(size_t tbl, int idx) const { size_t o = field(tbl, idx); return o ? indirect(o) : 0; }
  // returns start of elements, sets len
  size_t vec(size_t tbl, int idx, uint32_t* len) const {
    size_t o = field(tbl, idx);
    if (!o) { *len = 0; return 0; }
    size_t v = indirect(o);
    *len = rd<uint32_t>(v);
    return v + 4;
  }
  std::vector<int32_t> vec_i32(size_t tbl, int idx) const {
    uint32_t len; size_t s = vec(tbl, idx, &len);
    std::vector<int32_t> r(len);
    for (uint32_t i = 0; i < len; i++) r[i] = rd<int32_t>(s + 4 * i);
    return r;
  }
  std::string str(size_t tbl, int idx) const {
    uint32_t len; size_t s = vec(tbl, idx, &len);
    return s ? std::string((const char*)b + s, len) : std::string();
  }
};

enum OpCode { ADD = 0, AVERAGE_POOL_2D = 1, CONCATENATION = 2, CONV_2D = 3, DEPTHWISE_CONV_2D = 4, DEQUANTIZE = 6,
              FULLY_CONNECTED = 9, LOGISTIC = 14, MUL = 18, RELU = 19, RELU6 = 21, RESIZE_BILINEAR = 23, CUSTOM = 32,
              HARD_SWISH = 117 };
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 3 };

struct OTensor {
  std::vector<int> shape;
  int type = 0;  // 0 f32, 1 f16, 2 i32
  std::string name;
  std::vector<float> f;     // f32 data (activations, dequantised constants)
  std::vector<int32_t> i;   // i32 constants
  bool is_const = false;
  size_t count() const { size_t c = 1; for (int d : shape) c *= (size_t)d; return c; }
};

struct OOp {
  int code = 0;
  std::string custom;
  std::vector<int> in, out;
  int padding = 0, stride_w = 1, stride_h = 1, act = 0, dil_w = 1, dil_h = 1, depth_mult = 1;
  int filter_w = 0, filter_h = 0, axis = 0, align_corners = 0, half_pixel = 0, keep_num_dims = 0;
  std::vector<uint8_t> custom_opts;
  bool folded = false;  // constant-folded at load
};

static float half_to_float(uint16_t h) {
  uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, out;
  if (e == 0) {
    if (m == 0) out = s << 31;
    else { e = 127 - 15 + 1; while (!(m & 1024)) { m <<= 1; e--; } m &= 1023; out = (s << 31) | (e << 23) | (m << 13); }
  } else if (e == 31) out = (s << 31) | 0x7f800000u | (m << 13);
  else out = (s << 31) | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &out, 4); return f;
}

struct OModel {
  std::vector<uint8_t> file;
  std::vector<OTensor> t;
  std::vector<OOp> ops;
  std::vector<int> inputs, outputs;
  std::string err, desc_json;
};

static bool load_model(const char* path, OModel& m) {
  FILE* fp = fopen(path, "rb");
  if (!fp) { m.err = "cannot open"; return false; }
  fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
  m.file.resize(sz);
  if (fread(m.file.data(), 1, sz, fp) != (size_t)sz) { fclose(fp); m.err = "short read"; return false; }
  fclose(fp);
  if (sz < 16) { m.err = "too small"; return false; }
  FB fb; fb.b = m.file.data(); fb.n = m.file.size();
  size_t root = fb.indirect(0);
  uint32_t n; size_t s;
  std::vector<std::pair<int, std::string>> codes;
  s = fb.vec(root, 1, &n);
  for (uint32_t i = 0; i < n; i++) {
    size_t oc = fb.indirect(s + 4 * i);
    int dep = fb.scalar<int8_t>(oc, 0, 0), full = fb.scalar<int32_t>(oc, 3, 0);
    codes.push_back({std::max(dep, full), fb.str(oc, 1)});
  }
  std::vector<std::pair<size_t, uint32_t>> bufs;
  s = fb.vec(root, 4, &n);
  for (uint32_t i = 0; i < n; i++) { size_t bt = fb.indirect(s + 4 * i); uint32_t len; size_t d = fb.vec(bt, 0, &len); bufs.push_back({d, len}); }
  s = fb.vec(root, 2, &n);
  if (!n) { m.err = "no subgraph"; return false; }
  size_t sg = fb.indirect(s);
  s = fb.vec(sg, 0, &n);
  for (uint32_t i = 0; i < n; i++) {
    size_t tt = fb.indirect(s + 4 * i);
    OTensor t;
    t.shape = fb.vec_i32(tt, 0);
    t.type = fb.scalar<int8_t>(tt, 1, 0);
    uint32_t bi = fb.scalar<uint32_t>(tt, 2, 0);
    t.name = fb.str(tt, 3);
    if (bi < bufs.size() && bufs[bi].second) {
      size_t d = bufs[bi].first; uint32_t len = bufs[bi].second;
      if (d + len > fb.n) { m.err = "buffer out of range"; return false; }
      t.is_const = true;
      if (t.type == 0) { t.f.resize(len / 4); memcpy(t.f.data(), fb.b + d, len); }
      else if (t.type == 1) { t.f.resize(len / 2); for (size_t k = 0; k < len / 2; k++) t.f[k] = half_to_float(fb.rd<uint16_t>(d + 2 * k)); }
      else if (t.type == 2) { t.i.resize(len / 4); memcpy(t.i.data(), fb.b + d, len); }
    }
    m.t.push_back(std::move(t));
  }
  m.inputs = fb.vec_i32(sg, 1);
  m.outputs = fb.vec_i32(sg, 2);
  s = fb.vec(sg, 3, &n);
  for (uint32_t i = 0; i < n; i++) {
    size_t ot = fb.indirect(s + 4 * i);
    OOp op;
    uint32_t ci = fb.scalar<uint32_t>(ot, 0, 0);
    if (ci >= codes.size()) { m.err = "bad opcode index"; return false; }
    op.code = codes[ci].first; op.custom = codes[ci].second;
    for (int v : fb.vec_i32(ot, 1)) op.in.push_back(v);
    for (int v : fb.vec_i32(ot, 2)) op.out.push_back(v);
    size_t o = fb.table(ot, 4);
    if (o) {
      switch (op.code) {
        case CONV_2D: op.padding = fb.scalar<int8_t>(o, 0, 0); op.stride_w = fb.scalar<int32_t>(o, 1, 1); op.stride_h = fb.scalar<int32_t>(o, 2, 1);
          op.act = fb.scalar<int8_t>(o, 3, 0); op.dil_w = fb.scalar<int32_t>(o, 4, 1); op.dil_h = fb.scalar<int32_t>(o, 5, 1); break;
        case DEPTHWISE_CONV_2D: op.padding = fb.scalar<int8_t>(o, 0, 0); op.stride_w = fb.scalar<int32_t>(o, 1, 1); op.stride_h = fb.scalar<int32_t>(o, 2, 1);
          op.depth_mult = fb.scalar<int32_t>(o, 3, 1); op.act = fb.scalar<int8_t>(o, 4, 0); op.dil_w = fb.scalar<int32_t>(o, 5, 1); op.dil_h = fb.scalar<int32_t>(o, 6, 1); break;
        case AVERAGE_POOL_2D: op.padding = fb.scalar<int8_t>(o, 0, 0); op.stride_w = fb.scalar<int32_t>(o, 1, 1); op.stride_h = fb.scalar<int32_t>(o, 2, 1);
          op.filter_w = fb.scalar<int32_t>(o, 3, 0); op.filter_h = fb.scalar<int32_t>(o, 4, 0); op.act = fb.scalar<int8_t>(o, 5, 0); break;
        case FULLY_CONNECTED: op.act = fb.scalar<int8_t>(o, 0, 0); op.keep_num_dims = fb.scalar<uint8_t>(o, 2, 0); break;
        case CONCATENATION: op.axis = fb.scalar<int32_t>(o, 0, 0); op.act = fb.scalar<int8_t>(o, 1, 0); break;
        case ADD: case MUL: op.act = fb.scalar<int8_t>(o, 0, 0); break;
        case RESIZE_BILINEAR: op.align_corners = fb.scalar<uint8_t>(o, 2, 0); op.half_pixel = fb.scalar<uint8_t>(o, 3, 0); break;
        default: break;
      }
    }
    uint32_t cl; size_t cs = fb.vec(ot, 5, &cl);
    if (cs) op.custom_opts.assign(fb.b + cs, fb.b + cs + cl);
    m.ops.push_back(std::move(op));
  }
  return true;
}

// =======================================================================================
// 2. TFLite float reference kernels (tensorflow/lite/kernels/internal/reference/*, v2.8.0,
//    restated from the published algorithm; call site lib/libbackscrub.cc:307 Invoke()).
// =======================================================================================
static inline float apply_act(float v, int act) {
  if (act == ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == ACT_RELU6) return std::min(std::max(v, 0.f), 6.f);
  return v;
}

// SAME/VALID output size + leading pad (tflite ComputeOutSize / ComputePaddingWithOffset)
static void conv_geom(int in, int k, int stride, int dil, int padding, int* out, int* pad) {
  int eff = (k - 1) * dil + 1;
  if (padding == 0) *out = (in + stride - 1) / stride;       // SAME
  else *out = (in + stride - eff) / stride;                  // VALID
  int total = (*out - 1) * stride + eff - in;
  if (total < 0) total = 0;
  *pad = total / 2;
}

// CONV_2D: reference conv.h — per output: total over (fy,fx,ic) ascending, + bias, activation.
// Weights are pre-transposed to [kh][kw][ci][co] so that the inner loop runs over co; each
// co's accumulator still receives its products in exactly the reference order.
static void op_conv2d(const OTensor& x, const OTensor& w, const OTensor* b, OTensor& y, const OOp& op) {
  int H = x.shape[1], W = x.shape[2], Ci = x.shape[3];
  int Co = w.shape[0], kh = w.shape[1], kw = w.shape[2];
  int OH, OW, ph, pw;
  conv_geom(H, kh, op.stride_h, op.dil_h, op.padding, &OH, &ph);
  conv_geom(W, kw, op.stride_w, op.dil_w, op.padding, &OW, &pw);
  y.shape = {1, OH, OW, Co};
  y.f.assign((size_t)OH * OW * Co, 0.f);
  std::vector<float> wt((size_t)kh * kw * Ci * Co);
  for (int o = 0; o < Co; o++) for (int fy = 0; fy < kh; fy++) for (int fx = 0; fx < kw; fx++) for (int c = 0; c < Ci; c++)
    wt[(((size_t)fy * kw + fx) * Ci + c) * Co + o] = w.f[(((size_t)o * kh + fy) * kw + fx) * Ci + c];
  std::vector<float> acc(Co);
  for (int oy = 0; oy < OH; oy++) for (int ox = 0; ox < OW; ox++) {
    std::fill(acc.begin(), acc.end(), 0.f);
    for (int fy = 0; fy < kh; fy++) {
      int iy = oy * op.stride_h - ph + fy * op.dil_h;
      if (iy < 0 || iy >= H) continue;
      for (int fx = 0; fx < kw; fx++) {
        int ix = ox * op.stride_w - pw + fx * op.dil_w;
        if (ix < 0 || ix >= W) continue;
        const float* xp = &x.f[((size_t)iy * W + ix) * Ci];
        const float* wp = &wt[((size_t)fy * kw + fx) * Ci * Co];
        for (int c = 0; c < Ci; c++) {
          float xv = xp[c];
          const float* wr = wp + (size_t)c * Co;
          for (int o = 0; o < Co; o++) acc[o] += xv * wr[o];
        }
      }
    }
    float* yp = &y.f[((size_t)oy * OW + ox) * Co];
    for (int o = 0; o < Co; o++) yp[o] = apply_act(acc[o] + (b ? b->f[o] : 0.f), op.act);
  }
}

// DEPTHWISE_CONV_2D: reference depthwiseconv_float.h (depth_multiplier==1 in all shipped models).
static void op_dwconv(const OTensor& x, const OTensor& w, const OTensor* b, OTensor& y, const OOp& op) {
  int H = x.shape[1], W = x.shape[2], C = x.shape[3];
  int kh = w.shape[1], kw = w.shape[2], Co = w.shape[3];
  int dm = op.depth_mult;
  int OH, OW, ph, pw;
  conv_geom(H, kh, op.stride_h, op.dil_h, op.padding, &OH, &ph);
  conv_geom(W, kw, op.stride_w, op.dil_w, op.padding, &OW, &pw);
  y.shape = {1, OH, OW, Co};
  y.f.assign((size_t)OH * OW * Co, 0.f);
  std::vector<float> acc(Co);
  for (int oy = 0; oy < OH; oy++) for (int ox = 0; ox < OW; ox++) {
    std::fill(acc.begin(), acc.end(), 0.f);
    for (int fy = 0; fy < kh; fy++) {
      int iy = oy * op.stride_h - ph + fy * op.dil_h;
      if (iy < 0 || iy >= H) continue;
      for (int fx = 0; fx < kw; fx++) {
        int ix = ox * op.stride_w - pw + fx * op.dil_w;
        if (ix < 0 || ix >= W) continue;
        const float* xp = &x.f[((size_t)iy * W + ix) * C];
        const float* wp = &w.f[((size_t)fy * kw + fx) * Co];
        if (dm == 1) { for (int c = 0; c < Co; c++) acc[c] += xp[c] * wp[c]; }
        else { for (int c = 0; c < C; c++) for (int mth = 0; mth < dm; mth++) acc[c * dm + mth] += xp[c] * wp[c * dm + mth]; }
      }
    }
    float* yp = &y.f[((size_t)oy * OW + ox) * Co];
    for (int c = 0; c < Co; c++) yp[c] = apply_act(acc[c] + (b ? b->f[c] : 0.f), op.act);
  }
}

// FULLY_CONNECTED: reference fully_connected.h — weights [O,I], bias added after the sum.
static void op_fc(const OTensor& x, const OTensor& w, const OTensor* b, OTensor& y, const OOp& op) {
  int O = w.shape[0], I = w.shape[1];
  size_t batches = x.count() / I;
  y.shape = x.shape; y.shape.back() = O;
  if (!op.keep_num_dims) y.shape = {(int)batches, O};
  y.f.assign(batches * O, 0.f);
  for (size_t bb = 0; bb < batches; bb++) for (int o = 0; o < O; o++) {
    float total = 0.f;
    for (int d = 0; d < I; d++) total += x.f[bb * I + d] * w.f[(size_t)o * I + d];
    y.f[bb * O + o] = apply_act(total + (b ? b->f[o] : 0.f), op.act);
  }
}

// AVERAGE_POOL_2D: reference pooling.h — window clipped to the input, total / count.
static void op_avgpool(const OTensor& x, OTensor& y, const OOp& op) {
  int H = x.shape[1], W = x.shape[2], C = x.shape[3];
  int OH, OW, ph, pw;
  conv_geom(H, op.filter_h, op.stride_h, 1, op.padding, &OH, &ph);
  conv_geom(W, op.filter_w, op.stride_w, 1, op.padding, &OW, &pw);
  y.shape = {1, OH, OW, C};
  y.f.assign((size_t)OH * OW * C, 0.f);
  for (int oy = 0; oy < OH; oy++) for (int ox = 0; ox < OW; ox++) {
    int y0 = oy * op.stride_h - ph, x0 = ox * op.stride_w - pw;
    int fys = std::max(0, -y0), fye = std::min(op.filter_h, H - y0);
    int fxs = std::max(0, -x0), fxe = std::min(op.filter_w, W - x0);
    for (int c = 0; c < C; c++) {
      float total = 0.f; float cnt = 0.f;
      for (int fy = fys; fy < fye; fy++) for (int fx = fxs; fx < fxe; fx++) { total += x.f[((size_t)(y0 + fy) * W + (x0 + fx)) * C + c]; cnt += 1.f; }
      y.f[((size_t)oy * OW + ox) * C + c] = apply_act(total / cnt, op.act);
    }
  }
}

// ADD / MUL with numpy broadcasting over 4-D shapes (reference binary ops + BroadcastXSlow).
static void op_binary(const OTensor& a, const OTensor& b, OTensor& y, const OOp& op, bool mul) {
  std::vector<int> sa = a.shape, sb = b.shape;
  while (sa.size() < 4) sa.insert(sa.begin(), 1);
  while (sb.size() < 4) sb.insert(sb.begin(), 1);
  int so[4]; for (int i = 0; i < 4; i++) so[i] = std::max(sa[i], sb[i]);
  y.shape = {so[0], so[1], so[2], so[3]};
  y.f.resize((size_t)so[0] * so[1] * so[2] * so[3]);
  size_t idx = 0;
  for (int n = 0; n < so[0]; n++) for (int h = 0; h < so[1]; h++) for (int w = 0; w < so[2]; w++) for (int c = 0; c < so[3]; c++) {
    size_t ia = (((size_t)(sa[0] == 1 ? 0 : n) * sa[1] + (sa[1] == 1 ? 0 : h)) * sa[2] + (sa[2] == 1 ? 0 : w)) * sa[3] + (sa[3] == 1 ? 0 : c);
    size_t ib = (((size_t)(sb[0] == 1 ? 0 : n) * sb[1] + (sb[1] == 1 ? 0 : h)) * sb[2] + (sb[2] == 1 ? 0 : w)) * sb[3] + (sb[3] == 1 ? 0 : c);
    float v = mul ? a.f[ia] * b.f[ib] : a.f[ia] + b.f[ib];
    y.f[idx++] = apply_act(v, op.act);
  }
}

static void op_unary(const OTensor& x, OTensor& y, int code) {
  y.shape = x.shape; y.f.resize(x.f.size());
  for (size_t i = 0; i < x.f.size(); i++) {
    float v = x.f[i];
    switch (code) {
      case RELU: v = v < 0.f ? 0.f : v; break;
      case RELU6: v = std::min(std::max(v, 0.f), 6.f); break;
      // reference hard_swish.h (float): in * min(6, max(0, in + 3)) / 6
      case HARD_SWISH: v = v * std::min(6.f, std::max(0.f, v + 3.f)) / 6.f; break;
      // reference logistic.h (float): 1 / (1 + exp(-x))
      case LOGISTIC: v = 1.f / (1.f + std::exp(-v)); break;
    }
    y.f[i] = v;
  }
}

static void op_concat(const std::vector<const OTensor*>& xs, OTensor& y, int axis) {
  int nd = (int)xs[0]->shape.size();
  if (axis < 0) axis += nd;
  y.shape = xs[0]->shape; y.shape[axis] = 0;
  for (auto* x : xs) y.shape[axis] += x->shape[axis];
  size_t outer = 1, inner = 1;
  for (int i = 0; i < axis; i++) outer *= xs[0]->shape[i];
  for (int i = axis + 1; i < nd; i++) inner *= xs[0]->shape[i];
  y.f.resize(y.count());
  size_t ystride = (size_t)y.shape[axis] * inner;
  size_t off = 0;
  for (auto* x : xs) {
    size_t xs_ = (size_t)x->shape[axis] * inner;
    for (size_t o = 0; o < outer; o++) memcpy(&y.f[o * ystride + off], &x->f[o * xs_], xs_ * sizeof(float));
    off += xs_;
  }
}

// RESIZE_BILINEAR: reference resize_bilinear.h (v2.8.0) ComputeInterpolationValues.
static void interp_values(int v, float scale, bool half_pixel, int in_size, float* scaled, int* lo, int* hi) {
  if (half_pixel) *scaled = ((float)v + 0.5f) * scale - 0.5f; else *scaled = (float)v * scale;
  float fl = std::floor(*scaled);
  *lo = std::max((int)fl, 0);
  *hi = std::min((int)std::ceil(*scaled), in_size - 1);
}
static void op_resize_bilinear(const OTensor& x, int OH, int OW, OTensor& y, const OOp& op) {
  int H = x.shape[1], W = x.shape[2], C = x.shape[3];
  y.shape = {1, OH, OW, C}; y.f.resize((size_t)OH * OW * C);
  float hs = (float)H / (float)OH, ws = (float)W / (float)OW;
  if (op.align_corners && OH > 1) hs = (float)(H - 1) / (float)(OH - 1);
  if (op.align_corners && OW > 1) ws = (float)(W - 1) / (float)(OW - 1);
  for (int oy = 0; oy < OH; oy++) {
    float iy; int y0, y1; interp_values(oy, hs, op.half_pixel, H, &iy, &y0, &y1);
    for (int ox = 0; ox < OW; ox++) {
      float ix; int x0, x1; interp_values(ox, ws, op.half_pixel, W, &ix, &x0, &x1);
      float dy = iy - (float)y0, dx = ix - (float)x0;
      for (int c = 0; c < C; c++) {
        float v = x.f[((size_t)y0 * W + x0) * C + c] * (1.f - dy) * (1.f - dx) +
                  x.f[((size_t)y1 * W + x0) * C + c] * dy * (1.f - dx) +
                  x.f[((size_t)y0 * W + x1) * C + c] * (1.f - dy) * dx +
                  x.f[((size_t)y1 * W + x1) * C + c] * dy * dx;
        y.f[((size_t)oy * OW + ox) * C + c] = v;
      }
    }
  }
}

// Convolution2DTransposeBias — follows lib/transpose_conv_bias.cc:37-114 (scatter loops:
// output initialised with bias, then for each input (y,x,ic) accumulate into the
// influenced outputs) with padding from :210-228 (SAME → max(0,k-(in-1)%s-1), halved).
static void op_tconv_bias(const OTensor& x, const OTensor& w, const OTensor& b, OTensor& y, const OOp& op) {
  int H = x.shape[1], W = x.shape[2], Ci = x.shape[3];
  int Co = w.shape[0], kh = w.shape[1], kw = w.shape[2];
  int padding = 1, sw = 2, sh = 2;
  if (op.custom_opts.size() >= 12) { int32_t v[3]; memcpy(v, op.custom_opts.data(), 12); padding = v[0]; sw = v[1]; sh = v[2]; }
  int pad_h = 0, pad_w = 0;
  if (padding == 1 /*kTfLitePaddingSame*/) { pad_h = std::max(0, kh - (H - 1) % sh - 1); pad_w = std::max(0, kw - (W - 1) % sw - 1); }
  int OH = sh * (H - 1) + kh - pad_h, OW = sw * (W - 1) + kw - pad_w;  // transpose_conv_bias.cc:177-180
  int ph = pad_h / 2, pw = pad_w / 2;                                  // :225-226
  y.shape = {1, OH, OW, Co}; y.f.resize((size_t)OH * OW * Co);
  for (int oy = 0; oy < OH; oy++) for (int ox = 0; ox < OW; ox++) for (int oc = 0; oc < Co; oc++) y.f[((size_t)oy * OW + ox) * Co + oc] = b.f[oc];
  for (int iy = 0; iy < H; iy++) for (int ix = 0; ix < W; ix++) for (int ic = 0; ic < Ci; ic++) {
    int oxo = ix * sw - pw, oyo = iy * sh - ph;
    float xv = x.f[((size_t)iy * W + ix) * Ci + ic];
    for (int fy = 0; fy < kh; fy++) for (int fx = 0; fx < kw; fx++) for (int oc = 0; oc < Co; oc++) {
      int ox = oxo + fx, oy = oyo + fy;
      if (ox >= 0 && ox < OW && oy >= 0 && oy < OH)
        y.f[((size_t)oy * OW + ox) * Co + oc] += xv * w.f[(((size_t)oc * kh + fy) * kw + fx) * Ci + ic];
    }
  }
}

// Optional hook for CUSTOM ops: when set (only by oracle/ref_shim, which routes Convolution2DTransposeBias through the
// reference's OWN compiled Prepare/Eval, lib/transpose_conv_bias.cc:118-256), it replaces the restatement above.
// y_out points at hook-owned storage that stays valid until the next call.
typedef int (*bso_custom_fn)(const char* name, const unsigned char* opts, int n_opts, const float* x, const int* xs4, const float* w,
                             const int* ws4, const float* b, int nb, float** y_out, int* ys4);
static bso_custom_fn g_custom_hook = nullptr;

static bool run_op(OModel& m, OOp& op) {
  auto T = [&](int i) -> OTensor& { return m.t[i]; };
  auto opt = [&](size_t k) -> const OTensor* { return (k < op.in.size() && op.in[k] >= 0) ? &m.t[op.in[k]] : nullptr; };
  OTensor& y = T(op.out[0]);
  switch (op.code) {
    case DEQUANTIZE: y.f = T(op.in[0]).f; y.shape = T(op.in[0]).shape; return true;  // f16→f32 is exact
    case CONV_2D: op_conv2d(T(op.in[0]), T(op.in[1]), opt(2), y, op); return true;
    case DEPTHWISE_CONV_2D: op_dwconv(T(op.in[0]), T(op.in[1]), opt(2), y, op); return true;
    case FULLY_CONNECTED: op_fc(T(op.in[0]), T(op.in[1]), opt(2), y, op); return true;
    case AVERAGE_POOL_2D: op_avgpool(T(op.in[0]), y, op); return true;
    case ADD: op_binary(T(op.in[0]), T(op.in[1]), y, op, false); return true;
    case MUL: op_binary(T(op.in[0]), T(op.in[1]), y, op, true); return true;
    case RELU: case RELU6: case HARD_SWISH: case LOGISTIC: op_unary(T(op.in[0]), y, op.code); return true;
    case CONCATENATION: { std::vector<const OTensor*> xs; for (int i : op.in) xs.push_back(&T(i)); op_concat(xs, y, op.axis); return true; }
    case RESIZE_BILINEAR: { const OTensor& sz = T(op.in[1]); if (sz.i.size() < 2) return false; op_resize_bilinear(T(op.in[0]), sz.i[0], sz.i[1], y, op); return true; }
    case CUSTOM:
      if (g_custom_hook && op.in.size() >= 3) {
        const OTensor& x = T(op.in[0]); const OTensor& w = T(op.in[1]); const OTensor& b = T(op.in[2]);
        if (x.shape.size() != 4 || w.shape.size() != 4) return false;
        float* yp = nullptr; int ys[4] = {0, 0, 0, 0};
        if (g_custom_hook(op.custom.c_str(), op.custom_opts.data(), (int)op.custom_opts.size(), x.f.data(), x.shape.data(), w.f.data(), w.shape.data(),
                          b.f.data(), (int)b.f.size(), &yp, ys) != 0 || !yp) return false;
        y.shape.assign(ys, ys + 4); y.f.assign(yp, yp + y.count());
        return true;
      }
      if (op.custom == "Convolution2DTransposeBias") { op_tconv_bias(T(op.in[0]), T(op.in[1]), T(op.in[2]), y, op); return true; }
      return false;
    default: return false;
  }
}

// fold ops whose inputs are all constants (the 110 DEQUANTIZE weight ops in the f16 models)
static void fold_constants(OModel& m) {
  for (auto& op : m.ops) {
    bool all_const = !op.in.empty();
    for (int i : op.in) if (i >= 0 && !m.t[i].is_const) all_const = false;
    if (!all_const) continue;
    if (run_op(m, op)) { m.t[op.out[0]].is_const = true; op.folded = true; }
  }
}

static bool invoke(OModel& m, const float* in, float* out) {
  OTensor& ti = m.t[m.inputs[0]];
  ti.f.assign(in, in + ti.count());
  for (auto& op : m.ops) { if (op.folded) continue; if (!run_op(m, op)) return false; }
  const OTensor& to = m.t[m.outputs[0]];
  if (out) memcpy(out, to.f.data(), to.f.size() * sizeof(float));
  return true;
}

// =======================================================================================
// 3. OpenCV 8-bit image ops (imgproc, 4.2.0 C fall-back paths restated; call sites cited)
// =======================================================================================
static inline int cv_round(double v) { return (int)lrint(v); }          // round-half-even (default FE mode)
static inline int cv_roundf(float v) { return (int)lrintf(v); }
static inline int cv_floorf(float v) { int i = (int)v; return i - (i > v); }
static inline short sat_short(int v) { return (short)std::min(std::max(v, -32768), 32767); }
static inline int reflect101(int p, int len) {  // BORDER_REFLECT_101 (cv::borderInterpolate)
  if (len == 1) return 0;
  while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
  return p;
}

// cv::resize(..., INTER_LINEAR) for CV_8UC{1,3} (resize.cpp: fixed-point coefficients
// short(round(w*2048)), HResizeLinear int32, VResizeLinear FixedPtCast<int,uchar,22>
// written as (((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2).  Same-size → copy.  When both
// integer scales are exactly 2 INTER_LINEAR is silently replaced by INTER_AREA (2x2 mean,
// (s+2)>>2).  Call sites: lib/libbackscrub.cc:289 (frame ROI ↓), :368 (mask ↑),
// app/background.cc:186,190 (background → frame size).
static void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst, int dw, int dh, size_t dstride) {
  if (sw == dw && sh == dh) { for (int y = 0; y < dh; y++) memcpy(dst + y * dstride, src + y * sstride, (size_t)dw * cn); return; }
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int iscale_x = (int)(int64_t)std::min(std::max(lrint(scale_x), -2147483647L), 2147483647L);
  int iscale_y = (int)(int64_t)std::min(std::max(lrint(scale_y), -2147483647L), 2147483647L);
  bool is_area_fast = std::abs(scale_x - iscale_x) < 2.220446049250313e-16 && std::abs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; y++) for (int x = 0; x < dw; x++) for (int c = 0; c < cn; c++) {
      const uint8_t* s0 = src + (size_t)(2 * y) * sstride + (size_t)(2 * x) * cn + c;
      const uint8_t* s1 = s0 + sstride;
      dst[y * dstride + (size_t)x * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
    }
    return;
  }
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> xa(2 * dw), ya(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    xa[2 * dx] = sat_short(cv_roundf((1.f - fx) * 2048.f));
    xa[2 * dx + 1] = sat_short(cv_roundf(fx * 2048.f));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floorf(fy);
    fy -= sy;
    yofs[dy] = sy;
    ya[2 * dy] = sat_short(cv_roundf((1.f - fy) * 2048.f));
    ya[2 * dy + 1] = sat_short(cv_roundf(fy * 2048.f));
  }
  std::vector<int> r0((size_t)dw * cn), r1((size_t)dw * cn);
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = std::min(std::max(yofs[dy], 0), sh - 1), sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)sy0 * sstride; const uint8_t* S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
      int a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
      for (int c = 0; c < cn; c++) {
        r0[(size_t)dx * cn + c] = S0[(size_t)sx * cn + c] * a0 + S0[(size_t)sx1 * cn + c] * a1;
        r1[(size_t)dx * cn + c] = S1[(size_t)sx * cn + c] * a0 + S1[(size_t)sx1 * cn + c] * a1;
      }
    }
    int b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int i = 0; i < dw * cn; i++) {
      int v = (((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2;
      D[i] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

// cv::bilateralFilter(src,dst,d=5,sigmaColor,sigmaSpace) for CV_8UC3 (bilateral_filter:
// radius 2, the 13 taps with sqrt(i²+j²)<=2 in row-major order, weight =
// space_w[k]*color_w[|Δ0|+|Δ1|+|Δ2|], REFLECT_101 border, f32 sums in k order,
// out = cvRound(sum * (1/wsum))).  Call site: lib/libbackscrub.cc:297.
static void bilateral_c3(const uint8_t* src, int w, int h, uint8_t* dst, int d, double sigma_color, double sigma_space) {
  const int cn = 3;
  if (sigma_color <= 0) sigma_color = 1;
  if (sigma_space <= 0) sigma_space = 1;
  double gc = -0.5 / (sigma_color * sigma_color), gs = -0.5 / (sigma_space * sigma_space);
  int radius = d <= 0 ? cv_round(sigma_space * 1.5) : d / 2;
  radius = std::max(radius, 1);
  std::vector<float> color_w(cn * 256);
  for (int i = 0; i < cn * 256; i++) color_w[i] = (float)std::exp(i * i * gc);
  std::vector<float> space_w; std::vector<int> oy, ox;
  for (int i = -radius; i <= radius; i++) for (int j = -radius; j <= radius; j++) {
    double r = std::sqrt((double)i * i + (double)j * j);
    if (r > radius) continue;
    space_w.push_back((float)std::exp(r * r * gs)); oy.push_back(i); ox.push_back(j);
  }
  int maxk = (int)space_w.size();
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t* p0 = src + ((size_t)y * w + x) * cn;
    int b0 = p0[0], g0 = p0[1], r0 = p0[2];
    float sb = 0, sg = 0, sr = 0, ws = 0;
    for (int k = 0; k < maxk; k++) {
      int yy = reflect101(y + oy[k], h), xx = reflect101(x + ox[k], w);
      const uint8_t* p = src + ((size_t)yy * w + xx) * cn;
      int b = p[0], g = p[1], r = p[2];
      float wgt = space_w[k] * color_w[std::abs(b - b0) + std::abs(g - g0) + std::abs(r - r0)];
      sb += b * wgt; sg += g * wgt; sr += r * wgt; ws += wgt;
    }
    ws = 1.f / ws;
    uint8_t* q = dst + ((size_t)y * w + x) * cn;
    q[0] = (uint8_t)cv_roundf(sb * ws); q[1] = (uint8_t)cv_roundf(sg * ws); q[2] = (uint8_t)cv_roundf(sr * ws);
  }
}

// cv::blur(src,dst,Size(5,5)) on CV_8UC1: normalised box, REFLECT_101, int sums scaled by 1/25
// and rounded → (s+12)/25 (no ties possible).  Call site: lib/libbackscrub.cc:371.
static void blur5_u8(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
  // separable form of the 25-tap sum (row sums, then column sums — OpenCV's RowSum/ColumnSum structure); same integers
  std::vector<int> rows((size_t)h * w);
  std::vector<int> xi(w + 4);
  for (int x = -2; x < w + 2; x++) xi[x + 2] = reflect101(x, w);
  for (int y = 0; y < h; y++) {
    const uint8_t* r = src + (size_t)y * sstride;
    int* o = &rows[(size_t)y * w];
    for (int x = 0; x < w; x++) o[x] = r[xi[x]] + r[xi[x + 1]] + r[xi[x + 2]] + r[xi[x + 3]] + r[xi[x + 4]];
  }
  for (int y = 0; y < h; y++) {
    const int* r0 = &rows[(size_t)reflect101(y - 2, h) * w]; const int* r1 = &rows[(size_t)reflect101(y - 1, h) * w];
    const int* r2 = &rows[(size_t)y * w]; const int* r3 = &rows[(size_t)reflect101(y + 1, h) * w]; const int* r4 = &rows[(size_t)reflect101(y + 2, h) * w];
    uint8_t* d = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) d[x] = (uint8_t)cv_round((r0[x] + r1[x] + r2[x] + r3[x] + r4[x]) * (1. / 25));
  }
}

// cv::GaussianBlur(src, dst, Size(n, n), 0) for CV_8UC3, BORDER_REFLECT_101 — app/deepseg.cc:657-658 (`-p bgblur:<n>`, n odd,
// default 25, :420-429).  OpenCV's 8-bit path is a FIXED-POINT separable filter (imgproc smooth.dispatch.cpp / smooth.simd.hpp,
// "bit-exact" since 3.4): coefficients ufixedpoint16 (8 fractional bits) = cvRound(k[i] * 256) of the normalised kernel
// k[i] = exp(-x^2 / (2 sigma^2)) / sum, sigma = 0.3 * ((n-1) * 0.5 - 1) + 0.8 for sigma <= 0, with the exact tables
// {1}, {1,2,1}/4, {1,4,6,4,1}/16, {1,3.5,7,9,7,3.5,1}/32 for n = 1, 3, 5, 7; horizontal pass Σ c·src in saturating u16
// (8 fractional bits), vertical pass Σ c·h in saturating u32 (16 fractional bits), result = sat_u8((v + 2^15) >> 16).
// Recorded ambiguities (unpinned: OpenCV is absent from the checkout): (1) the kernel is evaluated in SOFTWARE double
// (softdouble) there, in libm double here — a coefficient could differ only if k*256 sits within 1e-13 of a rounding tie;
// (2) this is the coefficient rule of OpenCV 3.4 - 4.4 (README.md:63 names 4.2.0); 4.5+ diffuses the rounding error so that
// the coefficients sum to exactly 256.
static std::vector<uint16_t> gaussian_coeffs_u16(int n) {
  std::vector<uint16_t> c(n);
  if (n == 1) { c[0] = 256; return c; }
  if (n == 3) { c = {64, 128, 64}; return c; }
  if (n == 5) { c = {16, 64, 96, 64, 16}; return c; }
  if (n == 7) { c = {8, 28, 56, 72, 56, 28, 8}; return c; }
  const double sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2x = (-0.5 * 0.25) / (sigma * sigma);
  std::vector<double> v(n);
  double sum = 0;
  for (int i = 0, x = 1 - n; i < n; i++, x += 2) { v[i] = std::exp((double)(x * x) * scale2x); sum += v[i]; }
  const double inv = 1.0 / sum;
  for (int i = 0; i < n; i++) c[i] = (uint16_t)cv_round(v[i] * inv * 256.0);
  return c;
}
static void gaussian_blur_c3(const uint8_t* src, int w, int h, int n, uint8_t* dst) {
  const std::vector<uint16_t> c = gaussian_coeffs_u16(n);
  const int r = n / 2;
  std::vector<uint16_t> hbuf((size_t)w * h * 3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      for (int ch = 0; ch < 3; ch++) {
        uint32_t acc = 0;
        for (int k = 0; k < n; k++) {
          acc += (uint32_t)c[k] * src[((size_t)y * w + reflect101(x + k - r, w)) * 3 + ch];
          if (acc > 0xFFFFu) acc = 0xFFFFu;                      // ufixedpoint16 saturating add
        }
        hbuf[((size_t)y * w + x) * 3 + ch] = (uint16_t)acc;
      }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      for (int ch = 0; ch < 3; ch++) {
        uint64_t acc = 0;
        for (int k = 0; k < n; k++) {
          acc += (uint64_t)c[k] * hbuf[((size_t)reflect101(y + k - r, h) * w + x) * 3 + ch];
          if (acc > 0xFFFFFFFFull) acc = 0xFFFFFFFFull;          // ufixedpoint32 saturating add
        }
        const uint64_t v = (acc + (1u << 15)) >> 16;
        dst[((size_t)y * w + x) * 3 + ch] = (uint8_t)std::min<uint64_t>(v, 255);
      }
}

// alpha_blend — follows app/deepseg.cc:108-134 (int math, truncating /255; a=bg, b=frame).
static void alpha_blend(const uint8_t* a, const uint8_t* b, const uint8_t* m, uint8_t* o, size_t npix) {
  for (size_t p = 0; p < npix; p++) {
    int aw = m[p], bw = 255 - aw;
    for (int c = 0; c < 3; c++) o[3 * p + c] = (uint8_t)(((int)a[3 * p + c] * aw + (int)b[3 * p + c] * bw) / 255);
  }
}

// convert_rgb_to_yuyv — follows app/deepseg.cc:87-106: cv::cvtColor(COLOR_RGB2YUV) applied to
// the (BGR-ordered) data, i.e. channel0 is treated as R.  OpenCV 8u RGB2YUV (color_yuv):
// 14-bit fixed point, Y = (R*4899 + G*9617 + B*1868 + 8192)>>14; U = ((B-Y)*8061 + 128<<14 + 8192)>>14;
// V = ((R-Y)*14369 + ...)>>14, saturated.  Then 4:2:2 packing in byte order Y0,V,Y1,U
// with (c0+c1)/2 chroma (deepseg.cc:98-103).
// cv::cvtColor(COLOR_RGB2YUV) on 8-bit 3-channel pixels: channel 0 is taken as R (interleaved Y,U,V out)
static void rgb2yuv_u8(const uint8_t* in, size_t total, uint8_t* yuv) {
  const int shift = 14, half = 1 << (shift - 1), delta = 128 << shift;
  auto sat = [](int v) { return (uint8_t)std::min(std::max(v, 0), 255); };
  for (size_t i = 0; i < total; i++) {
    int R = in[3 * i], G = in[3 * i + 1], B = in[3 * i + 2];
    int yv = (R * 4899 + G * 9617 + B * 1868 + half) >> shift;
    int u = ((B - yv) * 8061 + delta + half) >> shift;
    int v = ((R - yv) * 14369 + delta + half) >> shift;
    yuv[3 * i] = sat(yv); yuv[3 * i + 1] = sat(u); yuv[3 * i + 2] = sat(v);
  }
}
static void bgr_to_yuyv(const uint8_t* in, int w, int h, uint8_t* out) {
  size_t total = (size_t)w * h;
  std::vector<uint8_t> yuv(3 * total), Y(total), U(total), V(total);
  rgb2yuv_u8(in, total, yuv.data());                            // RGB2YUV on BGR-ordered bytes (deepseg.cc:89)
  for (size_t i = 0; i < total; i++) { Y[i] = yuv[3 * i]; U[i] = yuv[3 * i + 1]; V[i] = yuv[3 * i + 2]; }   // cv::split (:91)
  for (size_t i = 0; i + 1 < total; i += 2) {
    uint8_t u = (uint8_t)(((int)U[i] + (int)U[i + 1]) / 2), v = (uint8_t)(((int)V[i] + (int)V[i + 1]) / 2);
    out[2 * i] = Y[i]; out[2 * i + 1] = v; out[2 * i + 2] = Y[i + 1]; out[2 * i + 3] = u;
  }
}

// YUYV (YUY2, bytes Y0 U Y1 V) → BGR — the ingest conversion OpenCV's VideoCapture performs for app/deepseg.cc:553
// (CAP_PROP_CONVERT_RGB) and the explicit cv::cvtColor(COLOR_YUV2BGR_YUYV) of the debug view (:725).  OpenCV 8u path
// (color_yuv, ITU-R BT.601 limited range, 20-bit fixed point): CY 1220542, CUB 2116026, CUG -409993, CVG -852492,
// CVR 1673527;  y = max(0, Y-16)*CY;  c = sat((y + (1<<19) + coeffs·(U-128, V-128)) >> 20).
static void yuyv_to_bgr(const uint8_t* in, int w, int h, uint8_t* out) {
  const int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
  auto sat = [](int v) { return (uint8_t)std::min(std::max(v, 0), 255); };
  size_t pairs = (size_t)w * h / 2;
  for (size_t i = 0; i < pairs; i++) {
    int y0 = in[4 * i], u = in[4 * i + 1] - 128, y1 = in[4 * i + 2], v = in[4 * i + 3] - 128;
    int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
    int ya = std::max(0, y0 - 16) * CY, yb = std::max(0, y1 - 16) * CY;
    uint8_t* o = out + 6 * i;
    o[0] = sat((ya + buv) >> SH); o[1] = sat((ya + guv) >> SH); o[2] = sat((ya + ruv) >> SH);
    o[3] = sat((yb + buv) >> SH); o[4] = sat((yb + guv) >> SH); o[5] = sat((yb + ruv) >> SH);
  }
}

// decode + temporal IIR — follows lib/libbackscrub.cc:317-357.
//   type 1 DeepLab: 21-way argmax (first max wins, init -10000), person==15 → 0 else 255
//   type 2 MLKit/BodyPix: p > 0.65 (double compare) → 0 else 255
//   type 3 Meet: e0=expf(l0), e1=expf(l1); (e0/(e0+e1) < e1/(e0+e1)) → 0 else 255
//   out[n] = (val & 0xE0) | (out[n] >> 3)
static void decode_iir(int type, const float* t, size_t npix, int nch, uint8_t* out) {
  for (size_t n = 0; n < npix; n++) {
    uint8_t val = 255;
    if (type == 1) {
      float maxval = -10000; size_t maxpos = 0;
      for (int i = 0; i < nch; i++) if (t[n * nch + i] > maxval) { maxval = t[n * nch + i]; maxpos = i; }
      val = maxpos == 15 ? 0 : 255;
    } else if (type == 2) {
      val = ((double)t[n] > 0.65) ? 0 : 255;
    } else if (type == 3) {
      float e0 = expf(t[2 * n]), e1 = expf(t[2 * n + 1]);
      float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      val = p0 < p1 ? 0 : 255;
    }
    out[n] = (uint8_t)((val & 0xE0) | (out[n] >> 3));
  }
}

// =======================================================================================
// 4. Context — mirrors backscrub_ctx_t and bs_maskgen_new/process (lib/libbackscrub.cc:28-54,
//    161-259, 279-376).
// =======================================================================================
struct Rect { int x, y, w, h; };
struct OCtx {
  OModel model;
  int modeltype = 0;  // 1 deeplab, 2 mlkit/bodypix, 3 meet
  float scaling = 0, offset = 0;
  int width = 0, height = 0, inW = 0, inH = 0, inC = 0, outW = 0, outH = 0, outC = 0;
  Rect roidim{}, in_roidim{};
  std::vector<uint8_t> mask, ofinal, in_u8_bgr, in_u8_rgb, filtered, tmpbuf, blurred;
  std::vector<float> input, output;
};

static int model_type_from_name(const std::string& n) {  // lib/libbackscrub.cc:116-130
  if (n.find("body-pix") != n.npos) return 4;
  if (n.find("deeplab") != n.npos) return 1;
  if (n.find("segm_") != n.npos) return 3;
  if (n.find("selfie") != n.npos) return 2;
  return 0;
}

static OCtx* ctx_new(const char* path, int width, int height) {
  std::unique_ptr<OCtx> c(new OCtx);
  if (!load_model(path, c->model)) return nullptr;
  fold_constants(c->model);
  c->modeltype = model_type_from_name(path);
  if (c->modeltype == 0) return nullptr;
  if (c->modeltype == 1) { c->scaling = (float)(1 / 127.5); c->offset = -1; }  // :132-148
  else { c->scaling = (float)(1 / 255.0); c->offset = 0; }
  if (c->modeltype == 4) c->modeltype = 2;  // BodyPix shares the threshold decode (:333)
  const OTensor& ti = c->model.t[c->model.inputs[0]];
  const OTensor& to = c->model.t[c->model.outputs[0]];
  if (ti.shape.size() != 4 || to.shape.size() != 4) return nullptr;
  c->inH = ti.shape[1]; c->inW = ti.shape[2]; c->inC = ti.shape[3];
  c->outH = to.shape[1]; c->outW = to.shape[2]; c->outC = to.shape[3];
  c->width = width; c->height = height;
  // :230-246 — float arithmetic with int truncation into cv::Rect
  float ratio = (float)c->inH / (float)c->inW;
  float frameratio = (float)height / (float)width;
  size_t uw = (size_t)width, uh = (size_t)height;
  if (frameratio < ratio) {
    c->roidim = Rect{(int)((uw - uh / ratio) / 2), 0, (int)(uh / ratio), height};
    c->in_roidim = Rect{0, 0, c->inW, c->inH};
  } else {
    c->roidim = Rect{0, 0, width, height};
    c->in_roidim = Rect{(int)((c->inW - c->inH / frameratio) / 2), 0, (int)(c->inH / frameratio), c->inH};
  }
  c->mask.assign((size_t)width * height, 255);                   // :248
  c->in_u8_bgr.assign((size_t)c->inW * c->inH * 3, 0);           // :251
  c->ofinal.assign((size_t)c->outW * c->outH, 0);                // :257 (uninitialised there; 0 here)
  c->in_u8_rgb.resize(c->in_u8_bgr.size()); c->filtered.resize(c->in_u8_bgr.size());
  c->input.resize((size_t)c->inW * c->inH * c->inC); c->output.resize((size_t)c->outW * c->outH * c->outC);
  c->tmpbuf.resize((size_t)c->roidim.w * c->roidim.h); c->blurred.resize(c->tmpbuf.size());
  return c.release();
}

static void ctx_prep(OCtx& c, const uint8_t* frame) {  // :285-302
  const Rect& r = c.roidim; const Rect& q = c.in_roidim;
  resize_linear_u8(frame + ((size_t)r.y * c.width + r.x) * 3, r.w, r.h, (size_t)c.width * 3, 3,
                   c.in_u8_bgr.data() + ((size_t)q.y * c.inW + q.x) * 3, q.w, q.h, (size_t)c.inW * 3);
  size_t np = (size_t)c.inW * c.inH;
  for (size_t i = 0; i < np; i++) { c.in_u8_rgb[3 * i] = c.in_u8_bgr[3 * i + 2]; c.in_u8_rgb[3 * i + 1] = c.in_u8_bgr[3 * i + 1]; c.in_u8_rgb[3 * i + 2] = c.in_u8_bgr[3 * i]; }
  bilateral_c3(c.in_u8_rgb.data(), c.inW, c.inH, c.filtered.data(), 5, 100.0, 100.0);
  for (size_t i = 0; i < np * 3; i++) c.input[i] = (float)c.filtered[i] * c.scaling + c.offset;  // convertTo: v*alpha+beta in f32
}

static void ctx_post(OCtx& c) {  // :314-371
  decode_iir(c.modeltype, c.output.data(), (size_t)c.outW * c.outH, c.outC, c.ofinal.data());
  const Rect& r = c.roidim; const Rect& q = c.in_roidim;
  resize_linear_u8(c.ofinal.data() + (size_t)q.y * c.outW + q.x, q.w, q.h, (size_t)c.outW, 1, c.tmpbuf.data(), r.w, r.h, (size_t)r.w);
  blur5_u8(c.tmpbuf.data(), r.w, r.h, (size_t)r.w, c.mask.data() + (size_t)r.y * c.width + r.x, (size_t)c.width);
}

static bool ctx_process(OCtx& c, const uint8_t* frame, uint8_t* mask_out) {
  ctx_prep(c, frame);
  if (!invoke(c.model, c.input.data(), c.output.data())) return false;
  ctx_post(c);
  if (mask_out) memcpy(mask_out, c.mask.data(), c.mask.size());
  return true;
}

}  // namespace

// =======================================================================================
// C API (ctypes-friendly)
// =======================================================================================
extern "C" {

const char* bso_version(void) { return "bs_oracle 2 (CPU restatement; in-tree reference code pinned by oracle/_ref, OpenCV/TFLite semantics unpinned)"; }

void* bso_model_load(const char* path) {
  OModel* m = new OModel;
  if (!load_model(path, *m)) { delete m; return nullptr; }
  fold_constants(*m);
  return m;
}
void bso_model_free(void* h) { delete (OModel*)h; }
int bso_model_num_ops(void* h) { return (int)((OModel*)h)->ops.size(); }
int bso_model_num_tensors(void* h) { return (int)((OModel*)h)->t.size(); }
int bso_model_input(void* h) { return ((OModel*)h)->inputs[0]; }
int bso_model_output(void* h) { return ((OModel*)h)->outputs[0]; }
// shape of tensor i into out4 (padded with 1s on the left); returns rank
int bso_model_tensor_shape(void* h, int i, int* out4) {
  OModel* m = (OModel*)h; const auto& s = m->t[i].shape;
  for (int k = 0; k < 4; k++) out4[k] = 1;
  for (size_t k = 0; k < s.size() && k < 4; k++) out4[4 - s.size() + k] = s[k];
  return (int)s.size();
}
// copy tensor data (valid after invoke, or any time for constants); returns element count
long bso_model_tensor_data(void* h, int i, float* out, long cap) {
  OModel* m = (OModel*)h; const auto& f = m->t[i].f;
  long n = std::min<long>((long)f.size(), cap);
  if (out) memcpy(out, f.data(), n * sizeof(float));
  return (long)f.size();
}
// op record: [code, folded, n_in, in0..in3, out0, padding, stride_w, stride_h, act, dil_w, dil_h, depth_mult, filter_w, filter_h, axis, align, half_pixel]
int bso_model_op(void* h, int i, int* rec24) {
  OModel* m = (OModel*)h; const OOp& op = m->ops[i];
  for (int k = 0; k < 24; k++) rec24[k] = -1;
  rec24[0] = op.code; rec24[1] = op.folded; rec24[2] = (int)op.in.size();
  for (size_t k = 0; k < op.in.size() && k < 4; k++) rec24[3 + k] = op.in[k];
  rec24[7] = op.out[0];
  int v[] = {op.padding, op.stride_w, op.stride_h, op.act, op.dil_w, op.dil_h, op.depth_mult, op.filter_w, op.filter_h, op.axis, op.align_corners, op.half_pixel};
  for (int k = 0; k < 12; k++) rec24[8 + k] = v[k];
  return 0;
}
int bso_model_invoke(void* h, const float* in, float* out) { return invoke(*(OModel*)h, in, out) ? 0 : -1; }

void bso_resize_linear_u8(const uint8_t* src, int sw, int sh, long sstride, int cn, uint8_t* dst, int dw, int dh, long dstride) {
  resize_linear_u8(src, sw, sh, (size_t)sstride, cn, dst, dw, dh, (size_t)dstride);
}
void bso_bilateral_c3(const uint8_t* src, int w, int h, uint8_t* dst, int d, double sc, double ss) { bilateral_c3(src, w, h, dst, d, sc, ss); }
void bso_blur5_u8(const uint8_t* src, int w, int h, long sstride, uint8_t* dst, long dstride) { blur5_u8(src, w, h, (size_t)sstride, dst, (size_t)dstride); }
void bso_alpha_blend(const uint8_t* bg, const uint8_t* fr, const uint8_t* m, uint8_t* out, long npix) { alpha_blend(bg, fr, m, out, (size_t)npix); }
void bso_bgr_to_yuyv(const uint8_t* in, int w, int h, uint8_t* out) { bgr_to_yuyv(in, w, h, out); }
void bso_rgb2yuv_u8(const uint8_t* in, long npix, uint8_t* yuv) { rgb2yuv_u8(in, (size_t)npix, yuv); }
void bso_set_custom_op_hook(bso_custom_fn fn) { g_custom_hook = fn; }
void bso_gaussian_blur_c3(const uint8_t* src, int w, int h, int n, uint8_t* dst) { gaussian_blur_c3(src, w, h, n, dst); }
int bso_gaussian_coeffs(int n, uint16_t* out) { const std::vector<uint16_t> c = gaussian_coeffs_u16(n); memcpy(out, c.data(), c.size() * 2); return (int)c.size(); }
// the Convolution2DTransposeBias restatement on its own (custom options = padding, stride_w, stride_h); y == NULL queries the shape
int bso_tconv_bias(const float* x, const int* xs4, const float* w, const int* ws4, const float* b, int padding, int stride_w, int stride_h, float* y, int* ys4) {
  OTensor tx, tw, tb, ty; OOp op;
  tx.shape.assign(xs4, xs4 + 4); tx.f.assign(x, x + tx.count());
  tw.shape.assign(ws4, ws4 + 4); tw.f.assign(w, w + tw.count());
  tb.shape = {ws4[0]}; tb.f.assign(b, b + ws4[0]);
  const int32_t v[3] = {padding, stride_w, stride_h};
  op.custom_opts.assign((const uint8_t*)v, (const uint8_t*)v + 12);
  op_tconv_bias(tx, tw, tb, ty, op);
  for (int k = 0; k < 4; k++) ys4[k] = ty.shape[k];
  if (y) memcpy(y, ty.f.data(), ty.f.size() * sizeof(float));
  return 0;
}
void bso_yuyv_to_bgr(const uint8_t* in, int w, int h, uint8_t* out) { yuyv_to_bgr(in, w, h, out); }
void bso_decode_iir(int type, const float* t, long npix, int nch, uint8_t* out) { decode_iir(type, t, (size_t)npix, nch, out); }
void bso_convert_f32(const uint8_t* in, long n, float scale, float off, float* out) { for (long i = 0; i < n; i++) out[i] = (float)in[i] * scale + off; }

void* bso_ctx_new(const char* model_path, int width, int height) { return ctx_new(model_path, width, height); }
void bso_ctx_delete(void* h) { delete (OCtx*)h; }
int bso_ctx_process(void* h, const uint8_t* frame, uint8_t* mask_out) { return ctx_process(*(OCtx*)h, frame, mask_out) ? 0 : -1; }
// [modeltype, inW, inH, inC, outW, outH, outC, roi.x,y,w,h, in_roi.x,y,w,h]
void bso_ctx_geometry(void* h, int* g15) {
  OCtx* c = (OCtx*)h;
  int v[] = {c->modeltype, c->inW, c->inH, c->inC, c->outW, c->outH, c->outC, c->roidim.x, c->roidim.y, c->roidim.w, c->roidim.h,
             c->in_roidim.x, c->in_roidim.y, c->in_roidim.w, c->in_roidim.h};
  memcpy(g15, v, sizeof(v));
}
const float* bso_ctx_input(void* h) { return ((OCtx*)h)->input.data(); }
const float* bso_ctx_output(void* h) { return ((OCtx*)h)->output.data(); }
const uint8_t* bso_ctx_ofinal(void* h) { return ((OCtx*)h)->ofinal.data(); }
const uint8_t* bso_ctx_mask(void* h) { return ((OCtx*)h)->mask.data(); }
void* bso_ctx_model(void* h) { return &((OCtx*)h)->model; }
void bso_ctx_set_ofinal(void* h, const uint8_t* v) { OCtx* c = (OCtx*)h; memcpy(c->ofinal.data(), v, c->ofinal.size()); }
// stages run separately (used to test GPU stages one at a time on identical inputs)
void bso_ctx_prep(void* h, const uint8_t* frame) { ctx_prep(*(OCtx*)h, frame); }
int bso_ctx_infer(void* h) { OCtx* c = (OCtx*)h; return invoke(c->model, c->input.data(), c->output.data()) ? 0 : -1; }
void bso_ctx_set_output(void* h, const float* logits) { OCtx* c = (OCtx*)h; memcpy(c->output.data(), logits, c->output.size() * sizeof(float)); }
void bso_ctx_post(void* h) { ctx_post(*(OCtx*)h); }

// CPU baseline: `n_streams` independent contexts, `iters` frames each, full per-frame path
// (process + alpha_blend against a shared background), OpenMP over streams.
// frames: [n_streams][H][W][3]; bg: [H][W][3]; out: [n_streams][H][W][3] (last iteration).
// Returns elapsed seconds (wall) or <0 on error; stage_s[4] = prep, infer, post, blend seconds (summed over threads).
double bso_baseline_run(const char* model_path, int width, int height, int n_streams, int iters, int threads,
                        const uint8_t* frames, const uint8_t* bg, uint8_t* out, double* stage_s) {
  // one context per stream, created (and therefore first-touched: its tensors land on the NUMA node of the core that uses them) by the thread that runs it
  std::vector<OCtx*> ctxs(n_streams, nullptr);
  size_t fsz = (size_t)width * height * 3;
  double st[4] = {0, 0, 0, 0};
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int s = 0; s < n_streams; s++) { ctxs[s] = ctx_new(model_path, width, height); if (!ctxs[s]) bad++; }
  if (bad) { for (auto* c : ctxs) delete c; return -1.0; }
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(static) reduction(+ : st[:4])
  for (int s = 0; s < n_streams; s++) {
    OCtx& c = *ctxs[s];
    std::vector<uint8_t> local(out ? 0 : fsz);
    uint8_t* o = out ? out + s * fsz : local.data();
    for (int it = 0; it < iters; it++) {
      auto a = std::chrono::steady_clock::now();
      ctx_prep(c, frames + s * fsz);
      auto b = std::chrono::steady_clock::now();
      invoke(c.model, c.input.data(), c.output.data());
      auto d = std::chrono::steady_clock::now();
      ctx_post(c);
      auto e = std::chrono::steady_clock::now();
      alpha_blend(bg, frames + s * fsz, c.mask.data(), o, (size_t)width * height);
      auto f = std::chrono::steady_clock::now();
      st[0] += std::chrono::duration<double>(b - a).count(); st[1] += std::chrono::duration<double>(d - b).count();
      st[2] += std::chrono::duration<double>(e - d).count(); st[3] += std::chrono::duration<double>(f - e).count();
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  for (auto* c : ctxs) delete c;
  if (stage_s) for (int k = 0; k < 4; k++) stage_s[k] = st[k];
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
