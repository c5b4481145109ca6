// tflite_model.cpp — see tflite_model.hpp.  Flatbuffer layout per the public TFLite
// schema v3 (field ids recorded in SURVEY.md Appendix A).  Bounds-checked: a truncated or
// hostile file yields an error string, never an out-of-range read.
#include "tflite_model.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

namespace bsx {
namespace {

class Reader {
 public:
  Reader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  bool ok() const { return ok_; }

  template <class T> T get(size_t off) {
    T v{};
    if (off > n_ || n_ - off < sizeof(T)) { ok_ = false; return v; }
    std::memcpy(&v, p_ + off, sizeof(T));
    return v;
  }
  size_t follow(size_t off) { return off + get<uint32_t>(off); }
  // absolute position of table field `id`, 0 when absent
  size_t slot(size_t table, int id) {
    size_t vt = table - (size_t)(int64_t)get<int32_t>(table);
    uint16_t vt_len = get<uint16_t>(vt);
    size_t entry = 4 + 2 * (size_t)id;
    if (entry + 2 > vt_len) return 0;
    uint16_t rel = get<uint16_t>(vt + entry);
    return rel ? table + rel : 0;
  }
  template <class T> T field(size_t table, int id, T dflt) {
    size_t s = slot(table, id);
    return s ? get<T>(s) : dflt;
  }
  size_t sub(size_t table, int id) { size_t s = slot(table, id); return s ? follow(s) : 0; }
  struct Span { size_t at = 0; uint32_t len = 0; };
  Span vec(size_t table, int id) {
    Span sp;
    size_t s = slot(table, id);
    if (!s) return sp;
    size_t v = follow(s);
    sp.len = get<uint32_t>(v);
    sp.at = v + 4;
    return sp;
  }
  bool in_range(size_t at, size_t bytes) const { return at <= n_ && n_ - at >= bytes; }
  const uint8_t* raw(size_t at) const { return p_ + at; }
  std::string text(size_t table, int id) {
    Span s = vec(table, id);
    if (!s.at || !in_range(s.at, s.len)) return std::string();
    return std::string((const char*)p_ + s.at, s.len);
  }
  std::vector<int> ints(size_t table, int id) {
    Span s = vec(table, id);
    std::vector<int> out;
    if (!s.at || !in_range(s.at, 4ull * s.len)) return out;
    out.resize(s.len);
    for (uint32_t i = 0; i < s.len; i++) out[i] = get<int32_t>(s.at + 4ull * i);
    return out;
  }

 private:
  const uint8_t* p_;
  size_t n_;
  bool ok_ = true;
};

float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0x1f) {
    bits = sign | 0x7f800000u | (man << 13);
  } else if (exp != 0) {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  } else if (man == 0) {
    bits = sign;
  } else {  // subnormal half → normal float
    int shift = 0;
    while (!(man & 0x400u)) { man <<= 1; ++shift; }
    man &= 0x3ffu;
    bits = sign | ((uint32_t)(113 - shift) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// ActivationFunctionType: 0 NONE, 1 RELU, 3 RELU6 are implemented; RELU_N1_TO_1 (2), TANH (4), SIGN_BIT (5) are not —
// -1 makes the load fail instead of silently computing the wrong network.
int map_fused_act(int a) { return a == 0 ? kActNone : a == 1 ? kActRelu : a == 3 ? kActRelu6 : -1; }

}  // namespace

bool load_tflite(const std::string& path, Graph* g, std::string* err) {
  auto fail = [&](const std::string& m) { if (err) *err = m; return false; };
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return fail("unable to load model from file: '" + path + "'");
  std::streamsize sz = f.tellg();
  if (sz < 16) return fail("model file too small: '" + path + "'");
  std::vector<uint8_t> bytes((size_t)sz);
  f.seekg(0);
  if (!f.read((char*)bytes.data(), sz)) return fail("short read: '" + path + "'");

  Reader r(bytes.data(), bytes.size());
  size_t root = r.follow(0);
  if (r.field<uint32_t>(root, 0, 0) != 3) return fail("not a schema-v3 tflite file");
  g->description = r.text(root, 3);

  struct Code { int builtin; std::string custom; };
  std::vector<Code> codes;
  {
    auto v = r.vec(root, 1);
    if (!r.in_range(v.at, 4ull * v.len)) return fail("operator-code table out of file range");
    for (uint32_t i = 0; i < v.len; i++) {
      size_t t = r.follow(v.at + 4ull * i);
      int legacy = r.field<int8_t>(t, 0, 0), full = r.field<int32_t>(t, 3, 0);
      codes.push_back({legacy > full ? legacy : full, r.text(t, 1)});
    }
  }
  std::vector<Reader::Span> buffers;
  {
    auto v = r.vec(root, 4);
    if (!r.in_range(v.at, 4ull * v.len)) return fail("buffer table out of file range");
    for (uint32_t i = 0; i < v.len; i++) buffers.push_back(r.vec(r.follow(v.at + 4ull * i), 0));
  }
  // every flatbuffer vector element occupies at least one byte of the file: a length beyond the file size is malformed
  // (and would otherwise turn into an unbounded resize below)
  const size_t file_n = bytes.size();
  auto sgs = r.vec(root, 2);
  if (!sgs.len) return fail("model has no subgraph");
  size_t sg = r.follow(sgs.at);

  // ---- tensors
  {
    auto v = r.vec(sg, 0);
    if (!r.in_range(v.at, 4ull * v.len) || v.len > file_n) return fail("tensor table out of file range");
    g->tensors.resize(v.len);
    for (uint32_t i = 0; i < v.len; i++) {
      size_t t = r.follow(v.at + 4ull * i);
      TensorInfo& ti = g->tensors[i];
      ti.shape = r.ints(t, 0);
      ti.name = r.text(t, 3);
      if (ti.shape.size() > 4) return fail("tensor rank > 4 unsupported");
      uint64_t prod = 1;
      for (size_t k = 0; k < ti.shape.size(); k++) {
        if (ti.shape[k] <= 0) return fail("dynamic / empty tensor shape unsupported");
        ti.dims[4 - ti.shape.size() + k] = ti.shape[k];
        prod *= (uint64_t)ti.shape[k];
        if (prod > (1ull << 31)) return fail("tensor #" + std::to_string(i) + " has more than 2^31 elements");
      }
      int type = r.field<int8_t>(t, 1, 0);
      uint32_t b = r.field<uint32_t>(t, 2, 0);
      if (b < buffers.size() && buffers[b].len) {
        auto sp = buffers[b];
        if (!r.in_range(sp.at, sp.len)) return fail("constant buffer out of file range");
        size_t n = ti.elems();
        ti.is_const = true;
        if (type == 0) {
          if (sp.len < n * 4) return fail("f32 constant shorter than its shape");
          ti.f32.resize(n);
          std::memcpy(ti.f32.data(), r.raw(sp.at), n * 4);
        } else if (type == 1) {
          if (sp.len < n * 2) return fail("f16 constant shorter than its shape");
          ti.f32.resize(n);
          for (size_t k = 0; k < n; k++) { uint16_t h; std::memcpy(&h, r.raw(sp.at + 2 * k), 2); ti.f32[k] = f16_to_f32(h); }
        } else if (type == 2) {
          if (sp.len < n * 4) return fail("i32 constant shorter than its shape");
          ti.i32.resize(n);
          std::memcpy(ti.i32.data(), r.raw(sp.at), n * 4);
        } else {
          return fail("unsupported constant tensor type " + std::to_string(type));
        }
      } else if (type != 0) {
        return fail("activation tensor #" + std::to_string(i) + " is not float32");  // cf. libbackscrub.cc:88-91
      }
    }
  }
  auto ins = r.ints(sg, 1), outs = r.ints(sg, 2);
  if (ins.empty() || outs.empty()) return fail("model has no input/output");
  g->input = ins[0];
  g->output = outs[0];
  if (g->input < 0 || g->input >= (int)g->tensors.size() || g->output < 0 || g->output >= (int)g->tensors.size())
    return fail("input/output tensor index out of range");

  // ---- operators
  auto ops = r.vec(sg, 3);
  if (!r.in_range(ops.at, 4ull * ops.len)) return fail("operator table out of file range");
  g->n_file_ops = (int)ops.len;
  for (uint32_t i = 0; i < ops.len; i++) {
    size_t o = r.follow(ops.at + 4ull * i);
    uint32_t ci = r.field<uint32_t>(o, 0, 0);
    if (ci >= codes.size()) return fail("operator code index out of range");
    Node n;
    n.index = (int)i;
    n.inputs = r.ints(o, 1);
    auto oo = r.ints(o, 2);
    if (oo.size() != 1) return fail("operator with != 1 outputs unsupported");
    n.output = oo[0];
    for (int t : n.inputs) if (t >= (int)g->tensors.size() || t < -1) return fail("operator input index out of range");
    if (n.output < 0 || n.output >= (int)g->tensors.size()) return fail("operator output index out of range");
    size_t opt = r.sub(o, 4);
    int code = codes[ci].builtin;
    switch (code) {
      case 3:  // CONV_2D
        n.type = OpType::Conv;
        if (opt) {
          n.same_padding = r.field<int8_t>(opt, 0, 0) == 0;
          n.stride_w = r.field<int32_t>(opt, 1, 1); n.stride_h = r.field<int32_t>(opt, 2, 1);
          n.act = map_fused_act(r.field<int8_t>(opt, 3, 0));
          n.dil_w = r.field<int32_t>(opt, 4, 1); n.dil_h = r.field<int32_t>(opt, 5, 1);
        }
        break;
      case 4:  // DEPTHWISE_CONV_2D
        n.type = OpType::DwConv;
        if (opt) {
          n.same_padding = r.field<int8_t>(opt, 0, 0) == 0;
          n.stride_w = r.field<int32_t>(opt, 1, 1); n.stride_h = r.field<int32_t>(opt, 2, 1);
          n.depth_mult = r.field<int32_t>(opt, 3, 1);
          n.act = map_fused_act(r.field<int8_t>(opt, 4, 0));
          n.dil_w = r.field<int32_t>(opt, 5, 1); n.dil_h = r.field<int32_t>(opt, 6, 1);
        }
        break;
      case 1:  // AVERAGE_POOL_2D
        n.type = OpType::AvgPool;
        if (opt) {
          n.same_padding = r.field<int8_t>(opt, 0, 0) == 0;
          n.stride_w = r.field<int32_t>(opt, 1, 1); n.stride_h = r.field<int32_t>(opt, 2, 1);
          n.filter_w = r.field<int32_t>(opt, 3, 0); n.filter_h = r.field<int32_t>(opt, 4, 0);
          n.act = map_fused_act(r.field<int8_t>(opt, 5, 0));
        }
        break;
      case 9:  // FULLY_CONNECTED
        n.type = OpType::FullyConnected;
        if (opt) n.act = map_fused_act(r.field<int8_t>(opt, 0, 0));
        break;
      case 2:
        n.type = OpType::Concat;
        if (opt) { n.axis = r.field<int32_t>(opt, 0, 0); n.act = map_fused_act(r.field<int8_t>(opt, 1, 0)); if (n.act != kActNone) return fail("fused activation on CONCATENATION unsupported (op #" + std::to_string(i) + ")"); }
        break;
      case 0: n.type = OpType::Add; if (opt) n.act = map_fused_act(r.field<int8_t>(opt, 0, 0)); break;
      case 18: n.type = OpType::Mul; if (opt) n.act = map_fused_act(r.field<int8_t>(opt, 0, 0)); break;
      case 19: n.type = OpType::Relu; break;
      case 21: n.type = OpType::Relu6; break;
      case 117: n.type = OpType::HardSwish; break;
      case 14: n.type = OpType::Logistic; break;
      case 6: n.type = OpType::Dequantize; break;
      case 23:
        n.type = OpType::ResizeBilinear;
        if (opt) { n.align_corners = r.field<uint8_t>(opt, 2, 0) != 0; n.half_pixel = r.field<uint8_t>(opt, 3, 0) != 0; }
        break;
      case 32:
        if (codes[ci].custom != "Convolution2DTransposeBias")  // lib/libbackscrub.cc:207 registers exactly this one
          return fail("unsupported custom op '" + codes[ci].custom + "'");
        n.type = OpType::TransposeConvBias;
        {
          auto c = r.vec(o, 5);
          if (c.len >= 12 && r.in_range(c.at, 12)) {
            int32_t v[3];
            std::memcpy(v, r.raw(c.at), 12);
            n.tconv_padding_same = v[0] == 1; n.tconv_stride_w = v[1]; n.tconv_stride_h = v[2];
          }
        }
        break;
      default:
        return fail("unsupported builtin operator code " + std::to_string(code) + " at op #" + std::to_string(i));
    }
    if (!r.ok()) return fail("malformed flatbuffer (read out of range) at op #" + std::to_string(i));
    if (n.act < 0) return fail("unsupported fused activation at op #" + std::to_string(i));
    if (n.stride_w < 1 || n.stride_h < 1 || n.dil_w < 1 || n.dil_h < 1 || n.depth_mult < 1 || n.tconv_stride_w < 1 || n.tconv_stride_h < 1 ||
        n.stride_w > 4096 || n.stride_h > 4096 || n.dil_w > 4096 || n.dil_h > 4096 || n.tconv_stride_w > 4096 || n.tconv_stride_h > 4096 || n.filter_w < 0 || n.filter_h < 0)
      return fail("operator #" + std::to_string(i) + " has a non-positive / absurd stride, dilation or filter size");
    // -1 marks an OPTIONAL operand (a bias); the data operands every kernel dereferences must exist
    {
      size_t mandatory = 1;
      switch (n.type) {
        case OpType::Conv: case OpType::DwConv: case OpType::FullyConnected: case OpType::Add: case OpType::Mul: case OpType::ResizeBilinear: mandatory = 2; break;
        case OpType::TransposeConvBias: mandatory = 3; break;
        case OpType::Concat: mandatory = n.inputs.size(); break;
        default: mandatory = 1; break;
      }
      if (n.inputs.size() < mandatory) return fail("operator #" + std::to_string(i) + " has too few inputs");
      for (size_t k = 0; k < mandatory; k++) if (n.inputs[k] < 0) return fail("operator #" + std::to_string(i) + " lacks a mandatory input");
    }

    // fold constant-only DEQUANTIZE (weights): the f16 payload was already widened on load
    if (n.type == OpType::Dequantize) {
      if (n.inputs.empty() || n.inputs[0] < 0) return fail("DEQUANTIZE without input");
      const TensorInfo& src = g->tensors[n.inputs[0]];
      if (!src.is_const) return fail("DEQUANTIZE of a non-constant tensor unsupported");
      TensorInfo& dst = g->tensors[n.output];
      dst.is_const = true;
      dst.f32 = src.f32;
      continue;
    }
    g->nodes.push_back(std::move(n));
  }
  if (!r.ok()) return fail("malformed flatbuffer (read out of range)");
  return true;
}

}  // namespace bsx
