// kernels_frame.hip — the per-frame network program (see frame_program.hpp).
//
// One 1024-lane workgroup = one camera frame = one CU.  The workgroup interprets the fused step
// list; tensors live in the CU's LDS ([pixel][C+pad] rows) unless the planner spilled them to
// the frame's slice of the HBM arena.  Weights are read with wave-uniform addresses (scalar
// loads + SGPR-operand FMAs): a wave always works on ONE output-channel tile, lanes are pixels.
//
// Numerics are those of the per-launch kernels (kernels_nn.hip): ci-ascending FMA chains with the
// bias added last, un-contracted bilinear taps; only the single-pixel GEMV steps and the global
// average pools use tree reductions.
#include "frame_program.hpp"
#include "kernels.hpp"

namespace bsx {
namespace {

extern __shared__ __attribute__((aligned(16))) float smem[];

struct FrameCtx {
  float* arena;        // this frame's private slice
  float* net_in;       // batch-major network input  [n][inH][inW][inC]
  float* net_out;      // batch-major network output [n][outH][outW][outC]
  const float* weights;
  int frame;
};

__device__ __forceinline__ float fp_act(float v, int act) {
  switch (act) {
    case kActRelu: return fmaxf(v, 0.f);
    case kActRelu6: return fminf(fmaxf(v, 0.f), 6.f);
    case kActHswish: return v * fminf(6.f, fmaxf(0.f, v + 3.f)) / 6.f;
    case kActSigmoid: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

typedef __attribute__((address_space(4))) const Loc cloc_t;
__device__ __forceinline__ float* loc_ptr(cloc_t& l, const FrameCtx& c) {
  const int space = l.space, off = l.off, elems = l.elems;
  if (space == kLocLds) return smem + off;
  if (space == kLocGlobal) return c.arena + off;
  if (space == kLocInput) return c.net_in + (size_t)c.frame * (size_t)elems;
  if (space == kLocOutput) return c.net_out + (size_t)c.frame * (size_t)elems;
  return nullptr;
}

// Weights never change while a kernel runs: reading them through the CONSTANT address space makes every
// wave-uniform weight access a scalar load (s_load_dwordx8/16 into SGPRs) regardless of what alias analysis
// can prove about the generic (LDS-or-HBM) activation pointers around it.
typedef __attribute__((address_space(4))) const float cfloat_t;
// The micro-op table is immutable too; reading it through the constant address space keeps every field
// (dims, offsets, weight bases) wave-uniform in SGPRs — loads through generic pointers would be treated as divergent.
typedef __attribute__((address_space(4))) const MicroOp cop_t;
__device__ __forceinline__ cfloat_t* as_const(const float* p) { return (cfloat_t*)p; }

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- 1x1 convolution: wave = (64-pixel chunk, CT-channel tile), lane = pixel ------------------------
template <int CT>
__device__ __forceinline__ void mo_pw(cop_t& op, const FrameCtx& c) {
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const float* res = loc_ptr(op.res, c);
  const float* sc = loc_ptr(op.scale, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int xs = op.in0.stride, ys = op.out.stride, rs = op.res.stride;
  const int P = op.OH * op.OW, chunks = (P + 63) >> 6, tiles = op.cout_pad / CT;
  const int lane = threadIdx.x & 63, nw = kFrameThreads >> 6;
  const int Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act;
  for (int wi = wave_id(); wi < chunks * tiles; wi += nw) {
    const int tile = wi / chunks, chunk = wi - tile * chunks;
    const int p = (chunk << 6) + lane;
    if (p >= P) continue;
    const int co0 = tile * CT;
    const float* xp = x + (size_t)p * xs;
    cfloat_t* wp = as_const(w + co0);
    float acc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) acc[t] = 0.f;
    for (int ci = 0; ci < Cin; ci += 4) {
      float4 xv = *reinterpret_cast<const float4*>(xp + ci);
      if (sc) {
        float4 sv = *reinterpret_cast<const float4*>(sc + ci);
        xv.x *= sv.x; xv.y *= sv.y; xv.z *= sv.z; xv.w *= sv.w;
      }
      cfloat_t* w0 = wp + (size_t)ci * cout_pad;
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.x, w0[t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.y, w0[cout_pad + t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.z, w0[2 * cout_pad + t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.w, w0[3 * cout_pad + t], acc[t]);
    }
    float* yp = y + (size_t)p * ys + co0;
    const float* rp = res ? res + (size_t)p * rs + co0 : nullptr;
    cfloat_t* bp = as_const(bias + co0);
    if ((Cout & 3) == 0) {
#pragma unroll
      for (int t = 0; t < CT; t += 4) {
        if (co0 + t < Cout) {
          float4 v;
          v.x = fp_act(acc[t] + bp[t], act); v.y = fp_act(acc[t + 1] + bp[t + 1], act);
          v.z = fp_act(acc[t + 2] + bp[t + 2], act); v.w = fp_act(acc[t + 3] + bp[t + 3], act);
          if (rp) { float4 r = *reinterpret_cast<const float4*>(rp + t); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
          *reinterpret_cast<float4*>(yp + t) = v;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < CT; t++) {
        if (co0 + t < Cout) {
          float v = fp_act(acc[t] + bp[t], act);
          if (rp) v += rp[t];
          yp[t] = v;
        }
      }
    }
  }
}

// ---- 1x1 convolution on <= 4 pixels (SE / gate FCs): wave = one output value, lanes split K ----------
__device__ __forceinline__ void mo_gemv(cop_t& op, const FrameCtx& c) {
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const float* res = loc_ptr(op.res, c);
  const float* sc = loc_ptr(op.scale, c);
  const float* w = c.weights + op.w2_off;
  const float* bias = c.weights + op.b_off;
  const int P = op.OH * op.OW, Cin = op.Cin, Cout = op.Cout;
  const int lane = threadIdx.x & 63, nw = kFrameThreads >> 6;
  for (int item = wave_id(); item < P * Cout; item += nw) {
    const int p = item / Cout, co = item - p * Cout;
    const float* xp = x + (size_t)p * op.in0.stride;
    const float* wr = w + (size_t)co * Cin;
    float acc = 0.f;
    for (int ci = lane; ci < Cin; ci += 64) {
      float xv = xp[ci];
      if (sc) xv *= sc[ci];
      acc = fmaf(xv, wr[ci], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      float v = fp_act(acc + bias[co], op.act);
      if (res) v += res[(size_t)p * op.res.stride + co];
      y[(size_t)p * op.out.stride + co] = v;
    }
  }
}

// ---- dense k x k convolution (network stems): wave = (chunk, 16-channel tile), lane = output pixel ---------
__device__ __forceinline__ void mo_conv(cop_t& op, const FrameCtx& c) {
  constexpr int CT = 16;
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const float* res = loc_ptr(op.res, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int xs = op.in0.stride, ys = op.out.stride;
  const int P = op.OH * op.OW, chunks = (P + 63) >> 6, tiles = op.cout_pad / CT;
  const int lane = threadIdx.x & 63, nw = kFrameThreads >> 6;
  for (int wi = wave_id(); wi < chunks * tiles; wi += nw) {
    const int tile = wi / chunks, chunk = wi - tile * chunks;
    const int p = (chunk << 6) + lane;
    if (p >= P) continue;
    const int co0 = tile * CT;
    const int oy = p / op.OW, ox = p - oy * op.OW;
    float acc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) acc[t] = 0.f;
    for (int fy = 0; fy < op.kh; fy++) {
      int iy = oy * op.sh - op.pt + fy * op.dh;
      if (iy < 0 || iy >= op.H) continue;
      for (int fx = 0; fx < op.kw; fx++) {
        int ix = ox * op.sw - op.pl + fx * op.dw;
        if (ix < 0 || ix >= op.W) continue;
        const float* xp = x + ((size_t)iy * op.W + ix) * xs;
        cfloat_t* w0 = as_const(w + (size_t)(fy * op.kw + fx) * op.Cin * op.cout_pad + co0);
        for (int ci = 0; ci < op.Cin; ci++) {
          float xv = xp[ci];
#pragma unroll
          for (int t = 0; t < CT; t++) acc[t] = fmaf(xv, w0[(size_t)ci * op.cout_pad + t], acc[t]);
        }
      }
    }
    float* yp = y + (size_t)p * ys + co0;
#pragma unroll
    for (int t = 0; t < CT; t++) {
      if (co0 + t < op.Cout) {
        float v = fp_act(acc[t] + as_const(bias)[co0 + t], op.act);
        if (res) v += res[(size_t)p * op.res.stride + co0 + t];
        yp[t] = v;
      }
    }
  }
}

// ---- depthwise: lane = (output pixel, channel quad) -----------------------------------------------------------------
__device__ __forceinline__ void mo_dw(cop_t& op, const FrameCtx& c) {
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const float* res = loc_ptr(op.res, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int C = op.Cin, C4 = C >> 2, xs = op.in0.stride, ys = op.out.stride, rs = op.res.stride;
  const int total = op.OH * op.OW * C4;
  for (int i = threadIdx.x; i < total; i += kFrameThreads) {
    const int cq = i % C4, p = i / C4, ch = cq * 4;
    const int oy = p / op.OW, ox = p - oy * op.OW;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int fy = 0; fy < op.kh; fy++) {
      int iy = oy * op.sh - op.pt + fy * op.dh;
      if (iy < 0 || iy >= op.H) continue;
      for (int fx = 0; fx < op.kw; fx++) {
        int ix = ox * op.sw - op.pl + fx * op.dw;
        if (ix < 0 || ix >= op.W) continue;
        float4 xv = *reinterpret_cast<const float4*>(x + ((size_t)iy * op.W + ix) * xs + ch);
        float4 wv = *reinterpret_cast<const float4*>(w + (size_t)(fy * op.kw + fx) * C + ch);
        acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
        acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
      }
    }
    float4 b = *reinterpret_cast<const float4*>(bias + ch);
    float4 v;
    v.x = fp_act(acc.x + b.x, op.act); v.y = fp_act(acc.y + b.y, op.act); v.z = fp_act(acc.z + b.z, op.act); v.w = fp_act(acc.w + b.w, op.act);
    if (res) { float4 r = *reinterpret_cast<const float4*>(res + (size_t)p * rs + ch); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    *reinterpret_cast<float4*>(y + (size_t)p * ys + ch) = v;
  }
}

// ---- global average pool of one input into out[coff .. coff+C) (workgroup reduction through the scratch) ------
__device__ void gap_one(const float* x, int xs, int HW, int C, float* out, int coff) {
  float4* scratch = reinterpret_cast<float4*>(smem);  // kLdsScratchFloats = 1024 float4
  const int C4 = C >> 2;
  int CG = 1;
  while (CG * 2 <= C4 && CG * 2 <= 64) CG *= 2;
  const int rows = kFrameThreads / CG;
  const int cg = threadIdx.x % CG, row = threadIdx.x / CG;
  for (int base = 0; base < C4; base += CG) {
    const int cq = base + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < C4)
      for (int p = row; p < HW; p += rows) {
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * xs + cq * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    scratch[threadIdx.x] = acc;
    __syncthreads();
    // tree over rows (rows is a power of two)
    for (int half = rows >> 1; half > 0; half >>= 1) {
      if (row < half) {
        float4 a = scratch[row * CG + cg], b = scratch[(row + half) * CG + cg];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        scratch[row * CG + cg] = a;
      }
      __syncthreads();
    }
    if (row == 0 && cq < C4) {
      float4 t = scratch[cg];
      const float inv = (float)HW;
      t.x /= inv; t.y /= inv; t.z /= inv; t.w /= inv;
      *reinterpret_cast<float4*>(out + coff + cq * 4) = t;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void mo_gap(cop_t& op, const FrameCtx& c) {
  float* out = loc_ptr(op.out, c);
  const int HW = op.H * op.W;
  if (op.n_cat == 0) { gap_one(loc_ptr(op.in0, c), op.in0.stride, HW, op.Cin, out, 0); return; }
  int coff = 0;
  for (int k = 0; k < op.n_cat; k++) { gap_one(loc_ptr(op.cat[k], c), op.cat[k].stride, HW, op.cat_c[k], out, coff); coff += op.cat_c[k]; }
}

// ---- elementwise ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float elt1(float a, float b, float cc, int e) {
  switch (e) {
    case kEltAdd: return a + b;
    case kEltMul: return a * b;
    case kEltMulAdd: return __fadd_rn(__fmul_rn(a, b), cc);
    default: return a;
  }
}
__device__ __forceinline__ void mo_elt(cop_t& op, const FrameCtx& c) {
  const float* a = loc_ptr(op.in0, c);
  const float* b = loc_ptr(op.in1, c);
  const float* d = loc_ptr(op.in2, c);
  float* y = loc_ptr(op.out, c);
  const int C4 = op.Cin >> 2, total = op.H * op.W * C4;
  for (int i = threadIdx.x; i < total; i += kFrameThreads) {
    const int cq = i % C4, p = i / C4, ch = cq * 4;
    float4 av = *reinterpret_cast<const float4*>(a + (size_t)p * op.in0.stride + ch);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), dv = bv;
    if (op.elt != kEltUnary) bv = *reinterpret_cast<const float4*>(op.bcast1 ? b + ch : b + (size_t)p * op.in1.stride + ch);
    if (op.elt == kEltMulAdd) dv = *reinterpret_cast<const float4*>(d + (size_t)p * op.in2.stride + ch);
    float4 v;
    v.x = fp_act(elt1(av.x, bv.x, dv.x, op.elt), op.act); v.y = fp_act(elt1(av.y, bv.y, dv.y, op.elt), op.act);
    v.z = fp_act(elt1(av.z, bv.z, dv.z, op.elt), op.act); v.w = fp_act(elt1(av.w, bv.w, dv.w, op.elt), op.act);
    *reinterpret_cast<float4*>(y + (size_t)p * op.out.stride + ch) = v;
  }
}

// ---- bilinear resize (TFLite reference association) ---------------------------------------------------------------------------
__device__ __forceinline__ void fp_interp(int o, float scale, bool half_pixel, int in_size, float* frac, int* lo, int* hi) {
  float v = half_pixel ? __fadd_rn(__fmul_rn((float)o + 0.5f, scale), -0.5f) : __fmul_rn((float)o, scale);
  float fl = floorf(v);
  *lo = max((int)fl, 0);
  *hi = min((int)ceilf(v), in_size - 1);
  *frac = v - (float)*lo;
}
__device__ __forceinline__ float fp_bilerp(float x00, float x10, float x01, float x11, float dy, float dx) {
  float a = __fmul_rn(__fmul_rn(x00, 1.f - dy), 1.f - dx);
  float b = __fmul_rn(__fmul_rn(x10, dy), 1.f - dx);
  float cc = __fmul_rn(__fmul_rn(x01, 1.f - dy), dx);
  float d = __fmul_rn(__fmul_rn(x11, dy), dx);
  return __fadd_rn(__fadd_rn(__fadd_rn(a, b), cc), d);
}
__device__ __forceinline__ void mo_resize(cop_t& op, const FrameCtx& c) {
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const int xs = op.in0.stride, ys = op.out.stride;
  float hs = (float)op.H / (float)op.OH, ws = (float)op.W / (float)op.OW;
  if (op.align_corners && op.OH > 1) hs = (float)(op.H - 1) / (float)(op.OH - 1);
  if (op.align_corners && op.OW > 1) ws = (float)(op.W - 1) / (float)(op.OW - 1);
  const bool vec = (op.Cin & 3) == 0;
  const int CV = vec ? op.Cin >> 2 : op.Cin, total = op.OH * op.OW * CV;
  for (int i = threadIdx.x; i < total; i += kFrameThreads) {
    const int cv = i % CV, p = i / CV;
    const int oy = p / op.OW, ox = p - oy * op.OW;
    float dy, dx; int y0, y1, x0, x1;
    fp_interp(oy, hs, op.half_pixel, op.H, &dy, &y0, &y1);
    fp_interp(ox, ws, op.half_pixel, op.W, &dx, &x0, &x1);
    if (vec) {
      const int ch = cv * 4;
      float4 a = *reinterpret_cast<const float4*>(x + ((size_t)y0 * op.W + x0) * xs + ch);
      float4 b = *reinterpret_cast<const float4*>(x + ((size_t)y1 * op.W + x0) * xs + ch);
      float4 cc = *reinterpret_cast<const float4*>(x + ((size_t)y0 * op.W + x1) * xs + ch);
      float4 d = *reinterpret_cast<const float4*>(x + ((size_t)y1 * op.W + x1) * xs + ch);
      float4 v;
      v.x = fp_bilerp(a.x, b.x, cc.x, d.x, dy, dx); v.y = fp_bilerp(a.y, b.y, cc.y, d.y, dy, dx);
      v.z = fp_bilerp(a.z, b.z, cc.z, d.z, dy, dx); v.w = fp_bilerp(a.w, b.w, cc.w, d.w, dy, dx);
      *reinterpret_cast<float4*>(y + (size_t)p * ys + ch) = v;
    } else {
      y[(size_t)p * ys + cv] = fp_bilerp(x[((size_t)y0 * op.W + x0) * xs + cv], x[((size_t)y1 * op.W + x0) * xs + cv],
                                          x[((size_t)y0 * op.W + x1) * xs + cv], x[((size_t)y1 * op.W + x1) * xs + cv], dy, dx);
    }
  }
}

__device__ __forceinline__ void mo_concat(cop_t& op, const FrameCtx& c) {
  float* y = loc_ptr(op.out, c);
  const int P = op.OH * op.OW;
  int coff = 0;
  for (int k = 0; k < op.n_cat; k++) {
    const float* x = loc_ptr(op.cat[k], c);
    const int C4 = op.cat_c[k] >> 2, total = P * C4;
    for (int i = threadIdx.x; i < total; i += kFrameThreads) {
      const int cq = i % C4, p = i / C4;
      *reinterpret_cast<float4*>(y + (size_t)p * op.out.stride + coff + cq * 4) =
          *reinterpret_cast<const float4*>(x + (size_t)p * op.cat[k].stride + cq * 4);
    }
    coff += op.cat_c[k];
  }
}

// ---- Convolution2DTransposeBias, kernel == stride ---------------------------------------------------------------------------------
__device__ __forceinline__ void mo_tconv(cop_t& op, const FrameCtx& c) {
  const float* x = loc_ptr(op.in0, c);
  float* y = loc_ptr(op.out, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int C4 = op.Cin >> 2, P = op.OH * op.OW;
  for (int p = threadIdx.x; p < P; p += kFrameThreads) {
    const int oy = p / op.OW, ox = p - oy * op.OW;
    const int iy = oy / op.kh, fy = oy % op.kh, ix = ox / op.kw, fx = ox % op.kw;
    const float* xp = x + ((size_t)iy * op.W + ix) * op.in0.stride;
    for (int oc = 0; oc < op.Cout; oc++) {
      const float* wp = w + ((size_t)(fy * op.kw + fx) * op.Cout + oc) * op.Cin;
      float acc = bias[oc];
      for (int q = 0; q < C4; q++) {
        float4 xv = *reinterpret_cast<const float4*>(xp + q * 4), wv = *reinterpret_cast<const float4*>(wp + q * 4);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
      }
      y[(size_t)p * op.out.stride + oc] = fp_act(acc, op.act);
    }
  }
}

__global__ __launch_bounds__(kFrameThreads) void frame_program_k(const MicroOp* __restrict__ ops, int n_ops, float* arena, long per_frame_floats,
                                                                float* net_in, float* net_out, const float* __restrict__ weights) {
  FrameCtx c{arena + (size_t)blockIdx.x * (size_t)per_frame_floats, net_in, net_out, weights, (int)blockIdx.x};
  for (int i = 0; i < n_ops; i++) {
    cop_t& op = ((cop_t*)ops)[i];
    switch ((StepKind)op.kind) {
      case StepKind::PwConv:
        if (op.gemv) mo_gemv(op, c);
        else if (op.cout_tile == 8) mo_pw<8>(op, c);
        else if (op.cout_tile == 16) mo_pw<16>(op, c);
        else mo_pw<32>(op, c);
        break;
      case StepKind::Conv: mo_conv(op, c); break;
      case StepKind::DwConv: mo_dw(op, c); break;
      case StepKind::Gap: mo_gap(op, c); break;
      case StepKind::Eltwise: mo_elt(op, c); break;
      case StepKind::Resize: mo_resize(op, c); break;
      case StepKind::Concat: mo_concat(op, c); break;
      case StepKind::TConv: mo_tconv(op, c); break;
    }
    __syncthreads();
  }
}

}  // namespace

hipError_t frame_program_prepare(int lds_floats) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(frame_program_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * (int)sizeof(float));
}

hipError_t launch_frame_program(const MicroOp* d_ops, int n_ops, int lds_floats, float* arena, long per_frame_floats, float* net_in, float* net_out,
                                const float* weights, int n, hipStream_t s) {
  frame_program_k<<<n, kFrameThreads, (size_t)lds_floats * sizeof(float), s>>>(d_ops, n_ops, arena, per_frame_floats, net_in, net_out, weights);
  return hipGetLastError();
}

}  // namespace bsx
