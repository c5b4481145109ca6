// kernels_seg.hip — the spatially-parallel segment kernels of the Meet / MLKit networks (segments.hpp).
//
// Every kernel: grid = (tiles per frame, frames), 256 lanes = 4 waves per workgroup, several workgroups per CU.  A workgroup
// owns one tile of its segment's OUTPUT pixels and recomputes the halo its depthwise 3x3 needs; intermediate tensors of the
// tile live in LDS as dense [pixel][16] rows (conflict-free for both access patterns used here: the quad-transposed MFMA
// epilogue stores and the (pixel, channel-quad) lanes of the depthwise phases), never in HBM.
//
// 1x1 convolutions and the 3x3 stem run on v_mfma_f32_16x16x4_f32 — exact f32 FMA chains, so the results stay within
// float rounding of the reference order (measured against the oracle: < 1e-5 relative on the logits).  Operand maps as
// in kernels_frame.hip: A lane (li, g) = x[pixel li][k = 4g..4g+3]; B lane = w[k][channel li]; D is transposed inside lane
// quads (mfma_tile.hpp) so that a lane ends up with 4 consecutive channels of ONE pixel: 16-byte stores everywhere.
//
// Reference operators covered (file:line into /root/reference): Interpreter::Invoke() lib/libbackscrub.cc:307 — CONV_2D,
// DEPTHWISE_CONV_2D, RESIZE_BILINEAR (half-pixel), MUL/ADD gates, AVERAGE_POOL_2D (as partial sums), FULLY_CONNECTED /
// 1x1 gate convs; Convolution2DTransposeBias lib/transpose_conv_bias.cc:37-114; decode + IIR lib/libbackscrub.cc:317-357.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"
#include "mfma_tile.hpp"
#include "segments.hpp"

namespace bsx {
namespace {

extern __shared__ __attribute__((aligned(16))) float seg_smem[];

constexpr int kScrGate = 0;      // [0,16) gate vector, [16,48) means, [48,80) hidden
constexpr int kScrRed = 96;      // [96, 96 + 2*64) partial-sum meeting points (two sets of 4 waves x 16 channels)
constexpr int kScrFloats = kSegScratchFloats;  // tiles start here

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }

// same forms as the per-frame program (hardware exp2 / rcp)
__device__ __forceinline__ float sg_act(float v, int act) {
  if (act == kActNone) return v;
  if (act == kActSigmoid) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
  const float hi = act == kActRelu ? 3.0e38f : 6.f;
  if (act == kActHswish) return v * fminf(hi, fmaxf(0.f, v + 3.f)) * 0.16666667163372040f;
  return fminf(fmaxf(v, 0.f), hi);
}
__device__ __forceinline__ float4 sg_act4(float4 v, int act) { return make_float4(sg_act(v.x, act), sg_act(v.y, act), sg_act(v.z, act), sg_act(v.w, act)); }

// n / d for 0 <= n < 65536 with m = ceil(2^32 / d)
__device__ __forceinline__ int div_magic(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }
__host__ __device__ inline unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }

// one 16-pixel x 16-channel tile of a 1x1 convolution with Cin = 16: 4 MFMAs
__device__ __forceinline__ f4acc mma16(const float4 a, const float (&wr)[4]) {
  f4acc acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wr[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wr[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wr[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wr[3], acc, 0, 0, 0);
  return acc;
}
// B-operand registers of the [16][cout_pad] block at output channels n0..n0+15: lane (li, g) holds w[4g + r][n0 + li]
__device__ __forceinline__ void load_wtile(float (&wr)[4], const float* __restrict__ w, const SegConvW& c, int n0, int li, int g) {
#pragma unroll
  for (int r = 0; r < 4; r++) wr[r] = w[c.w_off + (long long)(4 * g + r) * c.cout_pad + n0 + li];
}

// ---- gate prologue: means from partial sums → FC → [FC] → s_gate[0..C) ---------------------------------------------------
// All 256 lanes call it.  GAP(a | b) (concatenated) or GAP(a) + GAP(b) (MLKit's GAP(skip + up)).  ONE cooperative memory round
// trip stages the partial sums and both FC weight blocks in LDS (`stage`, <= kGateStageFloats floats, may alias a tile region that
// is not live yet); everything after runs from LDS: a per-workgroup serial chain of dependent global loads here was most of the
// first version's run time.
constexpr int kGateStageFloats = kSegGateStageFloats;
__device__ __forceinline__ void seg_gate(const SegGate& gt, const float* __restrict__ fa, const float* __restrict__ w, float* scr, float* stage) {
  float* s_gate = scr + kScrGate;
  float* s_mean = scr + kScrGate + 16;
  float* s_hid = scr + kScrGate + 48;
  const int tid = threadIdx.x;
  const SegFc &f1 = gt.fc[0], &f2 = gt.fc[1];
  const int w1n = f1.Cin * f1.Cout, w2n = gt.n_fc == 2 ? f2.Cin * f2.Cout : 0;
  float* ps = stage;                      // [2 parts][16 slices][16 channels] slice sums (every pooled tensor here has 16 channels)
  float* w1 = ps + 512;
  float* b1 = w1 + w1n;
  float* w2 = b1 + f1.Cout;
  float* b2 = w2 + w2n;
  {
    const int c = tid & 15, slice = tid >> 4;
    for (int k = 0; k < gt.n_parts; k++) {
      const float* src = fa + gt.part[k].off + c;
      float s = 0.f;
#pragma unroll 4
      for (int i = slice; i < gt.part[k].n; i += 16) s += src[i * 16];
      ps[k * 256 + slice * 16 + c] = s;
    }
  }
  for (int i = tid; i < w1n; i += kSegThreads) w1[i] = w[f1.w_off + i];
  for (int i = tid; i < w2n; i += kSegThreads) w2[i] = w[f2.w_off + i];
  if (tid < f1.Cout) b1[tid] = w[f1.b_off + tid];
  if (gt.n_fc == 2 && tid < f2.Cout) b2[tid] = w[f2.b_off + tid];
  __syncthreads();
  const int Cm = gt.sum_parts ? 16 : 16 * gt.n_parts;
  if (tid < Cm) {
    float m = 0.f;
    for (int k = 0; k < gt.n_parts; k++) {
      if (!gt.sum_parts && (tid >> 4) != k) continue;
      float s = 0.f;
#pragma unroll
      for (int sl = 0; sl < 16; sl++) s += ps[k * 256 + sl * 16 + (tid & 15)];
      m += s / gt.part[k].hw;
    }
    s_mean[tid] = m;
  }
  __syncthreads();
  if (tid < f1.Cout) {
    float acc = 0.f;
    for (int k = 0; k < f1.Cin; k++) acc = fmaf(s_mean[k], w1[tid * f1.Cin + k], acc);
    (gt.n_fc == 1 ? s_gate : s_hid)[tid] = sg_act(acc + b1[tid], f1.act);
  }
  __syncthreads();
  if (gt.n_fc == 2) {
    if (tid < f2.Cout) {
      float acc = 0.f;
      for (int k = 0; k < f2.Cin; k++) acc = fmaf(s_hid[k], w2[tid * f2.Cin + k], acc);
      s_gate[tid] = sg_act(acc + b2[tid], f2.act);
    }
    __syncthreads();
  }
}

// ---- per-tile partial sums of a 16-channel tensor → partials[tile][16] -------------------------------------------------------
// mode 0: MFMA-epilogue lanes (lane (g, li) owns channel quad li >> 2); mode 1: item lanes (lane owns channel quad tid & 3).
template <int MODE>
__device__ __forceinline__ void wave_reduce16(float4 v, float* s_red /* [4 waves][16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto fold = [&](int o) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o); };
  if (MODE == 0) {
    fold(1); fold(2); fold(16); fold(32);
    if ((lane & 3) == 0 && lane < 16) st4(s_red + wave * 16 + (lane >> 2) * 4, v);
  } else {
    fold(4); fold(8); fold(16); fold(32);
    if (lane < 4) st4(s_red + wave * 16 + lane * 4, v);
  }
}
__device__ __forceinline__ void store_partials(const float* s_red, float* dst /* 16 floats */) {   // after a barrier
  if (threadIdx.x < 16) dst[threadIdx.x] = (s_red[threadIdx.x] + s_red[16 + threadIdx.x]) + (s_red[32 + threadIdx.x] + s_red[48 + threadIdx.x]);
}

// ---- bilinear sample of a [HL][WL][16] tensor at output pixel (oy, ox), channels c0..c0+3 (TFLite reference association) ---------
struct UpCoef { int y0, y1, x0, x1; float dy, dx; };
__device__ __forceinline__ void up_axis(int o, float scale, bool half_pixel, int in_size, int* lo, int* hi, float* frac) {
  const float v = half_pixel ? __fadd_rn(__fmul_rn((float)o + 0.5f, scale), -0.5f) : __fmul_rn((float)o, scale);
  const float fl = floorf(v);
  *lo = max((int)fl, 0);
  *hi = min((int)ceilf(v), in_size - 1);
  *frac = v - (float)*lo;
}
__device__ __forceinline__ float up_lerp(float x00, float x10, float x01, float x11, float dy, float dx) {
  const float a = __fmul_rn(__fmul_rn(x00, 1.f - dy), 1.f - dx), b = __fmul_rn(__fmul_rn(x10, dy), 1.f - dx);
  const float c = __fmul_rn(__fmul_rn(x01, 1.f - dy), dx), d = __fmul_rn(__fmul_rn(x11, dy), dx);
  return __fadd_rn(__fadd_rn(__fadd_rn(a, b), c), d);
}
__device__ __forceinline__ float4 up_sample(const float* __restrict__ lo, int WL, const UpCoef& u, int ch) {
  const float4 a = ld4(lo + (u.y0 * WL + u.x0) * 16 + ch), b = ld4(lo + (u.y1 * WL + u.x0) * 16 + ch);
  const float4 c = ld4(lo + (u.y0 * WL + u.x1) * 16 + ch), d = ld4(lo + (u.y1 * WL + u.x1) * 16 + ch);
  return make_float4(up_lerp(a.x, b.x, c.x, d.x, u.dy, u.dx), up_lerp(a.y, b.y, c.y, d.y, u.dy, u.dx), up_lerp(a.z, b.z, c.z, d.z, u.dy, u.dx),
                     up_lerp(a.w, b.w, c.w, d.w, u.dy, u.dx));
}
__device__ __forceinline__ float up_scale(int in, int out, bool align) { return (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out; }

// sum over the four lanes of a quad of p[quad]: a 4x4 "reduce-scatter" in 3 DPP exchanges (cf. quad_transpose)
__device__ __forceinline__ float quad_reduce_scatter(float p0, float p1, float p2, float p3, int quad) {
  const bool b0 = quad & 1, b1 = quad & 2;
  const float klo = (b0 ? p1 : p0) + dpp_quad(b0 ? p0 : p1, 1);     // positions {b0, 2 + b0} stay on this lane
  const float khi = (b0 ? p3 : p2) + dpp_quad(b0 ? p2 : p3, 1);
  return (b1 ? khi : klo) + dpp_quad(b1 ? klo : khi, 2);
}

// depthwise 3x3 (stride S) at tile pixel (py, px) of a dense [rows][ZW][16] LDS tile, channel quad `q`
template <int S>
__device__ __forceinline__ float4 dw3x3(const float* __restrict__ zt, int ZW, int py, int px, int q, const float4 (&wd)[9]) {
  float4 acc = f4zero();
#pragma unroll
  for (int fy = 0; fy < 3; fy++)
#pragma unroll
    for (int fx = 0; fx < 3; fx++) acc = f4fma(ld4(zt + ((S * py + fy) * ZW + S * px + fx) * 16 + 4 * q), wd[fy * 3 + fx], acc);
  return acc;
}

// decode of one model-resolution pixel (lib/libbackscrub.cc:333-357) — same arithmetic as decode_k / decode_meet4_k (kernels_img.hip)
__device__ __forceinline__ uint32_t seg_meet_val(float l0, float l1) {
  const float d = l1 - l0;
  if (fabsf(l0) <= 80.f && fabsf(l1) <= 80.f && fabsf(d) >= 1e-4f) return d > 0.f ? 0u : 255u;
  const float e0 = (float)exp((double)l0), e1 = (float)exp((double)l1);
  const float s = __fadd_rn(e0, e1);
  return __fdiv_rn(e0, s) < __fdiv_rn(e1, s) ? 0u : 255u;
}

// ==================================================================================================================================
// head: stem conv3x3/s2 (3 → 16) → 1x1 (16 → 16) → depthwise 3x3/s2; writes A (skip of the last decoder level), b0, and the
// pooled partial sums of both.  Tile = TR x TC pixels of b0.
// ==================================================================================================================================
__global__ __launch_bounds__(kSegThreads) void seg_head_k(const SegHead d, float* __restrict__ arena, long per_frame, const float* __restrict__ net_in,
                                                          const float* __restrict__ w) {
  const int f = blockIdx.y, ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC;
  const int AR = 2 * d.TR + 1, AC = 2 * d.TC + 1, ar0 = 2 * r0 - d.dw_pt, ac0 = 2 * c0 - d.dw_pl;
  const int IR = 2 * AR + 1, IC = 2 * AC + 1, ir0 = 2 * ar0 - d.stem_pt, ic0 = 2 * ac0 - d.stem_pl;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* in_t = seg_smem + kScrFloats;                              // [IR][IC][3]; later x_t = act(pw(A)) [AR*AC][16]
  const int r1 = max(IR * IC * 3, AR * AC * 16);
  float* a_t = in_t + ((r1 + 3) & ~3);                              // [AR*AC][16]
  float* x_t = in_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q = li & 3, cq4 = li & ~3;

  // 1. input tile (zero outside the image: SAME padding of the stem); a lane's loads are all issued before its first LDS store
  {
    const float* src = net_in + (size_t)f * (size_t)(d.H0 * d.W0 * 3);
    const int rowf = IC * 3, lo_rem = max(0, -ic0) * 3, hi_rem = min(IC, d.W0 - ic0) * 3;
    const unsigned mrow = d.m_rowf;
    constexpr int kB = 12;
    for (int base = 0; base < IR * rowf; base += kB * kSegThreads) {
      float v[kB];
#pragma unroll
      for (int j = 0; j < kB; j++) {
        const int i = base + j * kSegThreads + tid;
        const int row = div_magic(i, mrow), rem = i - row * rowf, gy = ir0 + row;
        v[j] = 0.f;
        if (i < IR * rowf && gy >= 0 && gy < d.H0 && rem >= lo_rem && rem < hi_rem) v[j] = src[((long)gy * d.W0 + ic0) * 3 + rem];
      }
#pragma unroll
      for (int j = 0; j < kB; j++) { const int i = base + j * kSegThreads + tid; if (i < IR * rowf) in_t[i] = v[j]; }
    }
  }
  // stem operand tables of this lane: k = 4s + g over the im2col axis (fy, fx, ci), 27 valid entries
  int koff[7];
  float ws[7];
#pragma unroll
  for (int s = 0; s < 7; s++) {
    const int k = 4 * s + g;
    const bool valid = k < 27;
    const int fy = k / 9, r9 = k - 9 * fy, fx = r9 / 3, ci = r9 - 3 * fx;
    koff[s] = valid ? (fy * IC + fx) * 3 + ci : 0;
    ws[s] = valid ? w[d.stem.w_off + (long long)k * d.stem.cout_pad + li] : 0.f;
  }
  const float4 bias_s = ld4(w + d.stem.b_off + cq4);
  __syncthreads();

  // 2. stem on the A region
  const int npix = AR * AC, ntile = (npix + 15) >> 4;
  const unsigned mac = d.m_ac;
  float4 sumA = f4zero();
  for (int t = wave; t < ntile; t += 4) {
    const int p = min(16 * t + li, npix - 1);
    const int ay = div_magic(p, mac), ax = p - ay * AC;
    const float* base = in_t + ((2 * ay) * IC + 2 * ax) * 3;
    float av[7];
#pragma unroll
    for (int s = 0; s < 7; s++) av[s] = base[koff[s]];
    f4acc acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ws[s], acc, 0, 0, 0);
    const int pp = 16 * t + 4 * g + q;
    float4 v = quad_transpose(acc, q);
    if (pp < npix) {
      const int y2 = div_magic(pp, mac), x2 = pp - y2 * AC, gy = ar0 + y2, gx = ac0 + x2;
      v = sg_act4(f4add(v, bias_s), d.stem.act);
      st4(a_t + pp * 16 + cq4, v);
      const bool inside = gy >= 0 && gy < d.H1 && gx >= 0 && gx < d.W1;
      if (inside && gy >= 2 * r0 && gy < 2 * r0 + 2 * d.TR && gx >= 2 * c0 && gx < 2 * c0 + 2 * d.TC) {   // each A pixel is stored by exactly one tile
        st4(fa + d.a_off + ((long)gy * d.W1 + gx) * 16 + cq4, v);
        sumA = f4add(sumA, v);
      }
    }
  }
  float wr[4];
  load_wtile(wr, w, d.pw, 0, li, g);
  const float4 bias_p = ld4(w + d.pw.b_off + cq4);
  __syncthreads();

  // 3. x = act(pw(A)) on the same region; zero outside the image (SAME padding of the depthwise)
  for (int t = wave; t < ntile; t += 4) {
    const int p = min(16 * t + li, npix - 1);
    const f4acc acc = mma16(ld4(a_t + p * 16 + 4 * g), wr);
    const int pp = 16 * t + 4 * g + q;
    float4 v = quad_transpose(acc, q);
    if (pp < npix) {
      const int y2 = div_magic(pp, mac), x2 = pp - y2 * AC, gy = ar0 + y2, gx = ac0 + x2;
      const bool inside = gy >= 0 && gy < d.H1 && gx >= 0 && gx < d.W1;
      v = inside ? sg_act4(f4add(v, bias_p), d.pw.act) : f4zero();
      st4(x_t + pp * 16 + cq4, v);
    }
  }
  const int quad = tid & 3;
  float4 wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ld4(w + d.dw.w_off + k * 16 + 4 * quad);
  const float4 bias_d = ld4(w + d.dw.b_off + 4 * quad);
  __syncthreads();

  // 4. depthwise 3x3 / stride 2 → b0
  float4 sumB = f4zero();
  const unsigned mtc = d.m_tc;
  for (int i = tid; i < d.TR * d.TC * 4; i += kSegThreads) {
    const int pix = i >> 2, py = div_magic(pix, mtc), px = pix - py * d.TC;
    if (r0 + py >= d.H2 || c0 + px >= d.W2) continue;
    const float4 v = sg_act4(f4add(dw3x3<2>(x_t, AC, py, px, quad, wd), bias_d), d.dw.act);
    st4(fa + d.b0_off + ((long)(r0 + py) * d.W2 + c0 + px) * 16 + 4 * quad, v);
    sumB = f4add(sumB, v);
  }
  float* s_red = seg_smem + kScrRed;
  wave_reduce16<0>(sumA, s_red);
  wave_reduce16<1>(sumB, s_red + 64);
  __syncthreads();
  store_partials(s_red, fa + d.part_a_off + (long)blockIdx.x * 16);
  store_partials(s_red + 64, fa + d.part_b0_off + (long)blockIdx.x * 16);
}

// ==================================================================================================================================
// k2: s = gate(GAP(b0)); B = pw_a(b0 * s) (skip of decoder level 2); x = act(pw_b(B)); c0 = act(dw3x3/s2(x)).  Tile = TR x TC of c0.
// The expanded tensor x (72 channels) exists only 16 channels at a time, in LDS.
// ==================================================================================================================================
__global__ __launch_bounds__(kSegThreads) void seg_k2_k(const SegK2 d, float* __restrict__ arena, long per_frame, const float* __restrict__ w) {
  const int f = blockIdx.y, ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC;
  const int BR = 2 * d.TR + 1, BC = 2 * d.TC + 1, br0 = 2 * r0 - d.dw_pt, bc0 = 2 * c0 - d.dw_pl;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  const int npix = BR * BC, ntile = (npix + 15) >> 4;
  float* B_t = seg_smem + kScrFloats;
  float* x_t = B_t + npix * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q = li & 3, cq4 = li & ~3;
  seg_gate(d.gate, fa, w, seg_smem, B_t);
  const float4 sv = ld4(seg_smem + kScrGate + 4 * g);
  float wr[4];
  load_wtile(wr, w, d.pw_a, 0, li, g);
  const float4 bias_a = ld4(w + d.pw_a.b_off + cq4);
  const unsigned mbc = d.m_bc;

  // 1. B on the region the depthwise needs (the next tile's operand is requested before this tile's MFMAs)
  float4 sumB = f4zero();
  auto load_b0 = [&](int t) {
    float4 a = f4zero();
    if (t < ntile) {
      const int p = min(16 * t + li, npix - 1);
      const int by = div_magic(p, mbc), bx = p - by * BC, gy = br0 + by, gx = bc0 + bx;
      if (gy >= 0 && gy < d.H2 && gx >= 0 && gx < d.W2) a = ld4(fa + d.b0_off + ((long)gy * d.W2 + gx) * 16 + 4 * g);
    }
    return a;
  };
  float4 a_next = load_b0(wave);
  for (int t = wave; t < ntile; t += 4) {
    float4 a = a_next;
    a_next = load_b0(t + 4);
    a = make_float4(__fmul_rn(a.x, sv.x), __fmul_rn(a.y, sv.y), __fmul_rn(a.z, sv.z), __fmul_rn(a.w, sv.w));
    const f4acc acc = mma16(a, wr);
    const int pp = 16 * t + 4 * g + q;
    float4 v = quad_transpose(acc, q);
    if (pp < npix) {
      const int y2 = div_magic(pp, mbc), x2 = pp - y2 * BC, hy = br0 + y2, hx = bc0 + x2;
      v = sg_act4(f4add(v, bias_a), d.pw_a.act);
      st4(B_t + pp * 16 + cq4, v);
      const bool inside = hy >= 0 && hy < d.H2 && hx >= 0 && hx < d.W2;
      if (inside && hy >= 2 * r0 && hy < 2 * r0 + 2 * d.TR && hx >= 2 * c0 && hx < 2 * c0 + 2 * d.TC) {
        st4(fa + d.B_off + ((long)hy * d.W2 + hx) * 16 + cq4, v);
        sumB = f4add(sumB, v);
      }
    }
  }
  float* s_red = seg_smem + kScrRed;
  wave_reduce16<0>(sumB, s_red);
  __syncthreads();
  store_partials(s_red, fa + d.part_B_off + (long)blockIdx.x * 16);

  // 2. 16 expanded channels at a time: x = act(pw_b(B)) → depthwise 3x3/s2 → c0
  const int C = d.dw.C, ngrp = (C + 15) >> 4, quad = tid & 3;
  const unsigned mtc = d.m_tc;
  for (int grp = 0; grp < ngrp; grp++) {
    load_wtile(wr, w, d.pw_b, 16 * grp, li, g);
    const float4 bias_b = ld4(w + d.pw_b.b_off + 16 * grp + cq4);
    for (int t = wave; t < ntile; t += 4) {
      const int p = min(16 * t + li, npix - 1);
      const f4acc acc = mma16(ld4(B_t + p * 16 + 4 * g), wr);
      const int pp = 16 * t + 4 * g + q;
      float4 v = quad_transpose(acc, q);
      if (pp < npix) {
        const int y2 = div_magic(pp, mbc), x2 = pp - y2 * BC, hy = br0 + y2, hx = bc0 + x2;
        const bool inside = hy >= 0 && hy < d.H2 && hx >= 0 && hx < d.W2;
        v = inside ? sg_act4(f4add(v, bias_b), d.pw_b.act) : f4zero();
        st4(x_t + pp * 16 + cq4, v);
      }
    }
    const int ch = 16 * grp + 4 * quad;
    float4 wd[9];
    float4 bias_d = f4zero();
    if (ch < C) {
#pragma unroll
      for (int k = 0; k < 9; k++) wd[k] = ld4(w + d.dw.w_off + (long long)k * C + ch);
      bias_d = ld4(w + d.dw.b_off + ch);
    }
    __syncthreads();
    if (ch < C)
      for (int i = tid; i < d.TR * d.TC * 4; i += kSegThreads) {
        const int pix = i >> 2, py = div_magic(pix, mtc), px = pix - py * d.TC;
        if (r0 + py >= d.H3 || c0 + px >= d.W3) continue;
        const float4 v = sg_act4(f4add(dw3x3<2>(x_t, BC, py, px, quad, wd), bias_d), d.dw.act);
        st4(fa + d.c0_off + ((long)(r0 + py) * d.W3 + c0 + px) * C + ch, v);
      }
    __syncthreads();
  }
}

// z = act(pw(skip * g + up(lo))) on the (TR+2) x (TC+2) halo region of a tile, zero outside the image — shared by k3 and the tail.
// A tile's five 16-byte operands (skip + four interpolation taps) are requested one tile AHEAD of the arithmetic that uses them.
struct GatedOperand { float4 s, a, b, c, d; float dy, dx; bool in; };
__device__ __forceinline__ void seg_gated_pw(const float* __restrict__ skip, const float* __restrict__ lo, int H, int W, int HL, int WL, bool half_pixel, bool align,
                                             const float* s_gate, const SegConvW& pw, const float* __restrict__ w, int r0, int c0, int ZH, int ZW, unsigned mzw,
                                             float* z_t) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q = li & 3, cq4 = li & ~3;
  float wr[4];
  load_wtile(wr, w, pw, 0, li, g);
  const float4 bias = ld4(w + pw.b_off + cq4);
  const float4 gv = ld4(s_gate + 4 * g);
  const float hs = up_scale(HL, H, align), wsc = up_scale(WL, W, align);
  const int npix = ZH * ZW, ntile = (npix + 15) >> 4;
  auto fetch = [&](int t) {
    GatedOperand o;
    o.in = false; o.dy = o.dx = 0.f;
    o.s = o.a = o.b = o.c = o.d = f4zero();
    if (t < ntile) {
      const int p = min(16 * t + li, npix - 1);
      const int zy = div_magic(p, mzw), zx = p - zy * ZW, iy = r0 - 1 + zy, ix = c0 - 1 + zx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        o.in = true;
        o.s = ld4(skip + ((long)iy * W + ix) * 16 + 4 * g);
        UpCoef u;
        up_axis(iy, hs, half_pixel, HL, &u.y0, &u.y1, &u.dy);
        up_axis(ix, wsc, half_pixel, WL, &u.x0, &u.x1, &u.dx);
        o.dy = u.dy; o.dx = u.dx;
        o.a = ld4(lo + (u.y0 * WL + u.x0) * 16 + 4 * g); o.b = ld4(lo + (u.y1 * WL + u.x0) * 16 + 4 * g);
        o.c = ld4(lo + (u.y0 * WL + u.x1) * 16 + 4 * g); o.d = ld4(lo + (u.y1 * WL + u.x1) * 16 + 4 * g);
      }
    }
    return o;
  };
  GatedOperand nx = fetch(wave);
  for (int t = wave; t < ntile; t += 4) {
    const GatedOperand o = nx;
    nx = fetch(t + 4);
    float4 a = f4zero();
    if (o.in) {
      const float4 up = make_float4(up_lerp(o.a.x, o.b.x, o.c.x, o.d.x, o.dy, o.dx), up_lerp(o.a.y, o.b.y, o.c.y, o.d.y, o.dy, o.dx),
                                    up_lerp(o.a.z, o.b.z, o.c.z, o.d.z, o.dy, o.dx), up_lerp(o.a.w, o.b.w, o.c.w, o.d.w, o.dy, o.dx));
      a = make_float4(__fadd_rn(__fmul_rn(o.s.x, gv.x), up.x), __fadd_rn(__fmul_rn(o.s.y, gv.y), up.y), __fadd_rn(__fmul_rn(o.s.z, gv.z), up.z),
                      __fadd_rn(__fmul_rn(o.s.w, gv.w), up.w));
    }
    const f4acc acc = mma16(a, wr);
    const int pp = 16 * t + 4 * g + q;
    float4 v = quad_transpose(acc, q);
    if (pp < npix) {
      const int y2 = div_magic(pp, mzw), x2 = pp - y2 * ZW, hy = r0 - 1 + y2, hx = c0 - 1 + x2;
      const bool inside = hy >= 0 && hy < H && hx >= 0 && hx < W;
      v = inside ? sg_act4(f4add(v, bias), pw.act) : f4zero();
      st4(z_t + pp * 16 + cq4, v);
    }
  }
}

// ==================================================================================================================================
// k3 (decoder level 2): z = act(pw1(B * g + up(lo2))); t = z + act(dw3x3(z)); lo = pw2(t).  Tile = TR x TC at the B resolution.
// ==================================================================================================================================
__global__ __launch_bounds__(kSegThreads) void seg_k3_k(const SegK3 d, float* __restrict__ arena, long per_frame, const float* __restrict__ w) {
  const int f = blockIdx.y, ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC, ZH = d.TR + 2, ZW = d.TC + 2;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* z_t = seg_smem + kScrFloats;
  float* t_t = z_t + ZH * ZW * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q = li & 3, cq4 = li & ~3;
  if (tid < 16) seg_smem[kScrGate + tid] = fa[d.g_off + tid];
  __syncthreads();
  seg_gated_pw(fa + d.skip_off, fa + d.lo2_off, d.H2, d.W2, d.HL, d.WL, d.half_pixel != 0, d.align_corners != 0, seg_smem + kScrGate, d.pw1, w, r0, c0, ZH, ZW, d.m_zw, z_t);
  const int quad = tid & 3;
  float4 wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ld4(w + d.dw.w_off + k * 16 + 4 * quad);
  const float4 bias_d = ld4(w + d.dw.b_off + 4 * quad);
  __syncthreads();
  const unsigned mtc = d.m_tc;
  for (int i = tid; i < d.TR * d.TC * 4; i += kSegThreads) {
    const int pix = i >> 2, py = div_magic(pix, mtc), px = pix - py * d.TC;
    const float4 zc = ld4(z_t + ((py + 1) * ZW + px + 1) * 16 + 4 * quad);
    const float4 dv = sg_act4(f4add(dw3x3<1>(z_t, ZW, py, px, quad, wd), bias_d), d.dw.act);
    st4(t_t + pix * 16 + 4 * quad, f4add(dv, zc));                 // dw epilogue: activation, then + residual z
  }
  float wr[4];
  load_wtile(wr, w, d.pw2, 0, li, g);
  const float4 bias2 = ld4(w + d.pw2.b_off + cq4);
  __syncthreads();
  const int npix = d.TR * d.TC, ntile = (npix + 15) >> 4;
  float4 sum = f4zero();
  for (int t = wave; t < ntile; t += 4) {
    const int p = min(16 * t + li, npix - 1);
    const f4acc acc = mma16(ld4(t_t + p * 16 + 4 * g), wr);
    const int pp = 16 * t + 4 * g + q;
    float4 v = quad_transpose(acc, q);
    if (pp < npix) {
      const int py = div_magic(pp, mtc), px = pp - py * d.TC;
      if (r0 + py < d.H2 && c0 + px < d.W2) {
        v = sg_act4(f4add(v, bias2), d.pw2.act);
        st4(fa + d.lo_off + ((long)(r0 + py) * d.W2 + c0 + px) * 16 + cq4, v);
        sum = f4add(sum, v);
      }
    }
  }
  float* s_red = seg_smem + kScrRed;
  wave_reduce16<0>(sum, s_red);
  __syncthreads();
  store_partials(s_red, fa + d.part_lo_off + (long)blockIdx.x * 16);
}

// ==================================================================================================================================
// tail (decoder level 1 + output): g = gate(GAP(A), GAP(lo)); z = act(pw(A * g + up(lo))); t = z + act(dw3x3(z));
// out = act3(Convolution2DTransposeBias 2x2 (t)) → logits (LOGITS) or straight into decode + temporal IIR on `ofinal`.
// Tile = TR x TC at the A resolution = 2TR x 2TC output pixels.  Phase B lanes = (pixel, channel quad): the depthwise runs on the
// lane's 4 channels, the 4 x CO transpose-conv dot products are split over the quad and reduce-scattered so that lane `quad`
// ends up with output position (fy, fx) = (quad >> 1, quad & 1) of its pixel.
// ==================================================================================================================================
template <int CO, bool LOGITS>
__global__ __launch_bounds__(kSegThreads) void seg_tail_k(const SegTail d, float* __restrict__ arena, long per_frame, float* __restrict__ net_out,
                                                          uint8_t* __restrict__ ofinal, const float* __restrict__ w) {
  const int f = blockIdx.y, ty = blockIdx.x / d.tiles_x, tx = blockIdx.x - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC, ZH = d.TR + 2, ZW = d.TC + 2;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* z_t = seg_smem + kScrFloats;
  seg_gate(d.gate, fa, w, seg_smem, z_t);
  seg_gated_pw(fa + d.skip_off, fa + d.lo_off, d.H1, d.W1, d.HL, d.WL, d.half_pixel != 0, d.align_corners != 0, seg_smem + kScrGate, d.pw, w, r0, c0, ZH, ZW, d.m_zw, z_t);
  const int tid = threadIdx.x, quad = tid & 3;
  float4 wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ld4(w + d.dw.w_off + k * 16 + 4 * quad);
  const float4 bias_d = ld4(w + d.dw.b_off + 4 * quad);
  float4 wt[4][CO];
  float bt[CO];
#pragma unroll
  for (int pos = 0; pos < 4; pos++)
#pragma unroll
    for (int oc = 0; oc < CO; oc++) wt[pos][oc] = ld4(w + d.tc_w_off + (long long)(pos * CO + oc) * 16 + 4 * quad);
#pragma unroll
  for (int oc = 0; oc < CO; oc++) bt[oc] = w[d.tc_b_off + oc];
  __syncthreads();
  const unsigned mtc = d.m_tc;
  const int fy = quad >> 1, fx = quad & 1;
  for (int i = tid; i < d.TR * d.TC * 4; i += kSegThreads) {
    const int pix = i >> 2, py = div_magic(pix, mtc), px = pix - py * d.TC;
    const int iy = r0 + py, ix = c0 + px;
    if (iy >= d.H1 || ix >= d.W1) continue;                         // uniform inside a quad (all four lanes share the pixel)
    const float4 zc = ld4(z_t + ((py + 1) * ZW + px + 1) * 16 + 4 * quad);
    const float4 t4 = f4add(sg_act4(f4add(dw3x3<1>(z_t, ZW, py, px, quad, wd), bias_d), d.dw.act), zc);
    float o[CO];
#pragma unroll
    for (int oc = 0; oc < CO; oc++) {
      float pp[4];
#pragma unroll
      for (int pos = 0; pos < 4; pos++) {
        const float4 wv = wt[pos][oc];
        pp[pos] = fmaf(t4.w, wv.w, fmaf(t4.z, wv.z, fmaf(t4.y, wv.y, t4.x * wv.x)));
      }
      o[oc] = sg_act(bt[oc] + quad_reduce_scatter(pp[0], pp[1], pp[2], pp[3], quad), d.act3);
    }
    const int oy = 2 * iy + fy, ox = 2 * ix + fx;
    const long opix = (long)f * d.H0 * d.W0 + (long)oy * d.W0 + ox;
    if (LOGITS) {
#pragma unroll
      for (int oc = 0; oc < CO; oc++) net_out[opix * CO + oc] = o[oc];
    } else {
      uint32_t val;
      if (CO == 2) val = seg_meet_val(o[0], o[CO - 1]);
      else val = ((double)o[0] > 0.65) ? 0u : 255u;                // MLKit: float promoted to double against the double literal (libbackscrub.cc:338)
      ofinal[opix] = (uint8_t)((val & 0xE0u) | (ofinal[opix] >> 3));
    }
  }
}

template <class K>
hipError_t allow_lds(K kernel, int lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

}  // namespace

hipError_t seg_prepare() {
  const int full = 160 * 1024;     // process-global kernel attributes: always the full LDS (cf. frame_program_prepare)
  hipError_t e;
  if ((e = allow_lds(seg_head_k, full)) != hipSuccess) return e;
  if ((e = allow_lds(seg_k2_k, full)) != hipSuccess) return e;
  if ((e = allow_lds(seg_k3_k, full)) != hipSuccess) return e;
  if ((e = allow_lds(seg_tail_k<1, false>, full)) != hipSuccess) return e;
  if ((e = allow_lds(seg_tail_k<1, true>, full)) != hipSuccess) return e;
  if ((e = allow_lds(seg_tail_k<2, false>, full)) != hipSuccess) return e;
  return allow_lds(seg_tail_k<2, true>, full);
}

hipError_t launch_seg_head(const SegHead& d, float* arena, long per_frame, const float* net_in, const float* weights, int n, hipStream_t s) {
  seg_head_k<<<dim3(d.tiles_y * d.tiles_x, n), kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, net_in, weights);
  return hipGetLastError();
}
hipError_t launch_seg_k2(const SegK2& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s) {
  seg_k2_k<<<dim3(d.tiles_y * d.tiles_x, n), kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights);
  return hipGetLastError();
}
hipError_t launch_seg_k3(const SegK3& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s) {
  seg_k3_k<<<dim3(d.tiles_y * d.tiles_x, n), kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights);
  return hipGetLastError();
}
hipError_t launch_seg_tail(const SegTail& d, float* arena, long per_frame, float* net_out, uint8_t* ofinal, const float* weights, bool logits, int n, hipStream_t s) {
  const dim3 grid(d.tiles_y * d.tiles_x, n);
  const size_t lds = (size_t)d.lds_floats * sizeof(float);
  if (d.Co == 2) {
    if (logits) seg_tail_k<2, true><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights);
    else seg_tail_k<2, false><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights);
  } else if (d.Co == 1) {
    if (logits) seg_tail_k<1, true><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights);
    else seg_tail_k<1, false><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace bsx
