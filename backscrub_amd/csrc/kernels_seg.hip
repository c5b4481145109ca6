// kernels_seg.hip — the spatially-parallel segment kernels of the Meet / MLKit networks (segments.hpp).
//
// Every kernel: grid = (tiles per frame, frames), 256 lanes = 4 waves per workgroup, several workgroups per CU.  A workgroup
// owns one tile of its segment's OUTPUT pixels and recomputes the halo its depthwise 3x3 needs; intermediate tensors of the
// tile live in LDS as [pixel][16] rows whose four channel quads are permuted per column (swz_a / swz_b / swz_l below: conflict-free
// for every access pattern used here — MFMA epilogue stores, MFMA operand reads, stride-1 and stride-2 depthwise taps), never in HBM.
//
// 1x1 convolutions and the 3x3 stem run on v_mfma_f32_16x16x4_f32 — exact f32 FMA chains, so the results stay within
// float rounding of the reference order (measured against the oracle: < 1e-5 relative on the logits).  Operand maps as
// in kernels_frame.hip, operands swapped: A lane (li, g) = w[k][channel li]; B lane = x[pixel li][k = 4g..4g+3]; D lane = channels 4g..4g+3 of pixel li —
// 4 consecutive channels of ONE pixel (acc_quad): 16-byte stores everywhere.
//
// Reference operators covered (file:line into /root/reference): Interpreter::Invoke() lib/libbackscrub.cc:307 — CONV_2D,
// DEPTHWISE_CONV_2D, RESIZE_BILINEAR (half-pixel), MUL/ADD gates, AVERAGE_POOL_2D (as partial sums), FULLY_CONNECTED /
// 1x1 gate convs; Convolution2DTransposeBias lib/transpose_conv_bias.cc:37-114; decode + IIR lib/libbackscrub.cc:317-357.
// TWO ways into this file (round 6):
//   * hipcc, ahead of time: every kernel a template instantiation that takes its segment descriptor as a kernel argument — any graph of the family, also what runs when
//     hipRTC is unavailable and what the stage tests' logits variant uses;
//   * hipRTC, when a context is created (gen_seg.cpp: this text + segments.hpp + mfma_tile.hpp flattened by build.py into seg_rtc_src.inc, with BSX_SEG_RTC, the
//     template arguments and the four descriptors of THE LOADED GRAPH as compile-time constants in front): extern "C" kernels bsx_seg_head / _k2 / _k3 / _tail in which
//     every geometry value, weight offset and activation kind is a constant — loops get their trip counts, index arithmetic its strength reduction, dead activation
//     branches disappear.  tools/seg_probe.sh measured that form first (profiles/r06o: head 39.7 -> 36.1 us, k2 40.2 -> 34.6, tail 44.7 -> 41.6, step -3.7 % at configs[1]).
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#ifndef BSX_SEG_RTC               // (the flattened translation unit of gen_seg.cpp carries segments.hpp and mfma_tile.hpp in front of this text and needs nothing else)
#include "debug_switches.hpp"

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"
#include "mfma_tile.hpp"
#include "segments.hpp"
#endif

// Where the kernels get their descriptor `d` from: the kernel argument (ahead-of-time build), the generator's constants (BSX_SEG_RTC: kSegHEAD ... in front of this text),
// or — tools/seg_probe.sh, the ahead-of-time experiment that preceded the hipRTC form — a dumped file of constants (-DBSX_SEG_PROBE=\"file\": one model only, never shipped).
#if defined(BSX_SEG_RTC)
#define BSX_SEG_D(T, NAME) constexpr T d = kSeg##NAME; (void)d_rt
#elif defined(BSX_SEG_PROBE)
#define BSX_SEG_D(T, NAME) constexpr T d = kProbe##NAME; (void)d_rt
#else
#define BSX_SEG_D(T, NAME) const T& d = d_rt
#endif

namespace bsx {
#ifdef BSX_SEG_RTC
namespace segrtc {               // (named: an extern "C" kernel inside an unnamed namespace would have internal linkage)
BSX_SEG_CONSTANTS                // constexpr SegHead kSegHEAD = ...; kSegK2; kSegK3; kSegTAIL  (gen_seg.cpp)
// the template arguments of the one variant this graph / mode needs
constexpr bool STEM_HSWISH = BSX_SEG_HS != 0, H16 = BSX_SEG_H16 != 0, U8IN = BSX_SEG_U8 != 0, LOGITS = false, SIGMOID = BSX_SEG_SIG != 0;
constexpr int CO = BSX_SEG_CO;
#else
namespace {
#endif
#ifdef BSX_SEG_PROBE
#include BSX_SEG_PROBE
#endif

extern __shared__ __attribute__((aligned(16))) float seg_smem[];

constexpr int kScrGate = 0;      // [0,16) gate vector, [16,48) means, [48,80) hidden
constexpr int kScrRed = 96;      // [96, 96 + 2*256) partial-sum meeting points (two sets of 16 slots x 16 channels)
constexpr int kScrFloats = kSegScratchFloats;  // tiles start here

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Activation tensors that cross a kernel boundary (A, b0, B, c0, lo2, lo — SURVEY Appendix B.1) are f32 by default; H16 = the 16-bit activation
// STORAGE mode (BSX_ACT16=1, what SetAllowFp16PrecisionForFp32 permits: /root/reference/lib/libbackscrub.cc:225): the same tensor at the same
// arena offset as packed halves (round to nearest even on store), all arithmetic still f32.  `idx` = element index inside the tensor.
typedef _Float16 h4s __attribute__((ext_vector_type(4)));
template <bool H16> __device__ __forceinline__ float4 ldg4(const float* base, unsigned idx) {
  if (H16) { const h4s h = *reinterpret_cast<const h4s*>(reinterpret_cast<const _Float16*>(base) + idx); return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }
  return *reinterpret_cast<const float4*>(base + idx);
}
template <bool H16> __device__ __forceinline__ void stg4(float* base, unsigned idx, float4 v) {
  if (H16) *reinterpret_cast<h4s*>(reinterpret_cast<_Float16*>(base) + idx) = h4s{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  else *reinterpret_cast<float4*>(base + idx) = v;
}
// ---- LDS tile layouts (bank model: MI355X_MICROARCH.md §LDS; evaluated for every access pattern of this file by tests/test_lds_layouts.py) ----------------------------
// A tile pixel is 16 floats = four 16-byte channel quads.  Stored densely ([pixel][16], quad q at +4q) the MFMA epilogue's ds_write_b128 — 8 consecutive lanes = 8
// consecutive pixels, ONE quad — touches two of the eight 16-byte slots of its 128-byte window (4-way conflict: 32 LDS cycles for an instruction that costs 13), the MFMA
// operand read (ds_read_b128, 16-lane groups of 8 pixels x 2 quads) is 2-way, and so is every tap of a STRIDE-2 depthwise (even columns only: half of the banks).
// SQ counters of round 4: 55 % (k2), 41 % (head), 40 % (tail) of the LDS-active cycles were conflict cycles.  Three layouts remove all of them:
//   swz_a — stride-1 tiles (k2's B, k3 / tail's z and t): quad q of column x lives at quad position q ^ swz_a(x).  The 8 pixels of an epilogue store then cover all 8
//           slots, the 16 (pixel, quad) pairs of an operand read all 16; the stride-1 depthwise (4 quads of one pixel per lane quad) is conflict-free under any permutation.
//   swz_b — tiles a STRIDE-2 depthwise reads (head's x, k2's x): columns de-interleaved — a row is [even columns | odd columns] — so a tap is a stride-1 read of one plane,
//           and quad position q ^ swz_b(x) (bit 1 = the plane) keeps the epilogue store (lanes alternate planes) on 8 distinct slots.
//   swz_l — the staged low-resolution window (k3 / tail): stride 16 floats instead of 20 (the padding halved nothing: every tap read was 2-way) with a period-16 swizzle under
//           which the 2x up-sampling taps (two neighbouring lanes share a source pixel) are conflict-free for every window alignment: 20 % less LDS for the window as well.
// Written as single-expression functions so that the test evaluates the kernels' own source.
__device__ __forceinline__ int swz_a(int x) { return (x >> 1) & 3; }
__device__ __forceinline__ int swz_b(int x) { return ((x >> 2) & 1) | ((x & 1) << 1); }
__device__ __forceinline__ int swz_l(int x) { return (((x >> 2) & 1) << 1) ^ ((x >> 3) & 1); }
// float offset of quad q of column x inside a tile row: stride-1 layout / de-interleaved layout (half = (row width + 1) / 2 columns in the even plane)
__device__ __forceinline__ int col_a(int x, int q) { return x * 16 + 4 * (q ^ swz_a(x)); }
__device__ __forceinline__ int col_b(int x, int q, int half) { return ((x & 1) * half + (x >> 1)) * 16 + 4 * (q ^ swz_b(x)); }
__device__ __forceinline__ int col_l(int x, int q) { return x * kSegLoStride + 4 * (q ^ swz_l(x)); }

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }

// Activations.  none / relu / relu6 are ONE branch-free v_med3_f32 with kernel-uniform bounds (the first version switched on the
// activation code per element: a dozen scalar branches per MFMA tile were most of its SALU stream and of its stalls); hard-swish
// (stems) and the logistic (MLKit's output) are compile-time variants of the kernels that need them.  Same arithmetic forms as the
// per-frame program (hardware exp2 / rcp).
struct Clamp { float lo, hi; };
__device__ __forceinline__ Clamp clamp_of(int act) {               // planner guarantees act in {none, relu, relu6} wherever this is used
  Clamp c;
  c.lo = act == kActNone ? -__builtin_huge_valf() : 0.f;
  c.hi = act == kActRelu6 ? 6.f : __builtin_huge_valf();
  return c;
}
__device__ __forceinline__ float4 clamp4(float4 v, Clamp c) {
  return make_float4(__builtin_amdgcn_fmed3f(v.x, c.lo, c.hi), __builtin_amdgcn_fmed3f(v.y, c.lo, c.hi), __builtin_amdgcn_fmed3f(v.z, c.lo, c.hi),
                     __builtin_amdgcn_fmed3f(v.w, c.lo, c.hi));
}
__device__ __forceinline__ float hswish1(float v) { return v * __builtin_amdgcn_fmed3f(v + 3.f, 0.f, 6.f) * 0.16666667163372040f; }
__device__ __forceinline__ float4 hswish4(float4 v) { return make_float4(hswish1(v.x), hswish1(v.y), hswish1(v.z), hswish1(v.w)); }
__device__ __forceinline__ float sigmoid1(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)); }
// gate prologue only (a handful of lanes, once per workgroup)
__device__ __forceinline__ float sg_act(float v, int act) {
  if (act == kActNone) return v;
  if (act == kActSigmoid) return sigmoid1(v);
  if (act == kActHswish) return hswish1(v);
  if (act == kActRelu) return fmaxf(v, 0.f);
  return __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
}

// The weights are the A operand of every MFMA here (rows = output channels) and the activations the B operand (columns = pixels): the accumulator of lane
// (li, g) is then D[channels 4g..4g+3][pixel li] — four consecutive channels of ONE pixel, the 16-byte unit every store, bias and residual in these kernels
// works in — without the 4x4 transpose inside lane quads (12 selects + 4 DPP moves per tile) the pixel-major operand order needed.  Same products, same
// accumulation order over k: bit-identical values.
__device__ __forceinline__ float4 acc_quad(const f4acc acc) { return make_float4(acc[0], acc[1], acc[2], acc[3]); }
// one 16-pixel x 16-channel tile of a 1x1 convolution with Cin = 16: 4 MFMAs
__device__ __forceinline__ f4acc mma16(const float4 a, const float (&wr)[4]) {
  f4acc acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], a.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1], a.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2], a.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3], a.w, acc, 0, 0, 0);
  return acc;
}
// B-operand registers of the [16][cout_pad] block at output channels n0..n0+15: lane (li, g) holds w[4g + r][n0 + li]
__device__ __forceinline__ void load_wtile(float (&wr)[4], const float* __restrict__ w, const SegConvW& c, int n0, int li, int g) {
#pragma unroll
  for (int r = 0; r < 4; r++) wr[r] = (w + c.w_off)[(unsigned)((4 * g + r) * c.cout_pad + n0 + li)];
}

// ---- gate prologue: means from partial sums → FC → [FC] → s_gate[0..C) ---------------------------------------------------
// All 256 lanes call it.  GAP(a | b) (concatenated) or GAP(a) + GAP(b) (MLKit's GAP(skip + up)).  ONE cooperative memory round
// trip stages the partial sums and both FC weight blocks in LDS (`stage`, <= kGateStageFloats floats, may alias a tile region that
// is not live yet); everything after runs from LDS: a per-workgroup serial chain of dependent global loads here was most of the
// first version's run time.
constexpr int kGateStageFloats = kSegGateStageFloats;
__device__ __forceinline__ void seg_gate(const SegGate& gt, const float* __restrict__ fa, const float* __restrict__ w, float* scr, float* stage) {
  float* s_gate = scr + kScrGate;
  if (gt.timing_skip) { if (threadIdx.x < 16) s_gate[threadIdx.x] = 0.5f; __syncthreads(); return; }      // upper bound of what hoisting the gate out could buy
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
      const float* src = fa + gt.part[k].off;
      float s = 0.f;
#pragma unroll 4
      for (int i = slice; i < gt.part[k].n; i += 16) s += src[(unsigned)(i * 16 + c)];
      ps[k * 256 + slice * 16 + c] = s;
    }
  }
  {
    const float* g1 = w + f1.w_off;
    const float* g2 = w + f2.w_off;
    for (int i = tid; i < w1n; i += kSegThreads) w1[i] = g1[(unsigned)i];
    for (int i = tid; i < w2n; i += kSegThreads) w2[i] = g2[(unsigned)i];
    if (tid < f1.Cout) b1[tid] = (w + f1.b_off)[(unsigned)tid];
    if (gt.n_fc == 2 && tid < f2.Cout) b2[tid] = (w + f2.b_off)[(unsigned)tid];
  }
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
  // FC layers: lane = (output, slice of the inputs) — 8 consecutive lanes share an output and meet through DPP.  (One lane per output walked
  // its whole weight row with a stride of Cin floats: every lane on the same two LDS banks, a 16-way conflict per step, 2 us per layer.)
  auto fc = [&](const SegFc& f, const float* x, const float* wl, const float* bl, float* y) {
    const int out = tid >> 3, ks = tid & 7, kper = f.Cin >> 3;               // Cin is 16 or 32 (checked by the planner)
    float acc = 0.f;
    if (out < f.Cout)
      for (int j = 0; j < kper; j++) acc = fmaf(x[ks * kper + j], wl[out * f.Cin + ks * kper + j], acc);
    acc += dpp_quad(acc, 1);
    acc += dpp_quad(acc, 2);
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x141, 0xf, 0xf, true));   // row_half_mirror: both quads of the 8
    if (out < f.Cout && ks == 0) y[out] = sg_act(acc + bl[out], f.act);
  };
  fc(f1, s_mean, w1, b1, gt.n_fc == 1 ? s_gate : s_hid);
  __syncthreads();
  if (gt.n_fc == 2) {
    fc(f2, s_hid, w2, b2, s_gate);
    __syncthreads();
  }
}

// ---- per-tile partial sums of a 16-channel tensor → partials[tile][16] -------------------------------------------------------
// Lanes hold float4 partial sums of their channel quad.  MODE 0: MFMA-epilogue lanes (lane (g, li): channel quad g, the 16 lanes of
// the row differ in pixel); MODE 1: depthwise lanes (lane = 4 * pixel + quad inside a 16-lane DPP row).  DPP row shifts leave
// the partial float4s of a wave; they meet in s_red[16 slots][16 channels] and the first 16 lanes of the workgroup finish the sum.
template <int SHR>
__device__ __forceinline__ float dpp_row_shr(float v) {        // lane i <- lane i - SHR inside its row of 16 (0 where there is none)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SHR, 0xf, 0xf, true));
}
template <int MODE>
__device__ __forceinline__ void wave_reduce16(float4 v, float* s_red, int wave, int lane) {
  if (MODE == 0) {                                                   // MFMA-epilogue lanes: row g of 16 lanes = 16 pixels of channel quad g
    v.x += dpp_row_shr<1>(v.x); v.y += dpp_row_shr<1>(v.y); v.z += dpp_row_shr<1>(v.z); v.w += dpp_row_shr<1>(v.w);
    v.x += dpp_row_shr<2>(v.x); v.y += dpp_row_shr<2>(v.y); v.z += dpp_row_shr<2>(v.z); v.w += dpp_row_shr<2>(v.w);
    v.x += dpp_row_shr<4>(v.x); v.y += dpp_row_shr<4>(v.y); v.z += dpp_row_shr<4>(v.z); v.w += dpp_row_shr<4>(v.w);
    v.x += dpp_row_shr<8>(v.x); v.y += dpp_row_shr<8>(v.y); v.z += dpp_row_shr<8>(v.z); v.w += dpp_row_shr<8>(v.w);
    const int li = lane & 15;                                        // lane 15 of the row holds its sum → slot 4 * wave; lanes 12..14 zero the wave's other three slots
    if (li >= 12) st4(s_red + (wave * 4 + (15 - li)) * 16 + 4 * (lane >> 4), li == 15 ? v : f4zero());
  } else {
    v.x += dpp_row_shr<4>(v.x); v.y += dpp_row_shr<4>(v.y); v.z += dpp_row_shr<4>(v.z); v.w += dpp_row_shr<4>(v.w);
    v.x += dpp_row_shr<8>(v.x); v.y += dpp_row_shr<8>(v.y); v.z += dpp_row_shr<8>(v.z); v.w += dpp_row_shr<8>(v.w);
    if ((lane & 12) == 12) st4(s_red + (wave * 4 + (lane >> 4)) * 16 + 4 * (lane & 3), v);     // slot (wave, row), channels 4 * quad
  }
}
__device__ __forceinline__ void store_partials(const float* s_red, float* dst /* 16 floats */) {   // after a barrier
  if (threadIdx.x < 16) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) s += s_red[j * 16 + threadIdx.x];
    dst[threadIdx.x] = s;
  }
}

// ---- bilinear up-sampling (RESIZE_BILINEAR) as per-axis tables: index pair + fraction, TFLite's clamping ----------------------
__device__ __forceinline__ void up_axis(int o, float scale, bool half_pixel, int in_size, int* lo, int* hi, float* frac) {
  const float v = half_pixel ? __fadd_rn(__fmul_rn((float)o + 0.5f, scale), -0.5f) : __fmul_rn((float)o, scale);
  const float fl = floorf(v);
  *lo = max((int)fl, 0);
  *hi = min((int)ceilf(v), in_size - 1);
  *frac = v - (float)*lo;
}
__device__ __forceinline__ float up_scale(int in, int out, bool align) { return (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out; }

// sum over the four lanes of a quad of p[quad]: a 4x4 "reduce-scatter" in 3 DPP exchanges (cf. quad_transpose)
__device__ __forceinline__ float quad_reduce_scatter(float p0, float p1, float p2, float p3, int quad) {
  const bool b0 = quad & 1, b1 = quad & 2;
  const float klo = (b0 ? p1 : p0) + dpp_quad(b0 ? p0 : p1, 1);     // positions {b0, 2 + b0} stay on this lane
  const float khi = (b0 ? p3 : p2) + dpp_quad(b0 ? p2 : p3, 1);
  return (b1 ? khi : klo) + dpp_quad(b1 ? klo : khi, 2);
}

// depthwise 3x3 (stride S) at pixel (py, px) of a [rows][RW][16] LDS tile, channel quad `q`; packed FMAs (v_pk_fma_f32)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v ldv(const float* p) { return *reinterpret_cast<const f4v*>(p); }
__device__ __forceinline__ f4v tov(float4 a) { f4v r = {a.x, a.y, a.z, a.w}; return r; }
__device__ __forceinline__ float4 tof4(f4v a) { return make_float4(a.x, a.y, a.z, a.w); }
// `co[fx]` = float offset of the lane's quad at column S * px + fx of a tile row (col_a / col_b: the layout's swizzle is folded into the three per-lane constants),
// `row0` = the tile row of tap fy = 0, RWF = floats per tile row
__device__ __forceinline__ f4v dw3x3(const float* __restrict__ zt, int RWF, int row0, const int (&co)[3], const f4v (&wd)[9]) {
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  const float* base = zt + row0 * RWF;
#pragma unroll
  for (int fy = 0; fy < 3; fy++)
#pragma unroll
    for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(ldv(base + fy * RWF + co[fx]), wd[fy * 3 + fx], acc);
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

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// MFMA phases walk ROW-ALIGNED tiles: tile t = (row t / ctiles, 16 columns 16 * (t % ctiles) ..) of a [rows][RW = 16 * ctiles][16]
// LDS tile.  The row is wave-uniform (scalar index math, scalar in-image tests), the column of a lane is 16 * ct + its fixed lane
// offset: no per-pixel divisions, which were most of the VALU work of the first (linear-index) version of these kernels.
struct RowTile { int row, ct; };
__device__ __forceinline__ RowTile row_tile(int t, int ctiles, unsigned m_ct) {
  RowTile r;
  r.row = (int)(((unsigned)t * m_ct) >> 16);           // t / ctiles for t < 4096 (m_ct = ceil(65536 / ctiles))
  r.ct = t - r.row * ctiles;
  return r;
}

// ==================================================================================================================================
// head: stem conv3x3/s2 (3 → 16) → 1x1 (16 → 16) → depthwise 3x3/s2; writes A (skip of the last decoder level), b0, and the
// pooled partial sums of both.  Tile = TR (<= 4) x TC (<= 15) pixels of b0.
// ==================================================================================================================================
// U8IN: the network input arrives as the filtered 8-bit pixels (R | G<<8 | B<<16 per pixel, prep_fused_k<2>) and is normalised here with the
// same two roundings convertTo applies (libbackscrub.cc:302): fadd(fmul(float(q), scale), offset) — bit-identical to reading the f32 tensor.
#ifdef BSX_SEG_RTC
extern "C" __global__ __launch_bounds__(kSegThreads) void bsx_seg_head(
#else
template <bool STEM_HSWISH, bool H16, bool U8IN>
__global__ __launch_bounds__(kSegThreads) void seg_head_k(
#endif
    const SegHead d_rt, float* __restrict__ arena, long per_frame, const float* __restrict__ net_in,
                                                          const float* __restrict__ w, float in_scale, float in_offset, int n_frames) {
  BSX_SEG_D(SegHead, HEAD);
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(d.tiles_y * d.tiles_x), (unsigned)n_frames, &f_, &t_);
  const int f = (int)f_, ty = (int)t_ / d.tiles_x, tx = (int)t_ - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC;
  const int AR = 2 * d.TR + 1, AC = 2 * d.TC + 1, ar0 = 2 * r0 - d.dw_pt, ac0 = 2 * c0 - d.dw_pl, RW = d.rw, ctiles = RW >> 4;
  const int IR = 2 * AR + 1, IC = 2 * AC + 1, ir0 = 2 * ar0 - d.stem_pt, ic0 = 2 * ac0 - d.stem_pl, rowf = IC * 3;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* in_t = seg_smem;                                           // [IR][IC * 3]; no scratch block in front: the head has no gate prologue, and its partial-sum meeting
                                                                    // points (s_red, 512 floats) reuse this window once the stem is done with it — 2.5 KB that put the 4 x 14 tile's
                                                                    // workgroup exactly ON the 32 KB line (5 per CU only if the allocator wastes nothing)
  float* a_t = in_t + max(512, (IR * rowf + 3) & ~3);                      // x = act(pw(stem)) [AR][AC][16]: rows of AC pixels, not of RW = 16 ceil(AC / 16) — the MFMA tiles
                                                                    // round the COMPUTE up to 16 pixels, the storage need not (5 workgroups per CU instead of 4 at a 4 x 13 tile)
  float* A_out = fa + d.a_off;                                      // uniform bases + 32-bit lane offsets (global_load/store saddr forms)
  float* b0_out = fa + d.b0_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4, cq4 = 4 * g;

  // 1. input tile (zero outside the image: SAME padding of the stem): wave = rows, lane = row elements; every load of the lane is
  //    in flight before its first LDS store
  if (d.dbg_skip & 8) { /* timing: no input tile */ } else
  if (U8IN) {                                                       // lane = pixel of the row (IC <= 63): one dword = three input values
    // (addresses clamped into the image and the load unconditional: under `if (inside)` every row became an exec-masked region with its own s_waitcnt vmcnt(0) and
    //  a dozen register copies to merge the paths — five serialised memory round trips where this has five loads in flight; outside pixels are zeroed below.
    //  Measured: seg_head 55.5 -> 51.5 us lite, 249 -> 226 mlkit / HD.  The same rewrite of the k2 / k3 / tail prefetches, whose loads were already in flight,
    //  measured 1-3 % slower — more loads issued — and was not kept: profiles/r03aj)
    const uint32_t* src = reinterpret_cast<const uint32_t*>(net_in) + (size_t)f * (size_t)(d.H0 * d.W0);
    const bool colok = lane < IC && ic0 + lane >= 0 && ic0 + lane < d.W0;
    const int gxc = min(max(ic0 + lane, 0), d.W0 - 1);
    uint32_t v[5];
    bool ok[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int row = wave + 4 * j, gy = ir0 + row;
      ok[j] = row < IR && gy >= 0 && gy < d.H0 && colok;
      v[j] = src[(unsigned)(min(max(gy, 0), d.H0 - 1) * d.W0 + gxc)];
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int row = wave + 4 * j;
      if (row < IR && lane < IC) {
        float* o = in_t + row * rowf + 3 * lane;
        o[0] = ok[j] ? __fadd_rn(__fmul_rn((float)(v[j] & 255u), in_scale), in_offset) : 0.f;
        o[1] = ok[j] ? __fadd_rn(__fmul_rn((float)((v[j] >> 8) & 255u), in_scale), in_offset) : 0.f;
        o[2] = ok[j] ? __fadd_rn(__fmul_rn((float)((v[j] >> 16) & 255u), in_scale), in_offset) : 0.f;
      }
    }
  } else {
    const float* src = net_in + (size_t)f * (size_t)(d.H0 * d.W0 * 3);
    const int lo_rem = max(0, -ic0) * 3, hi_rem = min(IC, d.W0 - ic0) * 3;
    float v[5][3];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int row = wave + 4 * j, gy = ir0 + row;
      const bool rowok = row < IR && gy >= 0 && gy < d.H0;
#pragma unroll
      for (int e3 = 0; e3 < 3; e3++) {
        const int e = lane + 64 * e3;
        v[j][e3] = 0.f;
        if (rowok && e >= lo_rem && e < hi_rem) v[j][e3] = src[(unsigned)((gy * d.W0 + ic0) * 3 + e)];
      }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int row = wave + 4 * j;
#pragma unroll
      for (int e3 = 0; e3 < 3; e3++) { const int e = lane + 64 * e3; if (row < IR && e < rowf) in_t[row * rowf + e] = v[j][e3]; }
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
    koff[s] = valid ? fy * rowf + fx * 3 + ci : 0;
    ws[s] = valid ? (w + d.stem.w_off)[(unsigned)(k * d.stem.cout_pad + li)] : 0.f;
  }
  const float4 bias_s = ld4(w + d.stem.b_off + cq4);
  const Clamp cl_stem = clamp_of(d.stem.act), cl_pw = clamp_of(d.pw.act), cl_dw = clamp_of(d.dw.act);
  const int ntile = AR * ctiles;
  const int xe = li;                                        // column of the pixel this lane owns after the quad transpose
  __syncthreads();

  // 2 + 3. stem on the A region and x = act(pw(A)) on the same pixels, two tiles per iteration (their MFMA chains interleave).  With the weights as the MFMA's A
  //    operand the stem's accumulator lane (li, g) holds channels 4 g .. 4 g + 3 of pixel li — EXACTLY the B operand the 1x1's four MFMAs want from that lane: the
  //    activated stem output goes from the accumulator registers straight into them (round 4).  Until then it made a round trip through an LDS tile of its own
  //    ([AR][RW][16]: 18 KB written, a barrier, 18 KB read) and the 1x1 was a second loop over the tiles; the same values enter the same instructions, so the results
  //    are the same bits.  x lands where that tile was (a_t's region: in_t is still being read by the other waves' stem tiles); zero outside the image = SAME padding
  //    of the depthwise.
  float* x_f = a_t;                                                 // DENSE [AR][AC][16] rows: the de-interleaved, swizzled form (col_b) removes this tile's bank conflicts as it does in
                                                                    // k2 — and measured 4-5 % SLOWER here on all three networks (profiles/r05b: 44.4 -> 46.5 us lite, 389 -> 409 us segm_full / HD):
                                                                    // the head is bound by VALU issue, and the odd row width AC makes the swizzled address arithmetic per tile, not per lane
  float wr[4];
  load_wtile(wr, w, d.pw, 0, li, g);
  const float4 bias_p = ld4(w + d.pw.b_off + cq4);
  float4 sumA = f4zero();
  auto stem_pw_epilogue = [&](const f4acc acc, const RowTile rt) {
    float4 v = acc_quad(acc);
    const int x2 = 16 * rt.ct + xe, gy = ar0 + rt.row, gx = ac0 + x2;
    v = f4add(v, bias_s);
    v = STEM_HSWISH ? hswish4(v) : clamp4(v, cl_stem);
    // (lanes past the tile's last column hold the stem of column AC - 1, as the clamped read of the two-loop form did; their results are dropped below)
    f4acc pa = {0.f, 0.f, 0.f, 0.f};
    if (!(d.dbg_skip & 2)) {
      pa = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], v.x, pa, 0, 0, 0);
      pa = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1], v.y, pa, 0, 0, 0);
      pa = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2], v.z, pa, 0, 0, 0);
      pa = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3], v.w, pa, 0, 0, 0);
    }
    if (x2 < AC) {
      const bool row_owned = gy >= max(2 * r0, 0) && gy < min(2 * r0 + 2 * d.TR, d.H1);          // scalar
      if (row_owned && gx >= 2 * c0 && gx < min(2 * c0 + 2 * d.TC, d.W1)) {                      // each A pixel is stored by exactly one tile
        stg4<H16>(A_out, (unsigned)((gy * d.W1 + gx) * 16 + cq4), v);
        sumA = f4add(sumA, v);
      }
      float4 xv = acc_quad(pa);
      const bool inside = gy >= 0 && gy < d.H1 && gx >= 0 && gx < d.W1;
      xv = inside ? clamp4(f4add(xv, bias_p), cl_pw) : f4zero();
      st4(x_f + (rt.row * AC + x2) * 16 + cq4, xv);
    }
  };
  for (int t = wave; t < ntile && !(d.dbg_skip & 1); t += 8) {
    const bool two = t + 4 < ntile;                                  // scalar
    const RowTile r0t = row_tile(t, ctiles, d.m_ct), r1t = row_tile(two ? t + 4 : t, ctiles, d.m_ct);
    const float* base0 = in_t + (2 * r0t.row) * rowf + 6 * min(16 * r0t.ct + li, AC - 1);
    const float* base1 = in_t + (2 * r1t.row) * rowf + 6 * min(16 * r1t.ct + li, AC - 1);
    float av0[7], av1[7];
#pragma unroll
    for (int s7 = 0; s7 < 7; s7++) { av0[s7] = base0[koff[s7]]; av1[s7] = base1[koff[s7]]; }
    f4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s7 = 0; s7 < 7; s7++) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[s7], av0[s7], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[s7], av1[s7], acc1, 0, 0, 0);
    }
    stem_pw_epilogue(acc0, r0t);
    if (two) stem_pw_epilogue(acc1, r1t);
  }
  const int quad = lane & 3, px = lane >> 2;                        // depthwise lanes: (pixel of the row, channel quad)
  f4v wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ldv(w + d.dw.w_off + k * 16 + 4 * quad);
  const f4v bias_d = ldv(w + d.dw.b_off + 4 * quad);
  const int cox[3] = {(2 * px) * 16 + 4 * quad, (2 * px + 1) * 16 + 4 * quad, (2 * px + 2) * 16 + 4 * quad};
  __syncthreads();

  // 4. depthwise 3x3 / stride 2 → b0: one tile row per wave iteration
  float4 sumB = f4zero();
  for (int py = wave; py < d.TR && !(d.dbg_skip & 4); py += 4) {
    if (r0 + py >= d.H2) break;
    if (px < d.TC && c0 + px < d.W2) {
      const float4 v = clamp4(tof4(dw3x3(x_f, AC * 16, 2 * py, cox, wd) + bias_d), cl_dw);
      stg4<H16>(b0_out, (unsigned)(((r0 + py) * d.W2 + c0 + px) * 16 + 4 * quad), v);
      sumB = f4add(sumB, v);
    }
  }
  float* s_red = in_t;                                              // (every wave is past the stem — the barrier in front of the depthwise — so the window is free)
  wave_reduce16<0>(sumA, s_red, wave, lane);
  wave_reduce16<1>(sumB, s_red + 256, wave, lane);
  __syncthreads();
  store_partials(s_red, fa + d.part_a_off + (long)t_ * 16);
  store_partials(s_red + 256, fa + d.part_b0_off + (long)t_ * 16);
}

// ==================================================================================================================================
// k2: s = gate(GAP(b0)); B = pw_a(b0 * s) (skip of decoder level 2); x = act(pw_b(B)); c0 = act(dw3x3/s2(x)).  Tile = TR x TC of c0.
// The expanded tensor x (72 channels) exists only 16 channels at a time, in LDS.
// ==================================================================================================================================
#ifdef BSX_SEG_RTC
extern "C" __global__ __launch_bounds__(kSegThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void bsx_seg_k2(
#else
template <bool H16>
__global__ __launch_bounds__(kSegThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void seg_k2_k(
#endif
    const SegK2 d_rt, float* __restrict__ arena, long per_frame, const float* __restrict__ w, int n_frames) {
  BSX_SEG_D(SegK2, K2);
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(d.tiles_y * d.tiles_x), (unsigned)n_frames, &f_, &t_);
  const int f = (int)f_, ty = (int)t_ / d.tiles_x, tx = (int)t_ - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC;
  const int BR = 2 * d.TR + 1, BC = 2 * d.TC + 1, br0 = 2 * r0 - d.dw_pt, bc0 = 2 * c0 - d.dw_pl, RW = d.rw, ctiles = RW >> 4;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* B_t = seg_smem + kScrFloats;                               // [BR][RW][16]
  float* x_t = B_t + BR * RW * 16;
  const float* b0_in = fa + d.b0_off;
  float* B_out = fa + d.B_off;
  float* c0_out = fa + d.c0_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4, cq4 = 4 * g;
  const int ntile = BR * ctiles, xe = li;
  const int la = col_a(li, g), lb = col_b(li, g, RW >> 1);          // this lane's quad at column 16 ct + li of a B_t / x_t row = (16 | 8) ct columns further + these (the swizzles have period <= 16)
  // all of this wave's b0 operands are requested before the gate prologue (one memory round trip for the whole workgroup)
  constexpr int kB0 = 6;                                            // planner: ceil(BR * ctiles / 4) <= 6
  float4 b0v[kB0];
#pragma unroll
  for (int j = 0; j < kB0; j++) {
    const int t = wave + 4 * j;
    b0v[j] = f4zero();
    if (t < ntile) {
      const RowTile rt = row_tile(t, ctiles, d.m_ct);
      const int gy = br0 + rt.row, gx = bc0 + 16 * rt.ct + li;
      if (gy >= 0 && gy < d.H2 && gx >= 0 && gx < d.W2) b0v[j] = ldg4<H16>(b0_in, (unsigned)((gy * d.W2 + gx) * 16 + 4 * g));
    }
  }
  seg_gate(d.gate, fa, w, seg_smem, B_t);
  const float4 sv = ld4(seg_smem + kScrGate + 4 * g);
  float wr[4];
  load_wtile(wr, w, d.pw_a, 0, li, g);
  const float4 bias_a = ld4(w + d.pw_a.b_off + cq4);
  const Clamp cl_a = clamp_of(d.pw_a.act), cl_b = clamp_of(d.pw_b.act), cl_dw = clamp_of(d.dw.act);

  // 1. B on the region the depthwise needs
  float4 sumB = f4zero();
#pragma unroll
  for (int j = 0; j < kB0; j++) {
    const int t = wave + 4 * j;
    if (t >= ntile) break;
    const RowTile rt = row_tile(t, ctiles, d.m_ct);
    float4 a = b0v[j];
    a = make_float4(__fmul_rn(a.x, sv.x), __fmul_rn(a.y, sv.y), __fmul_rn(a.z, sv.z), __fmul_rn(a.w, sv.w));
    const f4acc acc = mma16(a, wr);
    float4 v = acc_quad(acc);
    const int x2 = 16 * rt.ct + xe, hy = br0 + rt.row, hx = bc0 + x2;
    if (x2 < BC) {
      v = clamp4(f4add(v, bias_a), cl_a);
      st4(B_t + (rt.row * RW + 16 * rt.ct) * 16 + la, v);
      const bool row_owned = hy >= max(2 * r0, 0) && hy < min(2 * r0 + 2 * d.TR, d.H2);
      if (row_owned && hx >= 2 * c0 && hx < min(2 * c0 + 2 * d.TC, d.W2)) {
        stg4<H16>(B_out, (unsigned)((hy * d.W2 + hx) * 16 + cq4), v);
        sumB = f4add(sumB, v);
      }
    }
  }
  float* s_red = seg_smem + kScrRed;
  wave_reduce16<0>(sumB, s_red, wave, lane);
  __syncthreads();
  store_partials(s_red, fa + d.part_B_off + (long)t_ * 16);

  // 2. 16 expanded channels at a time: x = act(pw_b(B)) → depthwise 3x3/s2 → c0
  //    (requesting group g + 1's 1x1 tile while group g runs — 8 more registers, 5 instead of 6 waves per SIMD or 16 B of scratch at a cap of 6 — measured SLOWER:
  //     45.4 → 51.3 us at 256 VGA streams, 170 → 187 us for mlkit / HD, profiles/r03af)
  const int C = d.dw.C, ngrp = (C + 15) >> 4, quad = lane & 3, px = lane >> 2;
  const int cox[3] = {col_b(2 * px, quad, RW >> 1), col_b(2 * px + 1, quad, RW >> 1), col_b(2 * px + 2, quad, RW >> 1)};   // x_t: de-interleaved columns (swz_b)
  for (int grp = 0; grp < ngrp && !(d.dbg_skip & 2); grp++) {
    load_wtile(wr, w, d.pw_b, 16 * grp, li, g);
    const float4 bias_b = ld4(w + d.pw_b.b_off + 16 * grp + cq4);
    for (int t = wave; t < ntile; t += 4) {
      const RowTile rt = row_tile(t, ctiles, d.m_ct);
      // (columns >= BC of the row hold nothing: their lanes feed MFMA columns whose results are dropped below — a column of D depends on the same column of B only)
      const f4acc acc = mma16(ld4(B_t + (rt.row * RW + 16 * rt.ct) * 16 + la), wr);
      float4 v = acc_quad(acc);
      const int x2 = 16 * rt.ct + xe, hy = br0 + rt.row, hx = bc0 + x2;
      if (x2 < BC) {
        const bool inside = hy >= 0 && hy < d.H2 && hx >= 0 && hx < d.W2;
        v = inside ? clamp4(f4add(v, bias_b), cl_b) : f4zero();
        st4(x_t + (rt.row * RW + 8 * rt.ct) * 16 + lb, v);
      }
    }
    const int ch = 16 * grp + 4 * quad;
    f4v wd[9];
    f4v bias_d = {0.f, 0.f, 0.f, 0.f};
    if (ch < C) {
#pragma unroll
      for (int k = 0; k < 9; k++) wd[k] = ldv(w + d.dw.w_off + (unsigned)(k * C + ch));
      bias_d = ldv(w + d.dw.b_off + ch);
    }
    __syncthreads();
    if (ch < C && px < d.TC && c0 + px < d.W3)
      for (int py = wave; py < d.TR && r0 + py < d.H3; py += 4) {
        const float4 v = clamp4(tof4(dw3x3(x_t, RW * 16, 2 * py, cox, wd) + bias_d), cl_dw);
        stg4<H16>(c0_out, (unsigned)(((r0 + py) * d.W3 + c0 + px) * C + ch), v);
      }
    __syncthreads();
  }
}

// z = act(pw(skip * g + up(lo))) on the (TR+2) x (TC+2 <= 16) halo region of a tile, zero outside the image — shared by k3 and the
// tail.  One MFMA tile = one region row (16 columns): the row's interpolation pair is scalar, the column's a per-lane constant;
// up = a*w00 + b*w10 + c*w01 + d*w11 with w = products of the two axis fractions (differs from the reference association
// ((a*(1-dy))*(1-dx) + ...) by float rounding only).
// Memory: at 3-4 workgroups per CU a dependent global load costs 1-2 us, so EVERYTHING the phase reads is requested in the first
// instructions of the kernel (gated_prefetch): the lane's skip operand of each of its <= kGatedRows rows into registers, the
// low-resolution rows/columns the tile interpolates from into LDS (l_t, [LR][LC][20]); the compute part then never touches HBM.
constexpr int kGatedRows = 5;           // region rows per wave: (TR + 2 + 3) / 4 <= 5  →  TR <= 18
constexpr int kLoStride = kSegLoStride; // floats per staged low-resolution pixel (16: dense, swizzled by swz_l)
struct GatedPre { float4 s[kGatedRows]; int ly0, lx0, LC; };
template <bool H16>
__device__ __forceinline__ GatedPre gated_prefetch(const float* __restrict__ skip, const float* __restrict__ lo, int H, int W, int HL, int WL, bool half_pixel,
                                                   float hs, float wsc, int r0, int c0, int ZH, int ZC, float* l_t) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4;
  GatedPre pre;
  const int ix = c0 - 1 + li;
  const bool col_in = li < ZC && ix >= 0 && ix < W;
#pragma unroll
  for (int j = 0; j < kGatedRows; j++) {
    const int zy = wave + 4 * j, iy = r0 - 1 + zy;
    pre.s[j] = f4zero();
    if (zy < ZH && iy >= 0 && iy < H && col_in) pre.s[j] = ldg4<H16>(skip, (unsigned)((iy * W + ix) * 16 + 4 * g));
  }
  // low-resolution window: rows y0(first image row of the region) .. y1(last), columns likewise (monotone maps).  hs / wsc: the two up-sampling scales — launch constants the
  // planner computes (SegK3 / SegTail::hs, ::ws: the same IEEE single-precision quotients); computed here they were four correctly-rounded divisions (both `align` forms of
  // each axis) of ~11 instructions in front of the prefetch's first address, and four more in gated_compute
  int ly0, ly1, lx0, lx1, t0, t1;
  float fr;
  up_axis(max(r0 - 1, 0), hs, half_pixel, HL, &ly0, &t1, &fr);
  up_axis(min(r0 - 2 + ZH, H - 1), hs, half_pixel, HL, &t0, &ly1, &fr);
  up_axis(max(c0 - 1, 0), wsc, half_pixel, WL, &lx0, &t1, &fr);
  up_axis(min(c0 - 2 + ZC, W - 1), wsc, half_pixel, WL, &t0, &lx1, &fr);
  const int LR = ly1 - ly0 + 1, LC = lx1 - lx0 + 1;                 // planner: LR <= 12, LC * 4 <= 64
  pre.ly0 = ly0; pre.lx0 = lx0; pre.LC = LC;
  float4 v[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int ly = wave + 4 * j;
    v[j] = f4zero();
    if (ly < LR && lane < LC * 4) v[j] = ldg4<H16>(lo, (unsigned)(((ly0 + ly) * WL + lx0 + (lane >> 2)) * 16 + 4 * (lane & 3)));
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int ly = wave + 4 * j;
    if (ly < LR && lane < LC * 4) st4(l_t + ly * LC * kLoStride + col_l(lane >> 2, lane & 3), v[j]);
  }
  return pre;
}
__device__ __forceinline__ void gated_compute(const GatedPre& pre, const float* l_t, int H, int W, int HL, int WL, bool half_pixel, float hs, float wsc, const float* s_gate,
                                              const SegConvW& pw, const float* __restrict__ w, int r0, int c0, int ZH, int ZC, float* z_t) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4, cq4 = 4 * g;
  float wr[4];
  load_wtile(wr, w, pw, 0, li, g);
  const float4 bias = ld4(w + pw.b_off + cq4);
  const float4 gv = ld4(s_gate + 4 * g);
  const Clamp cl = clamp_of(pw.act);
  // column constants of this lane: operand and epilogue side are the same pixel li
  const int ix = c0 - 1 + li;
  int x0, x1;
  float dx;
  up_axis(min(max(ix, 0), W - 1), wsc, half_pixel, WL, &x0, &x1, &dx);
  const int xo0 = col_l(x0 - pre.lx0, g), xo1 = col_l(x1 - pre.lx0, g);
  const int xe = li;
  const bool ecol_in = xe < ZC && ix >= 0 && ix < W;
#pragma unroll
  for (int j = 0; j < kGatedRows; j++) {
    const int zy = wave + 4 * j, iy = r0 - 1 + zy;
    if (zy >= ZH) break;
    const bool row_in = iy >= 0 && iy < H;                           // scalar
    int y0, y1;
    float dy;
    up_axis(min(max(iy, 0), H - 1), hs, half_pixel, HL, &y0, &y1, &dy);
    const float* l0 = l_t + (y0 - pre.ly0) * pre.LC * kLoStride;
    const float* l1 = l_t + (y1 - pre.ly0) * pre.LC * kLoStride;
    const float4 ta = ld4(l0 + xo0), tb = ld4(l1 + xo0), tc = ld4(l0 + xo1), td = ld4(l1 + xo1);
    const float w00 = (1.f - dy) * (1.f - dx), w10 = dy * (1.f - dx), w01 = (1.f - dy) * dx, w11 = dy * dx;
    const float4 sk = pre.s[j];
    float4 a;
    a.x = fmaf(sk.x, gv.x, fmaf(td.x, w11, fmaf(tc.x, w01, fmaf(tb.x, w10, ta.x * w00))));
    a.y = fmaf(sk.y, gv.y, fmaf(td.y, w11, fmaf(tc.y, w01, fmaf(tb.y, w10, ta.y * w00))));
    a.z = fmaf(sk.z, gv.z, fmaf(td.z, w11, fmaf(tc.z, w01, fmaf(tb.z, w10, ta.z * w00))));
    a.w = fmaf(sk.w, gv.w, fmaf(td.w, w11, fmaf(tc.w, w01, fmaf(tb.w, w10, ta.w * w00))));
    const f4acc acc = mma16(a, wr);
    float4 v = acc_quad(acc);
    v = (ecol_in && row_in) ? clamp4(f4add(v, bias), cl) : f4zero();
    if (xe < ZC) st4(z_t + zy * 256 + col_a(xe, g), v);
  }
}

// ==================================================================================================================================
// k3 (decoder level 2): z = act(pw1(B * g + up(lo2))); t = z + act(dw3x3(z)); lo = pw2(t).  Tile = TR x TC (<= 14) at the B resolution.
// ==================================================================================================================================
#ifdef BSX_SEG_RTC
extern "C" __global__ __launch_bounds__(kSegThreads) void bsx_seg_k3(
#else
template <bool H16>
__global__ __launch_bounds__(kSegThreads) void seg_k3_k(
#endif
    const SegK3 d_rt, float* __restrict__ arena, long per_frame, const float* __restrict__ w, int n_frames) {
  BSX_SEG_D(SegK3, K3);
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(d.tiles_y * d.tiles_x), (unsigned)n_frames, &f_, &t_);
  const int f = (int)f_, ty = (int)t_ / d.tiles_x, tx = (int)t_ - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC, ZH = d.TR + 2, ZC = d.TC + 2;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* z_t = seg_smem + kScrFloats;                               // [ZH][16][16]
  float* l_t = z_t + ZH * 256;                                      // staged window of lo2
  float* lo_out = fa + d.lo_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4, cq4 = 4 * g;
  const GatedPre pre = gated_prefetch<H16>(fa + d.skip_off, fa + d.lo2_off, d.H2, d.W2, d.HL, d.WL, d.half_pixel != 0, d.hs, d.ws, r0, c0, ZH, ZC, l_t);
  if (tid < 16) seg_smem[kScrGate + tid] = fa[d.g_off + tid];
  __syncthreads();
  gated_compute(pre, l_t, d.H2, d.W2, d.HL, d.WL, d.half_pixel != 0, d.hs, d.ws, seg_smem + kScrGate, d.pw1, w, r0, c0, ZH, ZC, z_t);
  // The depthwise runs on the MFMA's own lanes — lane (li, g) = (pixel of the row, channel quad) — so t = z + act(dw(z)) of the lane IS the B operand of pw2's four
  // MFMAs (round 5).  Until then the depthwise used (pixel = lane >> 2, quad = lane & 3) lanes and t made a round trip through an LDS tile of its own ([TR][16][16]:
  // 12 KB, a barrier): k3 34.8 -> 22.8 KiB per workgroup at a 12 x 14 tile.  Same values into the same instructions: bit-identical.  Under swz_a the (li, g) lanes read
  // z_t without bank conflicts, exactly like an MFMA operand read.  Lanes 14, 15 (TC <= 14) repeat lane 13's columns; their results are dropped.
  const int pc = min(li, 13);
  const int coz[3] = {col_a(pc, g), col_a(pc + 1, g), col_a(pc + 2, g)};
  f4v wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ldv(w + d.dw.w_off + k * 16 + cq4);
  const f4v bias_d = ldv(w + d.dw.b_off + cq4);
  const Clamp cl_dw = clamp_of(d.dw.act), cl_2 = clamp_of(d.pw2.act);
  float wr[4];
  load_wtile(wr, w, d.pw2, 0, li, g);
  const float4 bias2 = ld4(w + d.pw2.b_off + cq4);
  __syncthreads();
  float4 sum = f4zero();
  const int xe = li;
  for (int py = wave; py < d.TR && r0 + py < d.H2; py += 4) {
    const f4v zc = ldv(z_t + (py + 1) * 256 + coz[1]);
    const float4 dv = clamp4(tof4(dw3x3(z_t, 256, py, coz, wd) + bias_d), cl_dw);
    const f4acc acc = mma16(f4add(dv, tof4(zc)), wr);                  // dw epilogue: activation, then + residual z; straight into pw2
    float4 v = acc_quad(acc);
    if (xe < d.TC && c0 + xe < d.W2) {
      v = clamp4(f4add(v, bias2), cl_2);
      stg4<H16>(lo_out, (unsigned)(((r0 + py) * d.W2 + c0 + xe) * 16 + cq4), v);
      sum = f4add(sum, v);
    }
  }
  float* s_red = seg_smem + kScrRed;
  wave_reduce16<0>(sum, s_red, wave, lane);
  __syncthreads();
  store_partials(s_red, fa + d.part_lo_off + (long)t_ * 16);
}

// ==================================================================================================================================
// tail (decoder level 1 + output): g = gate(GAP(A), GAP(lo)); z = act(pw(A * g + up(lo))); t = z + act(dw3x3(z));
// out = act3(Convolution2DTransposeBias 2x2 (t)) → logits (LOGITS) or straight into decode + temporal IIR on `ofinal`.
// Tile = TR x TC (<= 14) at the A resolution = 2TR x 2TC output pixels.  Phase B lanes = the MFMA's: lane (li, g) = (pixel li of the tile row, channel
// quad g) runs the depthwise on its 4 channels, and its t = z + act(dw(z)) IS the B operand of the transpose convolution, one 16 x 16 MFMA tile per row
// (A = the 2 x 2 x CO filter as 16 rows n = 4 * pos + oc): lane (li, g) ends up with output position (fy, fx) = (g >> 1, g & 1) of pixel li.
// INVARIANT the MFMA forms here and in seg_k2_k / seg_k3_k rely on: a D column depends on the SAME B column only.  Lanes li >= TC (and LDS columns x >= BC in k2)
// feed unwritten LDS into B; their D columns are garbage and every store / sum of an MFMA result below is guarded by the lane's own pixel being inside the tile
// (lane_on / x2 < BC).  A new consumer of those results must carry the same guard.
// ==================================================================================================================================
#ifdef BSX_SEG_RTC
extern "C" __global__ __launch_bounds__(kSegThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void bsx_seg_tail(
#else
template <int CO, bool LOGITS, bool SIGMOID, bool H16>
__global__ __launch_bounds__(kSegThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void seg_tail_k(
#endif
    const SegTail d_rt, float* __restrict__ arena, long per_frame, float* __restrict__ net_out,
                                                          uint8_t* __restrict__ ofinal, const float* __restrict__ w, int n_frames) {
  BSX_SEG_D(SegTail, TAIL);
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(d.tiles_y * d.tiles_x), (unsigned)n_frames, &f_, &t_);
  const int f = (int)f_, ty = (int)t_ / d.tiles_x, tx = (int)t_ - ty * d.tiles_x;
  const int r0 = ty * d.TR, c0 = tx * d.TC, ZH = d.TR + 2, ZC = d.TC + 2;
  float* fa = arena + (size_t)f * (size_t)per_frame;
  float* z_t = seg_smem + kScrFloats;                               // [ZH][16][16]
  float* l_t = z_t + max(ZH * 256, kGateStageFloats);               // staged window of lo
  uint8_t* of = ofinal + (size_t)f * (size_t)(d.H0 * d.W0);
  float* no = net_out + (size_t)f * (size_t)(d.H0 * d.W0 * CO);
  // Phase B lanes = the MFMA's: lane (li, g) = (pixel li of the tile row, channel quad g) runs the depthwise on its 4 channels; its t = z + act(dw(z)) is then the B
  // operand of the transpose convolution, computed as ONE 16 x 16 MFMA tile per row (round 5): A = the 2x2xCO filter as 16 rows n = 4 * pos + oc (rows with oc >= CO are
  // zero) over k = 16 channels, so the accumulator of lane (li, g) holds out[pos = g][oc = 0..CO) of pixel li — the lane ends up with output position (fy, fx) = (g >> 1, g & 1)
  // of its pixel, as before.  Until then every lane held the 4 x CO filter quads of its channels (32 registers at CO = 2), did 16 CO FMAs and three DPP exchanges
  // per output (quad_reduce_scatter): 124 registers = 4 workgroups per CU where LDS allows 5.  (Summation order over the 16 channels changes: within float rounding of
  // the reference order, as every 1x1 convolution here.)
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), li = lane & 15, g = lane >> 4, cq4 = 4 * g;
  const int fy = g >> 1, fx = g & 1, ix = c0 + li;
  // every global read of the workgroup is requested here, before the first wait: skip operands, the window of lo, the temporal
  // state bytes this lane will update, then (inside seg_gate) the pooled partial sums and the gate weights
  const GatedPre pre = gated_prefetch<H16>(fa + d.skip_off, fa + d.lo_off, d.H1, d.W1, d.HL, d.WL, d.half_pixel != 0, d.hs, d.ws, r0, c0, ZH, ZC, l_t);
  constexpr int kRowsB = 5;                                          // TR <= 18 → <= 5 tile rows per wave in phase B
  const bool lane_on = li < d.TC && ix < d.W1;
  uint8_t prev[kRowsB];
  if (!LOGITS) {
#pragma unroll
    for (int j = 0; j < kRowsB; j++) {
      const int py = wave + 4 * j, iy = r0 + py;
      prev[j] = 0;
      if (py < d.TR && iy < d.H1 && lane_on) prev[j] = of[(unsigned)((2 * iy + fy) * d.W0 + 2 * ix + fx)];
    }
  }
  if (d.pre_gate_off >= 0) {                                          // computed once per frame by seg_gate_k
    if (tid < 16) seg_smem[kScrGate + tid] = fa[d.pre_gate_off + tid];
    __syncthreads();
  } else seg_gate(d.gate, fa, w, seg_smem, z_t);
  if (!(d.dbg_skip & 1)) gated_compute(pre, l_t, d.H1, d.W1, d.HL, d.WL, d.half_pixel != 0, d.hs, d.ws, seg_smem + kScrGate, d.pw, w, r0, c0, ZH, ZC, z_t);
  if (d.dbg_skip & 2) return;
  f4v wd[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wd[k] = ldv(w + d.dw.w_off + k * 16 + cq4);
  const f4v bias_d = ldv(w + d.dw.b_off + cq4);
  float wtr[4];                                                       // A operand: lane (li, g) holds Wt[n = li][k = 4 g + r], n = 4 * pos + oc
#pragma unroll
  for (int r = 0; r < 4; r++) wtr[r] = (li & 3) < CO ? w[d.tc_w_off + (unsigned)((((li >> 2) * CO + (li & 3)) * 16) + cq4 + r)] : 0.f;
  float bt[CO];
#pragma unroll
  for (int oc = 0; oc < CO; oc++) bt[oc] = w[d.tc_b_off + oc];
  const Clamp cl_dw = clamp_of(d.dw.act);
  const int pc = min(li, 13);                                         // TC <= 14: lanes 14, 15 repeat lane 13's columns (dropped)
  const int coz[3] = {col_a(pc, g), col_a(pc + 1, g), col_a(pc + 2, g)};
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRowsB; j++) {
    const int py = wave + 4 * j, iy = r0 + py;
    if (py >= d.TR || iy >= d.H1) break;                              // scalar: the MFMAs below run with every lane of the wave
    const f4v zc = ldv(z_t + (py + 1) * 256 + coz[1]);
    const float4 t4 = f4add(clamp4(tof4(dw3x3(z_t, 256, py, coz, wd) + bias_d), cl_dw), tof4(zc));
    const f4acc acc = mma16(t4, wtr);
    if (lane_on) {
      float o[CO];
#pragma unroll
      for (int oc = 0; oc < CO; oc++) {
        o[oc] = bt[oc] + acc[oc];
        if (SIGMOID) o[oc] = sigmoid1(o[oc]);
      }
      const int oy = 2 * iy + fy, ox = 2 * ix + fx;
      const unsigned opix = (unsigned)(oy * d.W0 + ox);
      if (LOGITS) {
#pragma unroll
        for (int oc = 0; oc < CO; oc++) no[opix * CO + oc] = o[oc];
      } else {
        uint32_t val;
        if (CO == 2) val = seg_meet_val(o[0], o[CO - 1]);
        else val = ((double)o[0] > 0.65) ? 0u : 255u;                // MLKit: float promoted to double against the double literal (libbackscrub.cc:338)
        of[opix] = (uint8_t)((val & 0xE0u) | ((uint32_t)prev[j] >> 3));
      }
    }
  }
}

#ifndef BSX_SEG_RTC                // ---- from here on: ahead-of-time build only (the per-frame gate kernel, a 5 us launch, and the host side)
// the gate of one decoder level, once per frame: workgroup = frame; the prologue of the tile kernels as a kernel of its own (SegTail::pre_gate_off)
__global__ __launch_bounds__(kSegThreads) void seg_gate_k(const SegGate gt, float* __restrict__ arena, long per_frame, const float* __restrict__ w, long long out_off) {
  float* fa = arena + (size_t)blockIdx.x * (size_t)per_frame;
  seg_gate(gt, fa, w, seg_smem, seg_smem + kScrFloats);
  if (threadIdx.x < 16) fa[out_off + threadIdx.x] = seg_smem[kScrGate + threadIdx.x];
}

template <class K>
hipError_t allow_lds(K kernel, int lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}
#endif

}  // namespace

#ifndef BSX_SEG_RTC
hipError_t seg_prepare() {
  const int full = 160 * 1024;     // process-global kernel attributes: always the full LDS (cf. frame_program_prepare)
  hipError_t e = hipSuccess;
  auto one = [&](auto k) { if (e == hipSuccess) e = allow_lds(k, full); };
  one(seg_head_k<true, false, false>); one(seg_head_k<false, false, false>); one(seg_head_k<true, true, false>); one(seg_head_k<false, true, false>);
  one(seg_head_k<true, false, true>); one(seg_head_k<false, false, true>); one(seg_head_k<true, true, true>); one(seg_head_k<false, true, true>);
  one(seg_k2_k<false>); one(seg_k2_k<true>); one(seg_k3_k<false>); one(seg_k3_k<true>);
  one(seg_tail_k<1, false, true, false>); one(seg_tail_k<1, true, true, false>); one(seg_tail_k<1, false, false, false>); one(seg_tail_k<1, true, false, false>);
  one(seg_tail_k<2, false, false, false>); one(seg_tail_k<2, true, false, false>);
  one(seg_tail_k<1, false, true, true>); one(seg_tail_k<1, true, true, true>); one(seg_tail_k<1, false, false, true>); one(seg_tail_k<1, true, false, true>);
  one(seg_tail_k<2, false, false, true>); one(seg_tail_k<2, true, false, true>);
  return e;
}

// n_frames = 0 tells xcd_frame_tile to keep the plain (frame-major) workgroup order: BSX_XCD_TILES=0, read once per process, for A/B timing
static int xcd_frames(int n) { static const bool on = !(BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 0); return on ? n : 0; }

// h16: the boundary tensors are stored as halves (BSX_ACT16; the middle program must have been generated for the same storage)
// u8: net_in points at the 8-bit network input ([n][H0][W0] u32 pixels, prep_fused_k<2>) and (scale, offset) is the model's normalisation
hipError_t launch_seg_head(const SegHead& d, float* arena, long per_frame, const void* net_in, const float* weights, int n, hipStream_t s, bool h16, bool u8, float in_scale,
                           float in_offset) {
  const dim3 grid((unsigned)(d.tiles_y * d.tiles_x) * (unsigned)n);
  const size_t lds = (size_t)d.lds_floats * sizeof(float);
  const bool hs = d.stem.act == kActHswish;
  const float* in = static_cast<const float*>(net_in);
  if (u8 && 2 * (2 * d.TC + 1) + 1 > 64) return hipErrorInvalidValue;          // one lane per pixel of an input-tile row (the planner's TC <= 15)
#define BSX_HEAD(HS, H, U) seg_head_k<HS, H, U><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, in, weights, in_scale, in_offset, xcd_frames(n))
  if (hs) { if (h16) { if (u8) BSX_HEAD(true, true, true); else BSX_HEAD(true, true, false); } else { if (u8) BSX_HEAD(true, false, true); else BSX_HEAD(true, false, false); } }
  else { if (h16) { if (u8) BSX_HEAD(false, true, true); else BSX_HEAD(false, true, false); } else { if (u8) BSX_HEAD(false, false, true); else BSX_HEAD(false, false, false); } }
#undef BSX_HEAD
  return hipGetLastError();
}
hipError_t launch_seg_k2(const SegK2& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s, bool h16) {
  const dim3 grid((unsigned)(d.tiles_y * d.tiles_x) * (unsigned)n);
  if (h16) seg_k2_k<true><<<grid, kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights, xcd_frames(n));
  else seg_k2_k<false><<<grid, kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights, xcd_frames(n));
  return hipGetLastError();
}
hipError_t launch_seg_k3(const SegK3& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s, bool h16) {
  const dim3 grid((unsigned)(d.tiles_y * d.tiles_x) * (unsigned)n);
  if (h16) seg_k3_k<true><<<grid, kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights, xcd_frames(n));
  else seg_k3_k<false><<<grid, kSegThreads, (size_t)d.lds_floats * sizeof(float), s>>>(d, arena, per_frame, weights, xcd_frames(n));
  return hipGetLastError();
}
template <bool H16>
static hipError_t launch_seg_tail_t(const SegTail& d, float* arena, long per_frame, float* net_out, uint8_t* ofinal, const float* weights, bool logits, int n, hipStream_t s) {
  const dim3 grid((unsigned)(d.tiles_y * d.tiles_x) * (unsigned)n);
  // BSX_SEG_TAIL_WGS=<k> (experiment switch, read once): pad the dynamic LDS so that at most k workgroups fit a CU — the A/B of the tail's 5th workgroup (round 5)
  static const int wgs_cap = BSX_DBG_ENV("BSX_SEG_TAIL_WGS") ? atoi(BSX_DBG_ENV("BSX_SEG_TAIL_WGS")) : 0;
  size_t lds = (size_t)d.lds_floats * sizeof(float);
  if (wgs_cap > 0) lds = std::max(lds, (size_t)(160 * 1024 / (wgs_cap + 1) + 256));
  const bool sig = d.act3 == kActSigmoid;
  if (d.Co == 2 && !sig) {
    if (logits) seg_tail_k<2, true, false, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
    else seg_tail_k<2, false, false, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
  } else if (d.Co == 1 && sig) {
    if (logits) seg_tail_k<1, true, true, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
    else seg_tail_k<1, false, true, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
  } else if (d.Co == 1) {
    if (logits) seg_tail_k<1, true, false, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
    else seg_tail_k<1, false, false, H16><<<grid, kSegThreads, lds, s>>>(d, arena, per_frame, net_out, ofinal, weights, xcd_frames(n));
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_seg_gate(const SegGate& gt, float* arena, long per_frame, const float* weights, long long out_off, int n, hipStream_t s) {
  seg_gate_k<<<n, kSegThreads, (size_t)(kSegScratchFloats + kSegGateStageFloats) * sizeof(float), s>>>(gt, arena, per_frame, weights, out_off);
  return hipGetLastError();
}
hipError_t launch_seg_tail(const SegTail& d, float* arena, long per_frame, float* net_out, uint8_t* ofinal, const float* weights, bool logits, int n, hipStream_t s, bool h16) {
  return h16 ? launch_seg_tail_t<true>(d, arena, per_frame, net_out, ofinal, weights, logits, n, s) : launch_seg_tail_t<false>(d, arena, per_frame, net_out, ofinal, weights, logits, n, s);
}

#endif

}  // namespace bsx
