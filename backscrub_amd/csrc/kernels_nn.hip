// kernels_nn.hip — batched NHWC f32 network kernels for gfx950 (wave64).
//
// These replace the TFLite/XNNPACK operators executed by Interpreter::Invoke()
// (/root/reference/lib/libbackscrub.cc:307) and the in-tree custom op
// (/root/reference/lib/transpose_conv_bias.cc:37-114).  The batch dimension is the set of
// independent camera streams; every launch covers all streams.
//
// Design notes (gfx950):
//  * Pointwise / dense convolutions: one lane owns one pixel and a tile of CT output
//    channels.  The weight address depends only on (ci, blockIdx.y), i.e. it is
//    wave-uniform, so hipcc emits scalar loads (s_load_dwordx8/16) and the FMAs take the
//    weight from an SGPR — no LDS staging, no per-lane weight traffic.  K and N of these
//    GEMMs are 8..128 in the Meet / MLKit graphs; this SGPR form serves the small layers.
//  * Pointwise convolutions with enough work (DeepLab: M = streams x 33x33 .. 129x129 pixels, K up to 960, N up to 256)
//    are real GEMMs: pw_gemm_mfma_k tiles them 128 x 64 x 32 through LDS onto v_mfma_f32_16x16x4_f32 (exact f32).
//  * Depthwise: lanes run along channel-quads then x, so every tap is a coalesced float4
//    row segment; neighbouring lanes re-use taps through L1.
//  * Accumulation is ci-ascending FMA with the bias added last, the same association as
//    the TFLite reference kernels the CPU oracle restates (differences are FMA rounding).
#include "debug_switches.hpp"
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "mfma_tile.hpp"

namespace bsx {
namespace {

constexpr int kThreads = 256;

// `act` is uniform across a launch.  Written as a switch, every call compiled into a chain of scalar compares and TAKEN branches around
// inlined code for all five kinds (incl. the IEEE divide and expf expansions): ~60 instructions and 4-6 pipeline refills per element,
// 32 elements per lane in a GEMM epilogue.  The three clamp kinds (none / relu / relu6 — everything DeepLab uses) are ONE v_med3_f32 with
// uniform bounds; only hswish / sigmoid take a (single, uniform) branch.
__device__ __forceinline__ float act_slow(float v, int act) {
  if (act == kActHswish) return v * fminf(6.f, fmaxf(0.f, v + 3.f)) / 6.f;
  return 1.f / (1.f + expf(-v));
}
struct ClampK { float lo, hi; };                                     // the three clamp activations as bounds of one v_med3_f32
__device__ __forceinline__ ClampK clamp_of(int act) { return ClampK{act == kActNone ? -__builtin_huge_valf() : 0.f, act == kActRelu6 ? 6.f : __builtin_huge_valf()}; }
__device__ __forceinline__ float clampf(float v, const ClampK& c) { return __builtin_amdgcn_fmed3f(v, c.lo, c.hi); }
__device__ __noinline__ float act_slow_call(float v, int act) { return act_slow(v, act); }     // out of line: keeps the hot epilogues small
__device__ __forceinline__ float act_fn(float v, int act) {
  if (__builtin_expect(act >= kActHswish, 0)) return act_slow_call(v, act);
  const float lo = act == kActNone ? -__builtin_huge_valf() : 0.f, hi = act == kActRelu6 ? 6.f : __builtin_huge_valf();
  return __builtin_amdgcn_fmed3f(v, lo, hi);          // med3(v, -inf, +inf) = v for every non-NaN v; max(0, v) / min(max(0, v), 6) otherwise
}

// -------------------------------------------------------------------------------------
// 1x1 convolution / fully-connected: y[p, co] = act(sum_ci x[p,ci]*s[n,ci] * w[ci,co] + b[co]) + res[p,co]
// -------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(kThreads) void pw_conv_k(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     const float* __restrict__ scale, const float* __restrict__ addx,
                                                     float* __restrict__ y, long M, int HW, int Cin, int Cout, int cout_pad, int act) {
  long p = (long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= M) return;
  const int co0 = blockIdx.y * CT;
  const float* xp = x + p * Cin;
  const float* ap = addx ? addx + p * Cin : nullptr;   // x' = x*s + a (a fused MUL+ADD feeding this conv): two roundings, like the graph
  const float* sp = scale ? scale + (p / HW) * (long)Cin : nullptr;
  const float* wp = w + co0;
  float acc[CT];
#pragma unroll
  for (int t = 0; t < CT; t++) acc[t] = 0.f;
  if ((Cin & 3) == 0) {
    for (int ci = 0; ci < Cin; ci += 4) {
      float4 xv = *reinterpret_cast<const float4*>(xp + ci);
      if (sp) {
        float4 sv = *reinterpret_cast<const float4*>(sp + ci);
        xv.x = __fmul_rn(xv.x, sv.x); xv.y = __fmul_rn(xv.y, sv.y); xv.z = __fmul_rn(xv.z, sv.z); xv.w = __fmul_rn(xv.w, sv.w);
      }
      if (ap) {
        float4 av = *reinterpret_cast<const float4*>(ap + ci);
        xv.x = __fadd_rn(xv.x, av.x); xv.y = __fadd_rn(xv.y, av.y); xv.z = __fadd_rn(xv.z, av.z); xv.w = __fadd_rn(xv.w, av.w);
      }
      const float* w0 = wp + (long)ci * cout_pad;
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.x, w0[t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.y, w0[cout_pad + t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.z, w0[2 * cout_pad + t], acc[t]);
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv.w, w0[3 * cout_pad + t], acc[t]);
    }
  } else {
    for (int ci = 0; ci < Cin; ci++) {
      float xv = xp[ci];
      if (sp) xv = __fmul_rn(xv, sp[ci]);
      if (ap) xv = __fadd_rn(xv, ap[ci]);
      const float* w0 = wp + (long)ci * cout_pad;
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xv, w0[t], acc[t]);
    }
  }
  float* yp = y + p * Cout + co0;
  const float* rp = res ? res + p * Cout + co0 : nullptr;
  const float* bp = bias + co0;
  if ((Cout & 3) == 0) {
#pragma unroll
    for (int t = 0; t < CT; t += 4) {
      if (co0 + t < Cout) {
        float4 v;
        v.x = act_fn(acc[t] + bp[t], act); v.y = act_fn(acc[t + 1] + bp[t + 1], act);
        v.z = act_fn(acc[t + 2] + bp[t + 2], act); v.w = act_fn(acc[t + 3] + bp[t + 3], act);
        if (rp) { float4 r = *reinterpret_cast<const float4*>(rp + t); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        *reinterpret_cast<float4*>(yp + t) = v;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < CT; t++) {
      if (co0 + t < Cout) {
        float v = act_fn(acc[t] + bp[t], act);
        if (rp) v += rp[t];
        yp[t] = v;
      }
    }
  }
}

// -------------------------------------------------------------------------------------
// 1x1 convolution as an LDS-tiled MFMA GEMM:  y[M][Cout] = act(x'[M][Cin] * w[Cin][cout_pad] + b) (+ res),  x' = x*s + a
// -------------------------------------------------------------------------------------
// Workgroup (4 waves) = 128 x 64 output tile, K in steps of 32.  Wave w owns rows 32w..32w+31 (2 m-tiles) x 4 n-tiles =
// 8 accumulators.  Per K step a lane reads ONE float4 of A per m-tile and 16-row chunk (k = 4g..4g+3 feed four successive
// MFMAs as .x/.y/.z/.w) and one scalar of B per MFMA (row 4g+j, column li) — the k assignment is a bijection shared by
// both operands, so the sum is the exact f32 FMA chain in a fixed order.  The next K tile is fetched into registers while
// the current one is multiplied; SE scale / fused a-add are applied while the A tile is written to LDS.
constexpr int kGemmBM = 128, kGemmBN = 64, kGemmBK = 32;
constexpr int kGemmSA = kGemmBK + 4;      // (SA / 4) odd: the 16-byte A reads of 16 consecutive rows hit distinct bank quads
constexpr int kGemmSB = kGemmBN + 4;      // 4 * SB = 16 (mod 32): the four k-rows a wave reads per MFMA split over both bank halves
__global__ __launch_bounds__(kThreads) void pw_gemm_mfma_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ res, const float* __restrict__ scale, const float* __restrict__ addx,
                                                          float* __restrict__ y, long M, int HW, int Cin, int Cout, int cout_pad, int act,
                                                          const float* __restrict__ fbias) {
  __shared__ __attribute__((aligned(16))) float As[kGemmBM * kGemmSA];
  __shared__ __attribute__((aligned(16))) float Bs[kGemmBK * kGemmSB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const long m_base = (long)blockIdx.x * kGemmBM;
  const int n_base = blockIdx.y * kGemmBN;
  const int nt = min(4, (cout_pad - n_base) >> 4);          // valid 16-wide n-tiles of this workgroup (wave-uniform)
  // loader mapping: A = 128 rows x 8 float4 (4 per lane), B = 32 rows x 16 float4 (2 per lane)
  float4 ra[4], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int f = tid + i * kThreads, row = f >> 3, kq = (f & 7) * 4;
      const long m = m_base + row;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && k0 + kq < Cin) {
        float4 v = *reinterpret_cast<const float4*>(x + m * Cin + k0 + kq);
        if (scale) {
          const float4 sv = *reinterpret_cast<const float4*>(scale + (m / HW) * (long)Cin + k0 + kq);
          v.x = __fmul_rn(v.x, sv.x); v.y = __fmul_rn(v.y, sv.y); v.z = __fmul_rn(v.z, sv.z); v.w = __fmul_rn(v.w, sv.w);
        }
        if (addx) {
          const float4 av = *reinterpret_cast<const float4*>(addx + m * Cin + k0 + kq);
          v.x = __fadd_rn(v.x, av.x); v.y = __fadd_rn(v.y, av.y); v.z = __fadd_rn(v.z, av.z); v.w = __fadd_rn(v.w, av.w);
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int f = tid + i * kThreads, kr = f >> 4, nq = (f & 15) * 4;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kr < Cin && n_base + nq < cout_pad) rb[i] = *reinterpret_cast<const float4*>(w + (long)(k0 + kr) * cout_pad + n_base + nq);
    }
  };
  f4acc acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = f4acc{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int k0 = 0; k0 < Cin; k0 += kGemmBK) {
#pragma unroll
    for (int i = 0; i < 4; i++) { const int f = tid + i * kThreads; *reinterpret_cast<float4*>(&As[(f >> 3) * kGemmSA + (f & 7) * 4]) = ra[i]; }
#pragma unroll
    for (int i = 0; i < 2; i++) { const int f = tid + i * kThreads; *reinterpret_cast<float4*>(&Bs[(f >> 4) * kGemmSB + (f & 15) * 4]) = rb[i]; }
    __syncthreads();
    if (k0 + kGemmBK < Cin) fetch(k0 + kGemmBK);            // in flight while this tile is multiplied
#pragma unroll
    for (int jj = 0; jj < kGemmBK / 16; jj++) {
      float4 a[2];
#pragma unroll
      for (int mi = 0; mi < 2; mi++) a[mi] = *reinterpret_cast<const float4*>(&As[(32 * wave + 16 * mi + li) * kGemmSA + 16 * jj + 4 * g]);
#pragma unroll
      for (int ni = 0; ni < 4; ni++) {
        if (ni < nt) {
          const float* br = &Bs[(16 * jj + 4 * g) * kGemmSB + 16 * ni + li];
          const float b0 = br[0], b1 = br[kGemmSB], b2 = br[2 * kGemmSB], b3 = br[3 * kGemmSB];
#pragma unroll
          for (int mi = 0; mi < 2; mi++) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].x, b0, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].y, b1, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].z, b2, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi].w, b3, acc[mi][ni], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  // epilogue: quad transpose → this lane owns pixel (4g + q) of the m-tile, channels c0 .. c0+3 of the n-tile
  const int q = li & 3;
  // the epilogue is instantiated twice and the (uniform) activation kind tested ONCE: a test per element is a taken branch per element
  auto epilogue = [&](auto actf) {
  #pragma unroll
    for (int mi = 0; mi < 2; mi++) {
      const long m = m_base + 32 * wave + 16 * mi + 4 * g + q;
  #pragma unroll
      for (int ni = 0; ni < 4; ni++) {
        if (ni >= nt) continue;
        const int c0 = n_base + 16 * ni + (li & ~3);
        const float4 v = quad_transpose(acc[mi][ni], q);
        if (m >= M || c0 >= Cout) continue;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if ((Cout & 3) == 0) {
          float4 bv = *reinterpret_cast<const float4*>(bias + c0);
          if (fbias) { const float4 fb = *reinterpret_cast<const float4*>(fbias + (m / HW) * (long)Cout + c0); bv.x += fb.x; bv.y += fb.y; bv.z += fb.z; bv.w += fb.w; }
          float4 o = make_float4(actf(vv[0] + bv.x), actf(vv[1] + bv.y), actf(vv[2] + bv.z), actf(vv[3] + bv.w));
          if (res) { const float4 r = *reinterpret_cast<const float4*>(res + m * Cout + c0); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
          *reinterpret_cast<float4*>(y + m * Cout + c0) = o;
        } else {
          for (int e = 0; e < 4 && c0 + e < Cout; e++) {
            float o = actf(vv[e] + bias[c0 + e] + (fbias ? fbias[(m / HW) * (long)Cout + c0 + e] : 0.f));
            if (res) o += res[m * Cout + c0 + e];
            y[m * Cout + c0 + e] = o;
          }
        }
      }
    }
  };
  if (act >= kActHswish) epilogue([&](float v) { return act_slow(v, act); });
  else { const ClampK ck = clamp_of(act); epilogue([&](float v) { return clampf(v, ck); }); }
}

// -------------------------------------------------------------------------------------
// 1x1 convolution as a SPLIT-f16 MFMA GEMM (the "fp16 MFMA pointwise" mode): same tiling as pw_gemm_mfma_k, but on
// v_mfma_f32_16x16x32_f16 — 16x the f32 MFMA rate — with every f32 operand carried as TWO halves, x = xh + xl:
//     x*w  ≈  xh*wh + xl*wh + xh*wl          (products of halves are exact in the f32 accumulator; the dropped xl*wl is 2^-22 relative)
// so the result keeps ~22 significant bits: the logits stay inside the 1e-4 parity bar against the f32 oracle (measured),
// unlike plain f16 activations.  3 MFMAs of K = 32 replace 8 f32 MFMAs of K = 4: ~5x less matrix-pipe time per MAC.
// Activations are split while the A tile is staged into LDS (v_cvt_pkrtz: round-toward-zero never overflows to inf; the low
// half absorbs the larger rounding error); the weights were split once on the host (plan.cpp, [hi | lo][cout_pad][Kp] halves,
// k contiguous so that a lane's 8 consecutive k of one output channel are one 16-byte LDS read).
// TERMS = 1 keeps only xh*wh (plain f16 inputs, f32 accumulate): the IoU-gated fast mode.
// -------------------------------------------------------------------------------------
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
// x = hi + lo for two floats: hi = the f16 below |x| (round toward zero: never overflows to inf), lo = f16(x - hi).  The residual is ONE mixed-precision
// FMA per element (v_fma_mix{lo,hi}_f16: f16 x f32 + f32 → f16, x - hi is exact in f32) instead of convert-back + subtract + pack: 3 instructions per
// pair, not 5 — the split is most of the VALU work of every kernel that multiplies f32 activations on the f16 matrix cores.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
template <int TERMS, typename Q>                    // Q: float4 or f4v
__device__ __forceinline__ void split8(const Q v0, const Q v1, h8v& hi, h8v& lo) {
  u4v h, l;
  if (TERMS == 3) {
    unsigned a, b;
    split_pair(v0.x, v0.y, a, b); h.x = a; l.x = b;
    split_pair(v0.z, v0.w, a, b); h.y = a; l.y = b;
    split_pair(v1.x, v1.y, a, b); h.z = a; l.z = b;
    split_pair(v1.z, v1.w, a, b); h.w = a; l.w = b;
  } else {
    h.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0.x, v0.y)); h.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0.z, v0.w));
    h.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v1.x, v1.y)); h.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v1.z, v1.w));
    l = h;
  }
  hi = __builtin_bit_cast(h8v, h);
  lo = __builtin_bit_cast(h8v, l);
}
constexpr int kHSA = kGemmBK + 8;        // halves per LDS row: 80 B → the 16-byte reads of 16 consecutive rows hit distinct bank quads
// (An unpadded 64-byte row with the chunk XOR-swizzled by (0, 3, 2, 1)[row / 4] is conflict-free for the real ds_read_b128 lane groups — the padded
//  rows leave SQ_LDS_BANK_CONFLICT at 50 % of SQ_LDS_IDX_ACTIVE — but measured 1-3 % SLOWER: the kernel is not LDS-bound and the swizzle costs VALU
//  in the staging stores.)
// NTW = 16-channel tiles per workgroup (4: a 128 x 64 tile; a 128 x 160 variant for the projection layers measured 25 % slower: registers)
// XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 and every XCD has its own L2, so "consecutive workgroups
// share the A tile" only helps if consecutive means consecutive ON ONE XCD.  A 1-D grid is re-indexed so that XCD k walks its own
// contiguous range of tiles, column tile fastest: the up-to-8 column tiles of a 128-row A block run back to back on the same XCD
// and the block leaves HBM once instead of once per column tile (bijective for any grid size; speed only, never correctness).
__device__ __forceinline__ void xcd_tile(unsigned ncol, unsigned* col, long* row) {
  const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, local = orig >> 3, q = nwg >> 3, r = nwg & 7;
  const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  *row = wg / ncol;
  *col = wg - (unsigned)*row * ncol;
}

// Epilogue of the 16x16-tile GEMM kernels: quad transpose → this lane owns pixel (4g + q) of each m-tile and channels c0 .. c0+3 of each n-tile.
// Instantiated twice and the (uniform) activation kind tested ONCE: a test per element is a taken branch per element.
// ALL of the lane's bias / per-frame bias / residual quads are requested up front from clamped (always valid) addresses: written as
// "skip the tile if it is outside" every (m-tile, n-tile) was its own basic block — load, wait, store, ten times in a row, each a full trip to
// L2 / HBM under load — and timing the kernel without its epilogue showed a third of its time there (480 -> 80 project layer: 690 -> 446 us).
template <int NTW>
__device__ __forceinline__ void gemm_store_tile(f4acc (&acc)[2][NTW], long m_wave, int li, int g, int nt, int n_base, long M, int HW, int Cout,
                                                const float* __restrict__ bias, const float* __restrict__ fbias, const float* __restrict__ res,
                                                float* __restrict__ y, int act) {
  const int q = li & 3;
  if ((Cout & 3) == 0) {
    auto epilogue = [&](auto actf) {
      float4 bv[NTW];
      int c0s[NTW];
      const bool fb_on = fbias != nullptr, res_on = res != nullptr;      // (uniform)
#pragma unroll
      for (int ni = 0; ni < NTW; ni++) {
        c0s[ni] = n_base + 16 * ni + (li & ~3);
        bv[ni] = *reinterpret_cast<const float4*>(bias + min(c0s[ni], Cout - 4));
      }
      // ex: the per-frame bias (added before the activation) OR the residual (after): launch_step never passes both.  One m-tile at a time when the
      // column tile is wide (NTW > 5: 2 x NTW quads would not fit the registers), both at once otherwise.
      constexpr int MB = NTW > 5 ? 1 : 2;
#pragma unroll
      for (int m0 = 0; m0 < 2; m0 += MB) {
        float4 ex[MB][NTW];
        long ms[MB];
#pragma unroll
        for (int mi = 0; mi < MB; mi++) {
          ms[mi] = m_wave + 16 * (m0 + mi) + 4 * g + q;
#pragma unroll
          for (int ni = 0; ni < NTW; ni++) ex[mi][ni] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (fb_on || res_on) {                                            // one uniform branch around all of the loads
#pragma unroll
          for (int mi = 0; mi < MB; mi++) {
            const long mc = min(ms[mi], M - 1);
            const float* ep = fb_on ? fbias + (mc / HW) * (long)Cout : res + mc * Cout;
#pragma unroll
            for (int ni = 0; ni < NTW; ni++) ex[mi][ni] = *reinterpret_cast<const float4*>(ep + min(c0s[ni], Cout - 4));
          }
        }
#pragma unroll
        for (int mi = 0; mi < MB; mi++) {
#pragma unroll
          for (int ni = 0; ni < NTW; ni++) {
            const float4 v = quad_transpose(acc[m0 + mi][ni], q);
            float4 b4 = bv[ni];
            const float4 e4 = ex[mi][ni];
            if (fb_on) { b4.x += e4.x; b4.y += e4.y; b4.z += e4.z; b4.w += e4.w; }
            float4 o = make_float4(actf(v.x + b4.x), actf(v.y + b4.y), actf(v.z + b4.z), actf(v.w + b4.w));
            if (!fb_on && res_on) { o.x += e4.x; o.y += e4.y; o.z += e4.z; o.w += e4.w; }
            if (ni < nt && ms[mi] < M && c0s[ni] < Cout) *reinterpret_cast<float4*>(y + ms[mi] * Cout + c0s[ni]) = o;
          }
        }
      }
    };
    if (act >= kActHswish) epilogue([&](float v) { return act_slow(v, act); });
    else { const ClampK ck = clamp_of(act); epilogue([&](float v) { return clampf(v, ck); }); }
    return;
  }
  auto epilogue = [&](auto actf) {
#pragma unroll
    for (int mi = 0; mi < 2; mi++) {
      const long m = m_wave + 16 * mi + 4 * g + q;
#pragma unroll
      for (int ni = 0; ni < NTW; ni++) {
        if (ni >= nt) continue;
        const int c0 = n_base + 16 * ni + (li & ~3);
        const float4 v = quad_transpose(acc[mi][ni], q);
        if (m >= M || c0 >= Cout) continue;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 4 && c0 + e < Cout; e++) {
          float o = actf(vv[e] + bias[c0 + e] + (fbias ? fbias[(m / HW) * (long)Cout + c0 + e] : 0.f));
          if (res) o += res[m * Cout + c0 + e];
          y[m * Cout + c0 + e] = o;
        }
      }
    }
  };
  if (act >= kActHswish) epilogue([&](float v) { return act_slow(v, act); });
  else { const ClampK ck = clamp_of(act); epilogue([&](float v) { return clampf(v, ck); }); }
}

// (min 4 waves per SIMD: left alone the compiler takes 180 registers = 2 workgroups per CU, and the skinny-K expand layers — three K slabs
//  per tile — then spend their time waiting for the first slab: 128 registers fit without spills)
template <int TERMS, int NTW, bool IN16 = false>      // IN16: x holds f16 values (written by ir_expand_dw_k<.., OUT16>), TERMS == 1 only
__global__ __launch_bounds__(kThreads, NTW > 4 ? 3 : 4) void pw_gemm_f16s_k(const float* __restrict__ x, const _Float16* __restrict__ w16, const float* __restrict__ bias,
                                                          const float* __restrict__ res, const float* __restrict__ scale, const float* __restrict__ addx,
                                                          float* __restrict__ y, long M, int HW, int Cin, int Kp, int Cout, int cout_pad, int act,
                                                          const float* __restrict__ fbias, int dbg = 0) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[kGemmBM * kHSA];
  __shared__ __attribute__((aligned(16))) _Float16 Al[kGemmBM * kHSA];
  __shared__ __attribute__((aligned(16))) _Float16 Bh[NTW * 16 * kHSA];
  __shared__ __attribute__((aligned(16))) _Float16 Bl[NTW * 16 * kHSA];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  unsigned tcol; long trow;
  xcd_tile((unsigned)((Cout + NTW * 16 - 1) / (NTW * 16)), &tcol, &trow);      // columns >= Cout are padding: no tile for them
  const long m_base = trow * kGemmBM;
  const int n_base = (int)tcol * (NTW * 16);
  const int nt = min(NTW, (cout_pad - n_base) >> 4);
  const _Float16* wh = w16;
  const _Float16* wl = w16 + (size_t)cout_pad * Kp;
  // loader mapping: A = 128 rows x 8 float4 (4 per lane); B = 64 channels x 32 halves = 4 x 16-byte chunks per channel (1 per lane, hi and lo)
  constexpr int kBP = (NTW * 64 + kThreads - 1) / kThreads;       // 16-byte weight pieces per lane and K step
  float4 ra[4];
  h8v rbh[kBP], rbl[kBP];
  // Every A load is unconditional: rows past M read row M - 1 (their results are dropped), quads past Cin read the slab's first quad (any finite
  // value: their weights are zero) — as predicated loads each came with a zero fill, a select and a divergent branch.
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int f = tid + i * kThreads, row = f >> 3, kq = (f & 7) * 4;
      const long m = min((dbg & 1) ? (long)row : m_base + row, M - 1);
      const int k = k0 + kq < Cin ? k0 + kq : k0;
      float4 v;
      if (IN16) { const h4v hv = *reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(x) + m * Cin + k); v = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w); }
      else v = *reinterpret_cast<const float4*>(x + m * Cin + k);
      if (scale) {                                                         // (uniform)
        const float4 sv = *reinterpret_cast<const float4*>(scale + (m / HW) * (long)Cin + k);
        v.x = __fmul_rn(v.x, sv.x); v.y = __fmul_rn(v.y, sv.y); v.z = __fmul_rn(v.z, sv.z); v.w = __fmul_rn(v.w, sv.w);
      }
      if (addx) {
        const float4 av = *reinterpret_cast<const float4*>(addx + m * Cin + k);
        v.x = __fadd_rn(v.x, av.x); v.y = __fadd_rn(v.y, av.y); v.z = __fadd_rn(v.z, av.z); v.w = __fadd_rn(v.w, av.w);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < kBP; i++) {
      const int f = tid + i * kThreads, ch = f >> 2, kc = (f & 3) * 8;
      rbh[i] = h8v{0, 0, 0, 0, 0, 0, 0, 0}; rbl[i] = rbh[i];
      if (ch < NTW * 16 && n_base + ch < cout_pad) {
        rbh[i] = *reinterpret_cast<const h8v*>(wh + (size_t)(n_base + ch) * Kp + k0 + kc);
        if (TERMS == 3) rbl[i] = *reinterpret_cast<const h8v*>(wl + (size_t)(n_base + ch) * Kp + k0 + kc);
      }
    }
  };
  f4acc acc[2][NTW];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < NTW; ni++) acc[mi][ni] = f4acc{0.f, 0.f, 0.f, 0.f};
  const int gs = 8 * g;
  fetch(0);
  for (int k0 = 0; k0 < Kp; k0 += kGemmBK) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int f = tid + i * kThreads, row = f >> 3, kq = (f & 7) * 4;
      const float4 v = ra[i];
      u2v hi, lo;
      if (TERMS == 3) {
        unsigned a, b;
        split_pair(v.x, v.y, a, b); hi.x = a; lo.x = b;
        split_pair(v.z, v.w, a, b); hi.y = a; lo.y = b;
        *reinterpret_cast<u2v*>(&Al[row * kHSA + kq]) = lo;
      } else {
        hi.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y)); hi.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
      }
      *reinterpret_cast<u2v*>(&Ah[row * kHSA + kq]) = hi;
    }
#pragma unroll
    for (int i = 0; i < kBP; i++) {
      const int f = tid + i * kThreads, ch = f >> 2, kc = (f & 3) * 8;
      if (ch < NTW * 16) {
        *reinterpret_cast<h8v*>(&Bh[ch * kHSA + kc]) = rbh[i];
        if (TERMS == 3) *reinterpret_cast<h8v*>(&Bl[ch * kHSA + kc]) = rbl[i];
      }
    }
    __syncthreads();
    if (k0 + kGemmBK < Kp) fetch(k0 + kGemmBK);              // in flight while this tile is multiplied
    h8v ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++) {
      ah[mi] = *reinterpret_cast<const h8v*>(&Ah[(32 * wave + 16 * mi + li) * kHSA + gs]);
      if (TERMS == 3) al[mi] = *reinterpret_cast<const h8v*>(&Al[(32 * wave + 16 * mi + li) * kHSA + gs]);
    }
#pragma unroll
    for (int ni = 0; ni < NTW; ni++) {
      if (ni < nt) {
        const h8v bh = *reinterpret_cast<const h8v*>(&Bh[(16 * ni + li) * kHSA + gs]);
        h8v bl = bh;
        if (TERMS == 3) bl = *reinterpret_cast<const h8v*>(&Bl[(16 * ni + li) * kHSA + gs]);
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
          if (TERMS == 3) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  if (dbg & 2) { if (acc[0][0][0] != 12345.678f) return; }
  gemm_store_tile<NTW>(acc, m_base + 32 * wave, li, g, nt, n_base, M, HW, Cout, bias, fbias, res, y, act);
}

// ---- helpers of the LDS-DMA staged operand path (ir_expand_dw_k's STAGE form) -----------------------------------------------------------------------
// (Round 3's ring GEMM — both operands of pw_gemm_f16s_k delivered by global_load_lds rings, f32 activations only — never beat the register-staged kernel above
//  and was deleted in round 6 with its switch; its swizzles live on in the staged expand below.)
typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef const __attribute__((address_space(1))) void* glb_vp_t;
__device__ __forceinline__ int asw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

// ---- inverted-residual front half: expand 1x1 (+act) → depthwise 3x3 (+act) as ONE kernel ----------------------------------------------
// Workgroup = (frame, row band, chunk of CH expanded channels).  Phase 1: the chunk of the expanded tensor for the band's rows (+ the rows
// the 3x3 window reaches: [rows][W][CH] f32, 139 KB for a whole 33x33 frame at CH = 32) is formed in LDS by the split-f16 MFMA product of
// pw_gemm_f16s_k (same operand split, same term order; A rows come straight from global memory into registers — every A element feeds
// only the chunk's column tiles — and the chunk's weights stay in registers).  Phase 2: the depthwise runs from LDS — stride 1 (any
// dilation d) as the sliding-window column walk of dw_col_k (lane = (phase, column, channel quad) stepping d rows at a time, 3 new quads
// per output), stride 2 as nine taps per output — and writes its result.  The expanded tensor — per inverted-residual block the largest
// write AND the largest read — never reaches HBM: 4 of the block's 6 big tensor passes become 2.
constexpr int kIrThreads = 512, kIrSeg = 9;
template <int TERMS, int SLABS, int CH, bool OUT16 = false, int THREADS = kIrThreads>      // OUT16: the depthwise result is stored as f16 (reduced-precision storage mode); THREADS: 512, or 1024 (four waves per SIMD at <= 128 registers)
// (row-banded layers run TWO workgroups per CU = 4 waves per SIMD, i.e. within 128 registers: the 24-channel variant fits by itself, the 16-channel one is held to it)
__global__ __launch_bounds__(THREADS, (THREADS == 1024 || (SLABS == 1 && CH == 16)) ? 4 : 2) void ir_expand_dw_k(const float* __restrict__ x, const _Float16* __restrict__ w16, const float* __restrict__ bias,
                                                            const float* __restrict__ dww, const float* __restrict__ dwb, float* __restrict__ y,
                                                            int H, int W, int Cin, int Kp, int Cexp, int cout_pad, int act1, int act2, int d, int S, int pt, int pl,
                                                            int OH, int OW, int BH, int nbands, int phases) {
  extern __shared__ __attribute__((aligned(16))) float ir_ex[];          // [rows of the band][W][CH]
  constexpr int NT = (CH + 15) / 16, CQ = CH / 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4, q = li & 3;
  // XCD-aware order: all chunk workgroups of a (frame, band) run back to back on ONE XCD, so the band's input is fetched into that XCD's L2
  // once (in dispatch order they would spread over all eight L2s: the input crossed the fabric eight times — that, not arithmetic, bound phase 1)
  unsigned chunk; long rest;
  xcd_tile((unsigned)(Cexp / CH), &chunk, &rest);
  const long frame = rest / nbands;
  const int band = (int)(rest - frame * nbands);
  const int oy0 = band * BH, oy1 = min(oy0 + BH, OH);
  const int e0 = max(S * oy0 - pt, 0), e1 = min(S * (oy1 - 1) - pt + 2 * d + 1, H);      // expanded rows [e0, e1) of this band
  const int HWb = (e1 - e0) * W, n_base = (int)chunk * CH;
  const ClampK k1 = clamp_of(act1), k2 = clamp_of(act2);                   // the planner fuses clamp activations only (none / relu / relu6)
  const float* xf = x + ((size_t)frame * (size_t)H + (size_t)e0) * (size_t)W * Cin;
  float* yf = y + (size_t)frame * (size_t)OH * OW * Cexp;
  _Float16* yh = reinterpret_cast<_Float16*>(y) + (size_t)frame * (size_t)OH * OW * Cexp;      // (OUT16) the same tensor as packed halves: [n][OH*OW][Cexp]
  // ---- phase 1: expand into LDS
  if (phases & 1) {
    const _Float16* wh = w16;
    const _Float16* wl = w16 + (size_t)cout_pad * Kp;
    h8v bh[SLABS][NT], bl[SLABS][NT];
#pragma unroll
    for (int s = 0; s < SLABS; s++)
#pragma unroll
      for (int ni = 0; ni < NT; ni++) {
        bh[s][ni] = h8v{0, 0, 0, 0, 0, 0, 0, 0}; bl[s][ni] = bh[s][ni];
        if (16 * ni + li < CH) {                                           // CH = 24: the second column tile is half empty
          bh[s][ni] = *reinterpret_cast<const h8v*>(wh + (size_t)(n_base + 16 * ni + li) * Kp + 32 * s + 8 * g);
          if (TERMS == 3) bl[s][ni] = *reinterpret_cast<const h8v*>(wl + (size_t)(n_base + 16 * ni + li) * Kp + 32 * s + 8 * g);
        }
      }
    float bch[NT];                                                        // bias of this lane's channel (16 ni + li) of each column tile
#pragma unroll
    for (int ni = 0; ni < NT; ni++) bch[ni] = 16 * ni + li < CH ? bias[n_base + 16 * ni + li] : 0.f;
    const int ntile = (HWb + 15) >> 4, nw = THREADS >> 6;
    // A operands: straight from global memory (L2: the band's input is read by all of its chunk workgroups), kIrPf row tiles in
    // flight per wave — one workgroup owns the CU (its LDS), so nobody else hides a load that is waited for on the spot.
    constexpr int kIrPf = THREADS == 1024 ? 2 : 3;                                              // (5 / 6 tiles in flight, which the registers of a CU-owning workgroup allow, measured 3-6 % SLOWER)
    f4v ra[kIrPf][SLABS][2];
    // K tail (Cin % 32 != 0): the quads past Cin are read from the row's first quad instead — any finite value does, their weights are zero (the
    // planner pads w16 with zeros) — so that every load is unconditional: predicated, each tile carried 24 selects, the zero fill of its 24
    // registers and 11 divergent branches, a third of the phase's instructions.
    //
    // STAGE (32-channel chunks): the MFMA A layout wants lane (li, g) to hold 32 bytes of ROW li — sixteen different rows in sixteen consecutive
    // lanes, i.e. one cache-line lookup per lane: the phase ran at ~64 cycles per 1 KB load instruction (measured: time = 7k + 64 x instructions
    // cycles per workgroup for K = 32 / 64 / 96).  Instead eight consecutive lanes read one row's 128 bytes (8 lines per instruction, not 64),
    // each wave re-orders the tile through its own 2 KB of LDS (written at lane x 16 B, the chunk order XOR-swizzled on the GLOBAL side
    // (asw above), fragments read back with two conflict-free ds_read_b128) — no barrier: the LDS serves one wave's operations in order.
    constexpr bool STAGE = CH == 32 && THREADS == 512;              // (the 1024-lane form has no LDS left for sixteen 2 KB buffers)
    const int lr = lane >> 3, lp = lane & 7;
    int koff[SLABS][2];
#pragma unroll
    for (int s = 0; s < SLABS; s++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int k = STAGE ? 32 * s + 4 * (lp ^ asw(8 * h + lr)) : 32 * s + 8 * g + 4 * h;      // Cin % 4 == 0: each float4 is valid or absent
        koff[s][h] = k < Cin ? k : 0;
      }
    }
    auto fetch = [&](f4v (&dst)[SLABS][2], int rt) {
      if (STAGE) {                                                       // dst[s][h]: rows 8h .. 8h+7 of the tile, this lane's 16 bytes of row 8h + lr
        const float* r0 = xf + (size_t)min(rt * 16 + lr, HWb - 1) * Cin;
        const float* r1 = xf + (size_t)min(rt * 16 + 8 + lr, HWb - 1) * Cin;
#pragma unroll
        for (int s = 0; s < SLABS; s++) {
          dst[s][0] = *reinterpret_cast<const f4v*>(r0 + koff[s][0]);
          dst[s][1] = *reinterpret_cast<const f4v*>(r1 + koff[s][1]);
        }
      } else {
        const float* rp = xf + (size_t)min(rt * 16 + li, HWb - 1) * Cin;
#pragma unroll
        for (int s = 0; s < SLABS; s++) {
          dst[s][0] = *reinterpret_cast<const f4v*>(rp + koff[s][0]);
          dst[s][1] = *reinterpret_cast<const f4v*>(rp + koff[s][1]);
        }
      }
    };
    float* stg = ir_ex + (size_t)(e1 - e0) * W * CH + 10 * CH + 4 + wave * 512;      // this wave's [16 rows][8 quads] buffer (STAGE)
    const int st_w = lane * 4, st_r = li * 32 + (((2 * g) ^ asw(li)) << 2);           // written at lane x 16 B (+ 1 KB for the second row group); fragment quads 2g, 2g + 1
    // (Measured and dropped: an "L2 warm-up" — each chunk workgroup first requests its 1 / nchunk share of the band's input so that the tile loop
    //  runs on L2 hits — 3-5 % slower; 5 or 6 tiles in flight per wave instead of 3 — 3-6 % slower.  The phase is not waiting for memory.)
    const int cnt = wave < ntile ? (ntile - wave + nw - 1) / nw : 0;      // this wave's row tiles: wave, wave + nw, ...
#pragma unroll
    for (int j = 0; j < kIrPf; j++) if (j < cnt) fetch(ra[j], wave + j * nw);
    // One tile: re-order (STAGE) and split the A rows, refill the prefetch slot, multiply, store.  (FAST = no branch anywhere in the tile, so that a
    // group of kIrPf tiles is ONE basic block and the scheduler may run tile j + 1's LDS round trip and split under tile j's MFMA chain: measured,
    // no gain where it fit the registers — 1- and 2-slab variants — and 240 bytes of spill, 1.09 -> 1.68 ms, in the 3-slab one.  Not used.)
    auto tile = [&](auto FASTC, int j, int k) {
      constexpr bool FAST = decltype(FASTC)::value;
      const int rt = wave + k * nw;
      h8v ah[SLABS], al[SLABS];
#pragma unroll
      for (int s = 0; s < SLABS; s++) {
        if (STAGE) {
          *reinterpret_cast<f4v*>(stg + st_w) = ra[j][s][0];
          *reinterpret_cast<f4v*>(stg + 256 + st_w) = ra[j][s][1];
          const f4v v0 = *reinterpret_cast<const f4v*>(stg + st_r), v1 = *reinterpret_cast<const f4v*>(stg + (st_r ^ 4));
          split8<TERMS>(v0, v1, ah[s], al[s]);
        } else split8<TERMS>(ra[j][s][0], ra[j][s][1], ah[s], al[s]);
      }
      if (FAST || k + kIrPf < cnt) fetch(ra[j], wave + (k + kIrPf) * nw);       // refill the slot just consumed
      f4acc acc[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ni++) acc[ni] = f4acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SLABS; s++)
#pragma unroll
        for (int ni = 0; ni < NT; ni++) {
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bh[s][ni], acc[ni], 0, 0, 0);
          if (TERMS == 3) {
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], bh[s][ni], acc[ni], 0, 0, 0);
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bl[s][ni], acc[ni], 0, 0, 0);
          }
        }
      // MFMA C layout as it is — lane (li, g) holds channel 16 ni + li of pixels 4g .. 4g + 3 — stored as four 4-byte LDS writes per column tile
      // (16 lanes = 64 contiguous bytes of one pixel; the two pixels of a 32-lane store group are 2-way on the banks, which a 4-byte store
      // hides).  Transposed to one 16-byte write per pixel it was 32 more vector instructions per tile (DPP + selects) in an issue-bound
      // phase (PMC: the two waves of a SIMD keep its issue port 72 % busy) and a 4-way bank conflict on every store.
      const int pb = rt * 16 + 4 * g;
      const bool whole = FAST || rt * 16 + 16 <= HWb;                      // (uniform: only the band's last tile can be short)
#pragma unroll
      for (int ni = 0; ni < NT; ni++) {
        if (CH % 16 != 0 && 16 * ni + li >= CH) continue;                 // CH = 24: the second column tile is half empty
        float* dst = ir_ex + (size_t)pb * CH + 16 * ni + li;
        const float o0 = clampf(acc[ni][0] + bch[ni], k1), o1 = clampf(acc[ni][1] + bch[ni], k1), o2 = clampf(acc[ni][2] + bch[ni], k1),
                    o3 = clampf(acc[ni][3] + bch[ni], k1);
        if (whole) { dst[0] = o0; dst[CH] = o1; dst[2 * CH] = o2; dst[3 * CH] = o3; }
        else {
          if (pb < HWb) dst[0] = o0;
          if (pb + 1 < HWb) dst[CH] = o1;
          if (pb + 2 < HWb) dst[2 * CH] = o2;
          if (pb + 3 < HWb) dst[3 * CH] = o3;
        }
      }
    };
    for (int k0 = 0; k0 < cnt; k0 += kIrPf) {
#pragma unroll
      for (int j = 0; j < kIrPf; j++) {
        if (k0 + j >= cnt) break;                                         // wave-uniform
        tile(std::false_type{}, j, k0 + j);
      }
    }
  }
  // the depthwise weights + bias of the chunk: [9][CH] + [CH] behind the tile (a lane's channel quad changes from item to item when CH / 4 does not
  // divide the workgroup size, so registers cannot hold them; nine global loads per item were most of the stride-2 phase)
  float* dwl = ir_ex + (size_t)(e1 - e0) * W * CH;
  if (tid < 4) dwl[10 * CH + tid] = 0.f;                                 // the quad every row outside the image is read from (phase 2, stride 1)
  for (int i = tid; i < 10 * CQ; i += THREADS) {
    const int k = i / CQ, cq = i - k * CQ;
    *reinterpret_cast<f4v*>(dwl + k * CH + 4 * cq) = k < 9 ? *reinterpret_cast<const f4v*>(dww + (size_t)k * Cexp + n_base + 4 * cq)
                                                           : *reinterpret_cast<const f4v*>(dwb + n_base + 4 * cq);
  }
  __syncthreads();
  if (!(phases & 2)) return;
  const f4v zero = {0.f, 0.f, 0.f, 0.f};
  if (S == 1) {
    // ---- phase 2, stride 1 (SAME, dilation d).  Item = (segment, phase r, column, channel quad); a column is walked d rows at a time in
    // segments of at most kIrSeg outputs so that every dilation offers ~1000 items to the 512 lanes.  The phase is VALU-issue bound (PMC: 0.3
    // VALU instructions per wave quad-cycle, MFMA pipe 11 %), so the walk is written for instruction count: the 3-row window ROTATES BY NAME
    // (three steps per loop trip: no register moves — they were 12 of ~60 instructions per output), rows outside the image are read from a quad of
    // zeros in LDS through an address select (no branch, no zero-fill of the window), and the item's two divisions are multiplications.
    const int L = (oy1 - oy0 + d - 1) / d, nseg = (L + kIrSeg - 1) / kIrSeg, cols = d * W, total = nseg * cols * CQ;
    const unsigned mcols = 0xFFFFFFFFu / (unsigned)cols + 1u, mw = 0xFFFFFFFFu / (unsigned)W + 1u;      // t / cols, rc / W: exact below 2^16 (checked at launch)
    const float* zq = dwl + 10 * CH;
    // one unit of work: `nsteps` consecutive outputs of item `item` starting at its step `k0`
    auto walk = [&](int item, int k0, int nsteps) {
      const int t = item / CQ, cq = item - t * CQ, seg = (int)__umulhi((unsigned)t, mcols), rc = t - seg * cols,
                r = (int)__umulhi((unsigned)rc, mw), xx = rc - r * W;
      f4v wq[9];
#pragma unroll
      for (int k = 0; k < 9; k++) wq[k] = *reinterpret_cast<const f4v*>(dwl + k * CH + 4 * cq);
      const f4v bq = *reinterpret_cast<const f4v*>(dwl + 9 * CH + 4 * cq);
      // columns outside the image: their taps get ZERO WEIGHTS (the expanded values are finite, clamp-bounded) and a clamped address,
      // instead of a select per loaded quad in every step
      const bool vl = xx - d >= 0, vr = xx + d < W;
      if (!vl) { wq[0] = zero; wq[3] = zero; wq[6] = zero; }
      if (!vr) { wq[2] = zero; wq[5] = zero; wq[8] = zero; }
      const int xl = vl ? xx - d : xx, xr = vr ? xx + d : xx, rstep = d * W * CH;
      const float* col = ir_ex + (size_t)xx * CH + 4 * cq;
      const float* coll = ir_ex + (size_t)xl * CH + 4 * cq;
      const float* colr = ir_ex + (size_t)xr * CH + 4 * cq;
      auto row = [&](int yy, int off, f4v (&o)[3]) {                   // off = (yy - e0) * W * CH
        const bool in = yy >= 0 && yy < H;
        o[0] = *reinterpret_cast<const f4v*>(in ? coll + off : zq);
        o[1] = *reinterpret_cast<const f4v*>(in ? col + off : zq);
        o[2] = *reinterpret_cast<const f4v*>(in ? colr + off : zq);
      };
      int yy = oy0 + r + (seg * kIrSeg + k0) * d;
      int off = (yy - e0) * W * CH;
      float* yp = yf + ((size_t)yy * OW + xx) * Cexp + n_base + 4 * cq;
      const size_t ystep = (size_t)d * OW * Cexp;
      auto step = [&](const f4v (&p)[3], const f4v (&c)[3], f4v (&nx)[3]) {
        row(yy + d, off + rstep, nx);
        f4v acc = zero;
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(p[fx], wq[fx], acc);
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(c[fx], wq[3 + fx], acc);
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(nx[fx], wq[6 + fx], acc);
        acc += bq;
        if (OUT16) *reinterpret_cast<h4v*>(yh + (yp - yf)) = h4v{(_Float16)clampf(acc.x, k2), (_Float16)clampf(acc.y, k2), (_Float16)clampf(acc.z, k2), (_Float16)clampf(acc.w, k2)};
        else *reinterpret_cast<float4*>(yp) = make_float4(clampf(acc.x, k2), clampf(acc.y, k2), clampf(acc.z, k2), clampf(acc.w, k2));
        yy += d; off += rstep; yp += ystep;
      };
      f4v ra[3], rb[3], rc3[3];
      row(yy - d, off - rstep, ra); row(yy, off, rb);
      static_assert(kIrSeg % 3 == 0, "the window rotates by name in groups of three steps");
      for (int left = nsteps; left > 0 && yy < oy1;) {
        step(ra, rb, rc3);
        if (--left == 0 || yy >= oy1) break;
        step(rb, rc3, ra);
        if (--left == 0 || yy >= oy1) break;
        step(rc3, ra, rb);
        --left;
      }
    };
    // Full rounds of 512 items walk their whole segment.  The items left over (always 32 x CQ / 8 of them at 33 x 33: 1056 = 2 x 512 + 32) would keep
    // ONE wave busy for a third round while seven idle — a third of the phase — so they are cut into single outputs, one per lane (9 quads loaded
    // for one output instead of 3: a round and a half of work instead of a whole one).
    // (Only a SMALL remainder is cut up: at a quarter of a round or more the whole walks are cheaper — the 65 x 65 bands measured +16 % cut up.)
    const int rem = total % THREADS, main_total = rem * 4 <= THREADS ? total - rem : total;
    for (int item = tid; item < main_total; item += THREADS) walk(item, 0, kIrSeg);
    for (int u = tid; u < (total - main_total) * kIrSeg; u += THREADS) { const int li2 = u / kIrSeg; walk(main_total + li2, u - li2 * kIrSeg, 1); }
  } else {
    // ---- phase 2, stride 2 (dilation 1): item = (output pixel of the band, channel quad), nine taps from LDS, (fy, fx) ascending
    const int total = (oy1 - oy0) * OW * CQ;
    for (int item = tid; item < total; item += THREADS) {
      const int t = item / CQ, cq = item - t * CQ, oyl = t / OW, ox = t - oyl * OW, oy = oy0 + oyl;
      f4v acc = zero;
#pragma unroll
      for (int fy = 0; fy < 3; fy++) {
        const int iy = S * oy - pt + fy;
#pragma unroll
        for (int fx = 0; fx < 3; fx++) {
          const int ix = S * ox - pl + fx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const f4v xv = *reinterpret_cast<const f4v*>(ir_ex + ((size_t)(iy - e0) * W + ix) * CH + 4 * cq);
            const f4v wv = *reinterpret_cast<const f4v*>(dwl + (fy * 3 + fx) * CH + 4 * cq);
            acc = __builtin_elementwise_fma(xv, wv, acc);
          }
        }
      }
      acc += *reinterpret_cast<const f4v*>(dwl + 9 * CH + 4 * cq);
      const size_t oo = ((size_t)oy * OW + ox) * Cexp + n_base + 4 * cq;
      if (OUT16) *reinterpret_cast<h4v*>(yh + oo) = h4v{(_Float16)clampf(acc.x, k2), (_Float16)clampf(acc.y, k2), (_Float16)clampf(acc.z, k2), (_Float16)clampf(acc.w, k2)};
      else *reinterpret_cast<float4*>(yf + oo) = make_float4(clampf(acc.x, k2), clampf(acc.y, k2), clampf(acc.z, k2), clampf(acc.w, k2));
    }
  }
}

// (Round 3's whole-block kernel — expand → depthwise → project + residual in one launch for the layers with <= 16 input channels, the project accumulators
//  held across a loop over expanded-channel chunks — was parity-green and slower than this kernel + the project GEMM (spills at 128 registers, the block input
//  re-read once per chunk); it was deleted in round 6.  docs/design/08-rejected-and-next-r1-r4.md has its measurements.)

// ---- three chained 1x1 convolutions as ONE kernel (DeepLab's ASPP head: 160 → 256 relu → 256 relu, + the pooled branch as a per-frame bias → 21 classes) ---
// As three GEMMs the two 256-channel tensors between them are written and read back: 4.4 GB of the 5.3 GB those launches move at 1024 streams, for 0.7 GB of
// input and 0.09 GB of logits.  Here every product is formed TRANSPOSED — A = a 16-channel tile of the weights, B = 16 pixels of the activations, so the
// accumulator of lane (li, g) holds output channels 4g .. 4g+3 of pixel li — and that is, element for element, one half of the B operand the NEXT
// convolution wants from the same lane (pixel li, eight K values): two neighbouring 16-channel tiles give the eight halves of one 32-deep K slab once the slab's
// K order is agreed to be {4g .. 4g+3} ∪ {16+4g .. 16+4g+3} — a permutation the planner bakes into the packed weights (plan.cpp: pack_chain3).  The
// activations therefore go from accumulator to operand in registers (bias, clamp, the same hi/lo split as pw_gemm_f16s_k) and never touch LDS or HBM.
// A wave owns 16 x NP pixels and ALL channels of them; what the waves of a workgroup share is the WEIGHT stream (448 KB per pass, L2-resident): one
// 32 / 36 KB round at a time through a two-slot LDS ring, delivered by global→LDS DMA issued between the MFMAs of the round before, one barrier per round.
//   stage 1 (S0 rounds, one K slab each):   acc1[16 tiles] += W1 tile · X slab         → bias, clamp, split → y1 (the stage-2 B operands, P1 slabs)
//   stage 2 (P2 rounds, one output pair p): acc2[2 tiles] = Σ_s W2 tile(p, s) · y1[s]  → bias + per-frame bias, clamp, split → y2 = K slab p of stage 3
//                                           acc3[2 tiles] += W3 tile(·, p) · y2
// Same products as the three pw_gemm_f16s_k launches — same operand split, same term order (x_hi·w_hi, x_lo·w_hi, x_hi·w_lo), K slabs ascending; only the
// order of the 32 products INSIDE one MFMA differs (the slab's K permutation), i.e. rounding-level differences against the unchained path.
constexpr int kChainWaves = 4, kChainNP = 2;              // 128 pixels per workgroup, two workgroups per CU (72 KB of LDS, <= 256 registers)
template <int WAVES, int NP, int S0, int P1, int P2, bool DT = false>      // DT: all of the next round's DMA pieces at the round top instead of one per MFMA group (A/B, debug build)
__global__ __launch_bounds__(WAVES * 64, NP == 1 ? 4 : 2) void pw_chain3_k(const float* __restrict__ x, const _Float16* __restrict__ ws, const float* __restrict__ b1,
                                                                            const float* __restrict__ b2, const float* __restrict__ fb2, const float* __restrict__ b3,
                                                                            float* __restrict__ y, long M, int HW, int C3, int act1, int act2, int act3) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_ring[];
  constexpr int kTile = 1024;                              // bytes of one operand tile: 64 lanes x 8 halves, in lane order
  constexpr int kR1 = 4 * P1, kR2 = 4 * P1 + 4;            // tiles per round: stage 1 = (hi, lo) of the 2 P1 output tiles of one K slab; stage 2 = (hi, lo) x 2 tiles x P1 slabs + W3's 2 x (hi, lo)
  constexpr int kSlot = kR2 * kTile, kRounds = S0 + P2;
  constexpr int kC0 = 32 * S0, kC2 = 32 * P2;
  constexpr int kNJ = (kR2 + WAVES - 1) / WAVES;           // DMA pieces per wave and round
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const long px0 = ((long)blockIdx.x * WAVES + wave) * (16 * NP);
  int pxs[NP];                                             // (chain3_on() keeps M x 32 S0 below 2^31)
#pragma unroll
  for (int np = 0; np < NP; np++) pxs[np] = (int)min(px0 + 16 * np + li, M - 1);      // pixels past the end compute on the last one; their stores are dropped
  const ClampK k1 = clamp_of(act1), k2 = clamp_of(act2), k3 = clamp_of(act3);
  // piece j of this wave for round r (wave-uniform LDS base + lane x 16 B; the stream is stored in exactly that order)
  auto dma = [&](int r, int j) {
    const int t = wave + j * WAVES, nt = r < S0 ? kR1 : kR2;
    if (r < kRounds && t < nt) {
      const long t0 = r < S0 ? (long)r * kR1 : (long)S0 * kR1 + (long)(r - S0) * kR2;
      __builtin_amdgcn_global_load_lds((glb_vp_t)(ws + (t0 + t) * 512 + lane * 8), (lds_vp_t)(ch_ring + (r & 1) * kSlot + t * kTile), 16, 0, 0);
    }
  };
  auto mfma3 = [&](f4acc& acc, const h8v wh, const h8v wl, const h8v xh, const h8v xl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < kNJ; j++) dma(0, j);
  // ---- stage 1
  h8v y1h[NP][P1], y1l[NP][P1];
  {
    f4acc acc1[NP][2 * P1];
#pragma unroll
    for (int np = 0; np < NP; np++)
#pragma unroll
      for (int t = 0; t < 2 * P1; t++) acc1[np][t] = f4acc{0.f, 0.f, 0.f, 0.f};
    f4v xr[NP][2];
#pragma unroll
    for (int np = 0; np < NP; np++) {
      xr[np][0] = *reinterpret_cast<const f4v*>(x + pxs[np] * kC0 + 8 * g);
      xr[np][1] = *reinterpret_cast<const f4v*>(x + pxs[np] * kC0 + 8 * g + 4);
    }
    for (int s = 0; s < S0; s++) {
      __syncthreads();                                     // (vmcnt(0) first) round s is in its slot for every wave, and nobody still reads the other slot
      h8v xh[NP], xl[NP];
#pragma unroll
      for (int np = 0; np < NP; np++) split8<3>(xr[np][0], xr[np][1], xh[np], xl[np]);
      if (s + 1 < S0) {
#pragma unroll
        for (int np = 0; np < NP; np++) {
          xr[np][0] = *reinterpret_cast<const f4v*>(x + pxs[np] * kC0 + 32 * (s + 1) + 8 * g);
          xr[np][1] = *reinterpret_cast<const f4v*>(x + pxs[np] * kC0 + 32 * (s + 1) + 8 * g + 4);
        }
      }
      const unsigned char* slot = ch_ring + (s & 1) * kSlot + lane * 16;
      // operand tiles one group ahead of the MFMAs that consume them; next round's DMA pieces one per group.  (A DMA between LDS reads makes hipcc wait for ALL
      // outstanding reads — lgkmcnt(0) — at the next use of any of them; issuing the round's pieces together at the top avoids that and measured 4 % SLOWER:
      // 636 vs 612 us, profiles/r06ab — nine back-to-back DMA issues stall the wave longer than the waits they remove.)
      if (DT) {
#pragma unroll
        for (int j = 0; j < kNJ; j++) dma(s + 1, j);
      }
      h8v wh = *reinterpret_cast<const h8v*>(slot), wl = *reinterpret_cast<const h8v*>(slot + kTile);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int t = 0; t < 2 * P1; t++) {
        if (!DT && t < kNJ) dma(s + 1, t);
        h8v nh = wh, nl = wl;
        if (t + 1 < 2 * P1) {
          nh = *reinterpret_cast<const h8v*>(slot + (2 * (t + 1)) * kTile);
          nl = *reinterpret_cast<const h8v*>(slot + (2 * (t + 1) + 1) * kTile);
        }
#pragma unroll
        for (int np = 0; np < NP; np++) acc1[np][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[np], acc1[np][t], 0, 0, 0);
#pragma unroll
        for (int np = 0; np < NP; np++) acc1[np][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[np], acc1[np][t], 0, 0, 0);
#pragma unroll
        for (int np = 0; np < NP; np++) acc1[np][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[np], acc1[np][t], 0, 0, 0);
        if (t + 1 < 2 * P1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NP, 0);
        wh = nh; wl = nl;
      }
    }
#pragma unroll
    for (int p = 0; p < P1; p++) {
      const float4 ba = *reinterpret_cast<const float4*>(b1 + 32 * p + 4 * g), bb = *reinterpret_cast<const float4*>(b1 + 32 * p + 16 + 4 * g);
#pragma unroll
      for (int np = 0; np < NP; np++) {
        const f4acc a = acc1[np][2 * p], b = acc1[np][2 * p + 1];
        const f4v va{clampf(a[0] + ba.x, k1), clampf(a[1] + ba.y, k1), clampf(a[2] + ba.z, k1), clampf(a[3] + ba.w, k1)};
        const f4v vb{clampf(b[0] + bb.x, k1), clampf(b[1] + bb.y, k1), clampf(b[2] + bb.z, k1), clampf(b[3] + bb.w, k1)};
        split8<3>(va, vb, y1h[np][p], y1l[np][p]);
      }
    }
  }
  // ---- stages 2 and 3
  f4acc acc3[NP][2];
#pragma unroll
  for (int np = 0; np < NP; np++) acc3[np][0] = acc3[np][1] = f4acc{0.f, 0.f, 0.f, 0.f};
  int frm[NP];
#pragma unroll
  for (int np = 0; np < NP; np++) frm[np] = fb2 ? (pxs[np] / HW) * kC2 : 0;
  for (int p = 0; p < P2; p++) {
    const int r = S0 + p;
    __syncthreads();
    float4 ba = *reinterpret_cast<const float4*>(b2 + 32 * p + 4 * g), bb = *reinterpret_cast<const float4*>(b2 + 32 * p + 16 + 4 * g);
    float4 fa[NP], fb[NP];
#pragma unroll
    for (int np = 0; np < NP; np++) { fa[np] = make_float4(0.f, 0.f, 0.f, 0.f); fb[np] = fa[np]; }
    if (fb2) {                                             // (uniform)
#pragma unroll
      for (int np = 0; np < NP; np++) {
        fa[np] = *reinterpret_cast<const float4*>(fb2 + frm[np] + 32 * p + 4 * g);
        fb[np] = *reinterpret_cast<const float4*>(fb2 + frm[np] + 32 * p + 16 + 4 * g);
      }
    }
    const unsigned char* slot = ch_ring + (r & 1) * kSlot + lane * 16;
    f4acc acc2[NP][2];
#pragma unroll
    for (int np = 0; np < NP; np++) acc2[np][0] = acc2[np][1] = f4acc{0.f, 0.f, 0.f, 0.f};
    // operand tiles one group ahead: the (hi, lo) pair of group i + 1 is requested before the six MFMAs of group i are issued — left to itself the compiler (at its
    // register limit) requests a pair two MFMAs before it waits for it, and a wave stalls for most of an LDS round trip per group (measured: 45 % of the MFMA peak)
    if (DT) {
#pragma unroll
      for (int j = 0; j < kNJ; j++) dma(r + 1, j);
    }
    h8v wh = *reinterpret_cast<const h8v*>(slot), wl = *reinterpret_cast<const h8v*>(slot + kTile);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    h8v w3h[2], w3l[2];
#pragma unroll
    for (int i = 0; i < 2 * P1; i++) {
      const int s = i >> 1, h = i & 1;
      if (!DT && i < kNJ) dma(r + 1, i);
      h8v nh = wh, nl = wl;
      if (i + 1 < 2 * P1) {
        nh = *reinterpret_cast<const h8v*>(slot + (2 * (i + 1)) * kTile);
        nl = *reinterpret_cast<const h8v*>(slot + (2 * (i + 1) + 1) * kTile);
      } else {
#pragma unroll
        for (int o = 0; o < 2; o++) {
          w3h[o] = *reinterpret_cast<const h8v*>(slot + (4 * P1 + 2 * o) * kTile);
          w3l[o] = *reinterpret_cast<const h8v*>(slot + (4 * P1 + 2 * o + 1) * kTile);
        }
      }
#pragma unroll
      for (int np = 0; np < NP; np++) acc2[np][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, y1h[np][s], acc2[np][h], 0, 0, 0);
#pragma unroll
      for (int np = 0; np < NP; np++) acc2[np][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, y1l[np][s], acc2[np][h], 0, 0, 0);
#pragma unroll
      for (int np = 0; np < NP; np++) acc2[np][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, y1h[np][s], acc2[np][h], 0, 0, 0);
      if (i + 1 < 2 * P1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NP, 0);
      wh = nh; wl = nl;
    }
#pragma unroll
    for (int np = 0; np < NP; np++) {
      // (the per-frame vector is added to the bias first, then the sum to the accumulator: gemm_store_tile's order)
      const float4 sa = fb2 ? make_float4(ba.x + fa[np].x, ba.y + fa[np].y, ba.z + fa[np].z, ba.w + fa[np].w) : ba;
      const float4 sb = fb2 ? make_float4(bb.x + fb[np].x, bb.y + fb[np].y, bb.z + fb[np].z, bb.w + fb[np].w) : bb;
      const f4acc a = acc2[np][0], b = acc2[np][1];
      const f4v va{clampf(a[0] + sa.x, k2), clampf(a[1] + sa.y, k2), clampf(a[2] + sa.z, k2), clampf(a[3] + sa.w, k2)};
      const f4v vb{clampf(b[0] + sb.x, k2), clampf(b[1] + sb.y, k2), clampf(b[2] + sb.z, k2), clampf(b[3] + sb.w, k2)};
      h8v y2h, y2l;
      split8<3>(va, vb, y2h, y2l);
#pragma unroll
      for (int o = 0; o < 2; o++) mfma3(acc3[np][o], w3h[o], w3l[o], y2h, y2l);
    }
  }
  // ---- logits: lane (li, g) holds classes 16 o + 4 g .. + 3 of pixel li
#pragma unroll
  for (int np = 0; np < NP; np++) {
    const long px = px0 + 16 * np + li;
#pragma unroll
    for (int o = 0; o < 2; o++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int cls = 16 * o + 4 * g + j;
        if (px < M && cls < C3) y[px * C3 + cls] = clampf(acc3[np][o][j] + b3[cls], k3);
      }
    }
  }
}

// ---- DeepLab's first three layers in one kernel: stem conv 3x3/s2 (3 → 16) → depthwise 3x3 → 1x1 (16 → C2 <= 16) ------------------------
// Workgroup = (frame, band of BH output rows).  The input rows the band needs, the stem's band (+1 halo row each side: SAME padding of the
// depthwise = zero rows outside the image) and the depthwise's band live in LDS; only the C2-channel result is written.  Unfused, the two
// 16-channel tensors at stem resolution (1 MB per frame each at 129x129) were written and read back: 5.6 MB of traffic per frame for 0.8 MB
// of input and 0.5 MB of output.  Plain f32 FMAs (3 / 16 input channels: an MFMA slab would be mostly padding), (fy, fx, ci) ascending, bias last.
// U8IN: x is the 8-bit network input (one u32 R | G<<8 | B<<16 per pixel, prep_fused_k<2>), normalised while it is staged with convertTo's two
// roundings fadd(fmul(float(q), in_scale), in_offset) (libbackscrub.cc:302): a quarter of the bytes of the f32 tensor, the same values bit for bit.
constexpr int kH0Threads = 512;
template <bool U8IN>
__global__ __launch_bounds__(kH0Threads) void dl_head0_k(const float* __restrict__ x, const float* __restrict__ ws, const float* __restrict__ bs,
                                                        const float* __restrict__ wd, const float* __restrict__ bd, const float* __restrict__ wp,
                                                        const float* __restrict__ bp, float* __restrict__ y, int H0, int W0, int H1, int W1, int pt, int pl,
                                                        int C2, int pw_cout_pad, int act_s, int act_d, int act_p, int BH, int nbands, int phases, float in_scale,
                                                        float in_offset, int n_frames) {
  extern __shared__ __attribute__((aligned(16))) float h0_lds[];
  const int tid = threadIdx.x;
  unsigned f_, t_;
  xcd_frame_tile((unsigned)nbands, (unsigned)n_frames, &f_, &t_);              // a frame's bands on one XCD: the halo rows two bands share are L2 hits
  const long frame = f_;
  const int band = (int)t_;
  const int oy0 = band * BH, oy1 = min(oy0 + BH, H1), SR = oy1 - oy0 + 2, sy0 = oy0 - 1;      // stem rows [sy0, sy0 + SR)
  const int IR = 2 * SR + 1, iy0 = 2 * sy0 - pt, rowf = (W0 + 2) * 3;                          // input rows [iy0, iy0 + IR), one zero pixel left and right
  float* in_t = h0_lds;                                         // [IR][W0 + 2][3]
  float* D = h0_lds;                                             // [BH][W1][16]: over the input rows, which are dead once the stem has run
  const int r0f = ((2 * (BH + 2) + 1) * rowf + 3) & ~3, r0d = BH * W1 * 16;
  float* S = h0_lds + (r0f > r0d ? r0f : r0d);                   // [SR][W1][16]
  float* wl = S + (BH + 2) * W1 * 16;                            // stem [27][16] + bias 16 | dw [9][16] + bias 16 | pw [16][16] + bias 16
  const ClampK ks = clamp_of(act_s), kd = clamp_of(act_d), kp = clamp_of(act_p);
  // ---- weights and the input rows
  for (int i = tid; i < 27 * 16 + 16 + 9 * 16 + 16 + 16 * 16 + 16; i += kH0Threads) {
    float v;
    if (i < 432) v = ws[i];
    else if (i < 448) v = bs[i - 432];
    else if (i < 592) v = wd[i - 448];
    else if (i < 608) v = bd[i - 592];
    else if (i < 864) { const int k = (i - 608) >> 4, co = (i - 608) & 15; v = co < pw_cout_pad ? wp[k * pw_cout_pad + co] : 0.f; }
    else v = (i - 864) < pw_cout_pad ? bp[i - 864] : 0.f;
    wl[i] = v;
  }
  const float* xf = x + (size_t)frame * (size_t)H0 * W0 * 3;
  if (U8IN && (phases & 1)) {
    // the band's rows as ONE flat stream of pixels (u32 each), four per 16-byte load; in_t pixel j = image column j - 1
    const uint32_t* xu = reinterpret_cast<const uint32_t*>(x) + (size_t)frame * (size_t)H0 * W0;
    const int gy_lo = max(iy0, 0), gy_hi = min(iy0 + IR, H0);
    const int r_off = gy_lo - iy0, np = max(gy_hi - gy_lo, 0) * W0, nq = np >> 2;
    const uint32_t* src = xu + (size_t)gy_lo * W0;
    constexpr int kMaxQ = 3;                                              // 21 rows x 257 pixels / 4 / 512 lanes
    u4v v[kMaxQ];
#pragma unroll
    for (int k = 0; k < kMaxQ; k++) {
      const int i = tid + k * kH0Threads;
      v[k] = *reinterpret_cast<const u4v*>(src + 4 * (size_t)min(i, max(nq - 1, 0)));
    }
    uint32_t tailv = 0u;
    if (tid < (np & 3)) tailv = src[4 * nq + tid];
    for (int i = tid; i < IR * 6; i += kH0Threads) { const int r = i / 6, e = i - 6 * r; in_t[r * rowf + (e < 3 ? e : rowf - 6 + e)] = 0.f; }
    for (int r = 0; r < IR; r++) {
      if (r >= r_off && r < r_off + (gy_hi - gy_lo)) continue;          // (uniform)
      for (int e = tid; e < W0 * 3; e += kH0Threads) in_t[r * rowf + 3 + e] = 0.f;
    }
    const unsigned gmagic = 0xFFFFFFFFu / (unsigned)W0 + 1u;              // p / W0 for p < 2^16
    auto put = [&](int p, uint32_t px) {
      const int r = (int)__umulhi((unsigned)p, gmagic), e = p - r * W0;
      float* o = in_t + (r_off + r) * rowf + 3 + 3 * e;
      o[0] = __fadd_rn(__fmul_rn((float)(px & 255u), in_scale), in_offset);
      o[1] = __fadd_rn(__fmul_rn((float)((px >> 8) & 255u), in_scale), in_offset);
      o[2] = __fadd_rn(__fmul_rn((float)((px >> 16) & 255u), in_scale), in_offset);
    };
#pragma unroll
    for (int k = 0; k < kMaxQ; k++) {
      const int i = tid + k * kH0Threads;
      if (i < nq) {
#pragma unroll
        for (int c = 0; c < 4; c++) put(4 * i + c, v[k][c]);
      }
    }
    if (tid < (np & 3)) put(4 * nq + tid, tailv);
  }
  if (!U8IN && (phases & 1)) {
    // The band's input rows are ONE contiguous piece of the frame (rows iy0 .. iy0 + IR - 1, clipped to the image: up to 21 x 771 floats): read it as
    // a flat stream of 16-byte quads — all of a lane's quads requested before the first LDS store — and scatter the floats to their (row, column)
    // in LDS.  (Row by row with one float per lane it took 22 dword loads per lane: 477 us of the kernel's 1024 just to bring the input in.)
    // in_t pixel j = image column j - 1: a row's floats land at r * rowf + 3 + e; the pad pixels and the rows outside the image are zero-filled.
    const int rowg = W0 * 3;
    const int gy_lo = max(iy0, 0), gy_hi = min(iy0 + IR, H0);              // valid image rows [gy_lo, gy_hi)
    const int r_off = gy_lo - iy0, nf = max(gy_hi - gy_lo, 0) * rowg, nq = nf >> 2;
    const float* src = xf + (size_t)gy_lo * rowg;                         // (4-byte aligned only: the hardware takes dword-aligned dwordx4 loads)
    constexpr int kMaxQ = 8;                                              // 21 rows x 771 floats / 4 / 512 lanes
    f4v v[kMaxQ];
#pragma unroll
    for (int k = 0; k < kMaxQ; k++) {
      const int i = tid + k * kH0Threads;
      v[k] = *reinterpret_cast<const f4v*>(src + 4 * (size_t)min(i, max(nq - 1, 0)));
    }
    float tailv = 0.f;
    if (tid < (nf & 3)) tailv = src[4 * nq + tid];
    // zero fill: pad pixels of every row, whole rows outside the image
    for (int i = tid; i < IR * 6; i += kH0Threads) { const int r = i / 6, e = i - 6 * r; in_t[r * rowf + (e < 3 ? e : rowf - 6 + e)] = 0.f; }
    for (int r = 0; r < IR; r++) {
      if (r >= r_off && r < r_off + (gy_hi - gy_lo)) continue;          // (uniform)
      for (int e = tid; e < rowg; e += kH0Threads) in_t[r * rowf + 3 + e] = 0.f;
    }
    const unsigned gmagic = 0xFFFFFFFFu / (unsigned)rowg + 1u;            // f / rowg for f < 2^16
#pragma unroll
    for (int k = 0; k < kMaxQ; k++) {
      const int i = tid + k * kH0Threads;
      if (i < nq) {
        const int f = 4 * i;
        int r = (int)__umulhi((unsigned)f, gmagic), e = f - r * rowg;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          in_t[(r_off + r) * rowf + 3 + e] = v[k][c];
          if (++e == rowg) { e = 0; r++; }
        }
      }
    }
    if (tid < (nf & 3)) { const int f = 4 * nq + tid, r = (int)__umulhi((unsigned)f, gmagic), e = f - r * rowg; in_t[(r_off + r) * rowf + 3 + e] = tailv; }
  }
  __syncthreads();
  // ---- stem on the matrix cores: v_mfma_f32_16x16x4_f32 over the im2col axis k = (fy, fx, ci) (27 of 28 slots used), wave = one tile of 16
  // pixels of a row x 16 channels; A comes from the LDS input rows through a per-lane offset table, B (the weights) stays in registers.
  // (The VALU form — lane = (column, channel quad), rows in registers — ran at 20 % of the vector peak: 270 LDS reads per lane.)
  if (phases & 2) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q = li & 3, c0 = li & ~3;
    int koff[7];
    float wsr[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
      const int k = 4 * t + g;
      const bool valid = k < 27;
      const int fy = k / 9, r9 = k - 9 * fy, fx = r9 / 3, ci = r9 - 3 * fx;
      koff[t] = valid ? fy * rowf + fx * 3 + ci : 0;
      wsr[t] = valid ? wl[k * 16 + li] : 0.f;
    }
    const float4 bias4 = *reinterpret_cast<const float4*>(wl + 432 + c0);
    const int ctiles = (W1 + 15) >> 4, ntile = SR * ctiles;
    for (int t = wave; t < ntile; t += 2 * (kH0Threads >> 6)) {          // two tiles per iteration: their dependent MFMA chains interleave
      const int t1 = t + (kH0Threads >> 6);
      const bool two = t1 < ntile;
      const int ra = t / ctiles, ca = t - ra * ctiles, rb = two ? t1 / ctiles : ra, cb = two ? t1 - rb * ctiles : ca;
      const float* ba = in_t + (2 * ra) * rowf + (2 * min(16 * ca + li, W1 - 1) - pl + 1) * 3;
      const float* bb = in_t + (2 * rb) * rowf + (2 * min(16 * cb + li, W1 - 1) - pl + 1) * 3;
      float va[7], vb[7];
#pragma unroll
      for (int u = 0; u < 7; u++) { va[u] = ba[koff[u]]; vb[u] = bb[koff[u]]; }
      f4acc acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 7; u++) {
        acca = __builtin_amdgcn_mfma_f32_16x16x4f32(va[u], wsr[u], acca, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[u], wsr[u], accb, 0, 0, 0);
      }
      auto put = [&](const f4acc acc, int row, int ct) {
        const float4 v = quad_transpose(acc, q);
        const int px = 16 * ct + 4 * g + q, sy = sy0 + row;
        if (px < W1) {
          float4 o = make_float4(clampf(v.x + bias4.x, ks), clampf(v.y + bias4.y, ks), clampf(v.z + bias4.z, ks), clampf(v.w + bias4.w, ks));
          if (sy < 0 || sy >= H1) o = make_float4(0.f, 0.f, 0.f, 0.f);     // rows outside the image are the depthwise's zero padding
          *reinterpret_cast<float4*>(S + ((size_t)row * W1 + px) * 16 + c0) = o;
        }
      };
      put(acca, ra, ca);
      if (two) put(accb, rb, cb);
    }
  }
  __syncthreads();
  // ---- depthwise 3x3 (SAME): lane = (column, channel quad) walks the band's rows with a 3-row window
  if (phases & 4) {
    const float* wdw = wl + 448;
    for (int item = tid; item < W1 * 4; item += kH0Threads) {
      const int xx = item >> 2, cq = item & 3;
      f4v wq[9];
#pragma unroll
      for (int k = 0; k < 9; k++) wq[k] = *reinterpret_cast<const f4v*>(wdw + k * 16 + 4 * cq);
      const f4v bq = *reinterpret_cast<const f4v*>(wdw + 144 + 4 * cq);
      const f4v zero = {0.f, 0.f, 0.f, 0.f};
      const bool vl = xx >= 1, vr = xx + 1 < W1;
      if (!vl) { wq[0] = zero; wq[3] = zero; wq[6] = zero; }
      if (!vr) { wq[2] = zero; wq[5] = zero; wq[8] = zero; }
      const float* col = S + (size_t)xx * 16 + 4 * cq;
      const int dl = vl ? -16 : 0, dr = vr ? 16 : 0, rs = W1 * 16;
      auto row = [&](int r, f4v (&o)[3]) { o[0] = *reinterpret_cast<const f4v*>(col + r * rs + dl); o[1] = *reinterpret_cast<const f4v*>(col + r * rs); o[2] = *reinterpret_cast<const f4v*>(col + r * rs + dr); };
      // (the 3-row window rotates by name, three steps per loop trip: no register moves)
      const int nr = oy1 - oy0;
      int r = 0;
      auto step = [&](const f4v (&p)[3], const f4v (&c)[3], f4v (&nx)[3]) {
        row(r + 2, nx);
        f4v acc = zero;
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(p[fx], wq[fx], acc);
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(c[fx], wq[3 + fx], acc);
#pragma unroll
        for (int fx = 0; fx < 3; fx++) acc = __builtin_elementwise_fma(nx[fx], wq[6 + fx], acc);
        acc += bq;
        *reinterpret_cast<float4*>(D + ((size_t)r * W1 + xx) * 16 + 4 * cq) = make_float4(clampf(acc.x, kd), clampf(acc.y, kd), clampf(acc.z, kd), clampf(acc.w, kd));
        r++;
      };
      f4v ra[3], rb[3], rc[3];
      row(0, ra); row(1, rb);
      while (r < nr) {
        step(ra, rb, rc);
        if (r >= nr) break;
        step(rb, rc, ra);
        if (r >= nr) break;
        step(rc, ra, rb);
      }
    }
  }
  __syncthreads();
  // ---- 1x1 (16 → C2): lane = (pixel of the band, output quad), ci ascending, bias last
  if (phases & 8) {
    const float* wpw = wl + 608;
    const int CQ2 = C2 >> 2, total = (oy1 - oy0) * W1 * CQ2;
    float* yf = y + ((size_t)frame * H1 + oy0) * (size_t)W1 * C2;
    for (int item = tid; item < total; item += kH0Threads) {
      const int pix = item / CQ2, cq = item - pix * CQ2;
      f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++) {
        const f4v dv = *reinterpret_cast<const f4v*>(D + (size_t)pix * 16 + 4 * k4);
        acc = __builtin_elementwise_fma((f4v)(dv.x), *reinterpret_cast<const f4v*>(wpw + (4 * k4) * 16 + 4 * cq), acc);
        acc = __builtin_elementwise_fma((f4v)(dv.y), *reinterpret_cast<const f4v*>(wpw + (4 * k4 + 1) * 16 + 4 * cq), acc);
        acc = __builtin_elementwise_fma((f4v)(dv.z), *reinterpret_cast<const f4v*>(wpw + (4 * k4 + 2) * 16 + 4 * cq), acc);
        acc = __builtin_elementwise_fma((f4v)(dv.w), *reinterpret_cast<const f4v*>(wpw + (4 * k4 + 3) * 16 + 4 * cq), acc);
      }
      acc += *reinterpret_cast<const f4v*>(wpw + 256 + 4 * cq);
      *reinterpret_cast<float4*>(yf + (size_t)pix * C2 + 4 * cq) = make_float4(clampf(acc.x, kp), clampf(acc.y, kp), clampf(acc.z, kp), clampf(acc.w, kp));
    }
  }
}

// Few pixels (squeeze-excite / gate FCs: one "pixel" per stream): lane = one output value, so a
// 256-stream batch still fills thousands of lanes instead of one workgroup.
__global__ __launch_bounds__(kThreads) void pw_small_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ res, const float* __restrict__ scale, const float* __restrict__ addx,
                                                      float* __restrict__ y, long total, int HW, int Cin, int Cout, int cout_pad, int act,
                                                      const float* __restrict__ fbias = nullptr) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  int co = (int)(i % Cout);
  long p = i / Cout;
  const float* xp = x + p * Cin;
  const float* sp = scale ? scale + (p / HW) * (long)Cin : nullptr;
  float acc = 0.f;
  int ci = 0;
  if (!sp && !addx) {
    // plain GEMV (the per-frame FC / pool-branch steps): eight weight loads in flight per lane — one load per iteration made the
    // 256-long dot products of DeepLab's ASPP cost one L2 round trip per element (63-98 us for 17 MFLOP)
    for (; ci + 8 <= Cin; ci += 8) {
      float wv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { wv[u] = w[(long)(ci + u) * cout_pad + co]; xv[u] = xp[ci + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) acc = fmaf(xv[u], wv[u], acc);
    }
  }
  for (; ci < Cin; ci++) {
    float xv = xp[ci];
    if (sp) xv = __fmul_rn(xv, sp[ci]);
    if (addx) xv = __fadd_rn(xv, addx[p * Cin + ci]);
    acc = fmaf(xv, w[(long)ci * cout_pad + co], acc);
  }
  float bb = bias[co];
  if (fbias) bb += fbias[(p / HW) * (long)Cout + co];        // per-frame bias vector (a broadcast concat branch folded into this conv)
  float v = act_fn(acc + bb, act);
  if (res) v += res[i];
  y[i] = v;
}

// -------------------------------------------------------------------------------------
// general dense convolution (stem 3x3 s2, anything that is not 1x1 s1): lane = output pixel
// -------------------------------------------------------------------------------------
struct ConvGeom { int H, W, Cin, OH, OW, Cout, cout_pad, kh, kw, sh, sw, dh, dw, pt, pl; };

template <int CT>
__global__ __launch_bounds__(kThreads) void conv_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  const float* __restrict__ res, float* __restrict__ y, long M, ConvGeom g, int act) {
  long p = (long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= M) return;
  const int co0 = blockIdx.y * CT;
  int ox = (int)(p % g.OW);
  long q = p / g.OW;
  int oy = (int)(q % g.OH);
  long n = q / g.OH;
  float acc[CT];
#pragma unroll
  for (int t = 0; t < CT; t++) acc[t] = 0.f;
  for (int fy = 0; fy < g.kh; fy++) {
    int iy = oy * g.sh - g.pt + fy * g.dh;
    if (iy < 0 || iy >= g.H) continue;
    for (int fx = 0; fx < g.kw; fx++) {
      int ix = ox * g.sw - g.pl + fx * g.dw;
      if (ix < 0 || ix >= g.W) continue;
      const float* xp = x + ((n * g.H + iy) * g.W + ix) * (long)g.Cin;
      const float* w0 = w + (long)(fy * g.kw + fx) * g.Cin * g.cout_pad + co0;
      for (int ci = 0; ci < g.Cin; ci++) {
        float xv = xp[ci];
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = fmaf(xv, w0[(long)ci * g.cout_pad + t], acc[t]);
      }
    }
  }
  float* yp = y + p * g.Cout + co0;
  const float* rp = res ? res + p * g.Cout + co0 : nullptr;
  if ((g.Cout & 3) == 0 && co0 + CT <= g.Cout) {                 // 16-byte stores: a lane's CT outputs are contiguous (scalar stores touched 64 lines per instruction)
#pragma unroll
    for (int t = 0; t < CT; t += 4) {
      float4 v = make_float4(act_fn(acc[t] + bias[co0 + t], act), act_fn(acc[t + 1] + bias[co0 + t + 1], act), act_fn(acc[t + 2] + bias[co0 + t + 2], act),
                             act_fn(acc[t + 3] + bias[co0 + t + 3], act));
      if (rp) { const float4 r = *reinterpret_cast<const float4*>(rp + t); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      *reinterpret_cast<float4*>(yp + t) = v;
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < CT; t++) {
    if (co0 + t < g.Cout) {
      float v = act_fn(acc[t] + bias[co0 + t], act);
      if (rp) v += rp[t];
      yp[t] = v;
    }
  }
}

// -------------------------------------------------------------------------------------
// depthwise convolution: lane = (pixel, channel quad)
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void dw_conv_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ res, float* __restrict__ y, long total, ConvGeom g, int act) {
  long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int C4 = g.Cin >> 2;
  int c = (int)(idx % C4) * 4;
  long p = idx / C4;
  int ox = (int)(p % g.OW);
  long q = p / g.OW;
  int oy = (int)(q % g.OH);
  long n = q / g.OH;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int fy = 0; fy < g.kh; fy++) {
    int iy = oy * g.sh - g.pt + fy * g.dh;
    if (iy < 0 || iy >= g.H) continue;
    for (int fx = 0; fx < g.kw; fx++) {
      int ix = ox * g.sw - g.pl + fx * g.dw;
      if (ix < 0 || ix >= g.W) continue;
      float4 xv = *reinterpret_cast<const float4*>(x + ((n * g.H + iy) * g.W + ix) * (long)g.Cin + c);
      float4 wv = *reinterpret_cast<const float4*>(w + (long)(fy * g.kw + fx) * g.Cin + c);
      acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
      acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
    }
  }
  float4 b = *reinterpret_cast<const float4*>(bias + c);
  float4 v;
  v.x = act_fn(acc.x + b.x, act); v.y = act_fn(acc.y + b.y, act); v.z = act_fn(acc.z + b.z, act); v.w = act_fn(acc.w + b.w, act);
  long o = p * g.Cin + c;
  if (res) { float4 r = *reinterpret_cast<const float4*>(res + o); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  *reinterpret_cast<float4*>(y + o) = v;
}

// Stride-1 3x3 depthwise with dilation d (DeepLab's 33x33 layers, d = 1 / 2 / 4): lane = (channel quad, column x, row residue ry)
// and walks the rows ry, ry + d, ry + 2d, …  Consecutive outputs of that walk share two of their three tap rows, so the lane keeps
// a 3 x 3 window of float4 in registers and loads ONE new tap row (3 x 16 bytes) per output instead of nine taps: the lane-per-output
// form ran at the L2's 9x read amplification (2.2 TB/s algorithmic, 38 % of DeepLab's network time).  Lanes of a wave are
// consecutive channel quads of one pixel: every load is a contiguous kilobyte.  FMA order per output = (fy, fx) ascending, as dw_conv_k.
__global__ __launch_bounds__(kThreads) void dw_col_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                    const float* __restrict__ res, float* __restrict__ y, long total, int H, int W, int C, int d, int act) {
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2;
  const int c = (int)(idx % C4) * 4;
  long q = idx / C4;
  const int ox = (int)(q % W);
  q /= W;
  const int ry = (int)(q % d);
  const long n = q / d;
  float4 wv[9];
#pragma unroll
  for (int k = 0; k < 9; k++) wv[k] = *reinterpret_cast<const float4*>(w + (long)k * C + c);
  const float4 b = *reinterpret_cast<const float4*>(bias + c);
  const float* base = x + n * (long)H * W * C + c;
  const bool vl = ox - d >= 0, vr = ox + d < W;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_row = [&](int iy, float4& l, float4& m, float4& r) {
    l = m = r = z;
    if (iy >= 0 && iy < H) {
      const float* p = base + ((long)iy * W + ox) * C;
      m = *reinterpret_cast<const float4*>(p);
      if (vl) l = *reinterpret_cast<const float4*>(p - (long)d * C);
      if (vr) r = *reinterpret_cast<const float4*>(p + (long)d * C);
    }
  };
  float4 t0, t1, t2, m0, m1, m2, b0, b1, b2;
  load_row(ry - d, t0, t1, t2);
  load_row(ry, m0, m1, m2);
  for (int oy = ry; oy < H; oy += d) {
    load_row(oy + d, b0, b1, b2);
    float4 acc = z;
#define BSX_TAP(v, k) acc.x = fmaf(v.x, wv[k].x, acc.x); acc.y = fmaf(v.y, wv[k].y, acc.y); acc.z = fmaf(v.z, wv[k].z, acc.z); acc.w = fmaf(v.w, wv[k].w, acc.w);
    BSX_TAP(t0, 0) BSX_TAP(t1, 1) BSX_TAP(t2, 2) BSX_TAP(m0, 3) BSX_TAP(m1, 4) BSX_TAP(m2, 5) BSX_TAP(b0, 6) BSX_TAP(b1, 7) BSX_TAP(b2, 8)
#undef BSX_TAP
    float4 v;
    v.x = act_fn(acc.x + b.x, act); v.y = act_fn(acc.y + b.y, act); v.z = act_fn(acc.z + b.z, act); v.w = act_fn(acc.w + b.w, act);
    const long o = ((n * H + oy) * (long)W + ox) * C + c;
    if (res) { const float4 r = *reinterpret_cast<const float4*>(res + o); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    *reinterpret_cast<float4*>(y + o) = v;
    t0 = m0; t1 = m1; t2 = m2; m0 = b0; m1 = b1; m2 = b2;
  }
}

// -------------------------------------------------------------------------------------
// global average pool: block = (frame, channel-quad chunk); rows of lanes stride the pixels
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void gap_k(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int CG, int out_c4_stride,
                                                 int out_c4_off, int accumulate) {
  __shared__ float4 sm[kThreads];
  const int C4 = C >> 2;
  const int n = blockIdx.x;
  const int cg = threadIdx.x % CG, row = threadIdx.x / CG, rows = kThreads / CG;
  const int cgi = blockIdx.y * CG + cg;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cgi < C4 && row < rows) {
    const float4* xp = reinterpret_cast<const float4*>(x) + (long)n * HW * C4 + cgi;
    for (int p = row; p < HW; p += rows) { float4 v = xp[(long)p * C4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (row == 0 && cgi < C4) {
    float4 t = sm[cg];
    for (int r = 1; r < rows; r++) { float4 v = sm[r * CG + cg]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    float inv = (float)HW;
    t.x /= inv; t.y /= inv; t.z /= inv; t.w /= inv;
    float4* o = reinterpret_cast<float4*>(y) + (long)n * out_c4_stride + out_c4_off + cgi;
    if (accumulate) { const float4 p = *o; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }   // GAP(a + b) = GAP(a) + GAP(b)
    *o = t;
  }
}

// -------------------------------------------------------------------------------------
// elementwise add / mul (optionally channel-vector broadcast) / unary act / a*s+c
// -------------------------------------------------------------------------------------
__device__ __forceinline__ float elt_one(float a, float b, float c, int op) {
  switch (op) {
    case kEltAdd: return a + b;
    case kEltMul: return a * b;
    case kEltMulAdd: return __fadd_rn(__fmul_rn(a, b), c);  // MUL rounded, then ADD — as two graph ops
    default: return a;
  }
}

__global__ __launch_bounds__(kThreads) void elt4_k(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                  float4* __restrict__ y, long total4, long per_frame4, int C4, int op, int bcast, int act) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total4) return;
  float4 av = a[i];
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cv = bv;
  if (op != kEltUnary) bv = bcast ? b[(i / per_frame4) * C4 + (i % C4)] : b[i];
  if (op == kEltMulAdd) cv = c[i];
  float4 v;
  v.x = act_fn(elt_one(av.x, bv.x, cv.x, op), act); v.y = act_fn(elt_one(av.y, bv.y, cv.y, op), act);
  v.z = act_fn(elt_one(av.z, bv.z, cv.z, op), act); v.w = act_fn(elt_one(av.w, bv.w, cv.w, op), act);
  y[i] = v;
}

__global__ __launch_bounds__(kThreads) void elt1_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                  float* __restrict__ y, long total, long per_frame, int C, int op, int bcast, int act) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  float bv = 0.f, cv = 0.f;
  if (op != kEltUnary) bv = bcast ? b[(i / per_frame) * C + (i % C)] : b[i];
  if (op == kEltMulAdd) cv = c[i];
  y[i] = act_fn(elt_one(a[i], bv, cv, op), act);
}

// -------------------------------------------------------------------------------------
// RESIZE_BILINEAR (TFLite reference resize_bilinear.h semantics; same association as the oracle)
// -------------------------------------------------------------------------------------
__device__ __forceinline__ void interp(int o, float scale, bool half_pixel, int in_size, float* frac, int* lo, int* hi) {
  float v = half_pixel ? __fadd_rn(__fmul_rn((float)o + 0.5f, scale), -0.5f) : __fmul_rn((float)o, scale);
  float fl = floorf(v);
  *lo = max((int)fl, 0);
  *hi = min((int)ceilf(v), in_size - 1);
  *frac = v - (float)*lo;
}
__device__ __forceinline__ float bilerp(float x00, float x10, float x01, float x11, float dy, float dx) {
  float a = __fmul_rn(__fmul_rn(x00, 1.f - dy), 1.f - dx);
  float b = __fmul_rn(__fmul_rn(x10, dy), 1.f - dx);
  float c = __fmul_rn(__fmul_rn(x01, 1.f - dy), dx);
  float d = __fmul_rn(__fmul_rn(x11, dy), dx);
  return __fadd_rn(__fadd_rn(__fadd_rn(a, b), c), d);
}

template <int V>  // V = 4 (float4 per lane) or 1
__global__ __launch_bounds__(kThreads) void resize_k(const float* __restrict__ x, float* __restrict__ y, long total, int H, int W, int C,
                                                    int OH, int OW, float hs, float ws, int half_pixel) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int CV = C / V;
  int c = (int)(i % CV) * V;
  long p = i / CV;
  int ox = (int)(p % OW);
  long q = p / OW;
  int oy = (int)(q % OH);
  long n = q / OH;
  float dy, dx; int y0, y1, x0, x1;
  interp(oy, hs, half_pixel, H, &dy, &y0, &y1);
  interp(ox, ws, half_pixel, W, &dx, &x0, &x1);
  const float* b = x + n * (long)H * W * C + c;
  const float* p00 = b + ((long)y0 * W + x0) * C; const float* p10 = b + ((long)y1 * W + x0) * C;
  const float* p01 = b + ((long)y0 * W + x1) * C; const float* p11 = b + ((long)y1 * W + x1) * C;
  float* o = y + p * C + c;
  if (V == 4) {
    float4 a = *reinterpret_cast<const float4*>(p00), bb = *reinterpret_cast<const float4*>(p10);
    float4 cc = *reinterpret_cast<const float4*>(p01), d = *reinterpret_cast<const float4*>(p11);
    float4 v;
    v.x = bilerp(a.x, bb.x, cc.x, d.x, dy, dx); v.y = bilerp(a.y, bb.y, cc.y, d.y, dy, dx);
    v.z = bilerp(a.z, bb.z, cc.z, d.z, dy, dx); v.w = bilerp(a.w, bb.w, cc.w, d.w, dy, dx);
    *reinterpret_cast<float4*>(o) = v;
  } else {
    o[0] = bilerp(p00[0], p10[0], p01[0], p11[0], dy, dx);
  }
}

// Channel counts that are not a multiple of 4 (DeepLab's 21 classes): lane = output PIXEL.  The interpolation coefficients
// are computed once per pixel instead of once per element, the C results of 256 consecutive pixels are staged in LDS
// (stride C: odd or small → conflict-free) and leave as fully coalesced 4-byte stores of one contiguous block.
constexpr int kResizePxMaxC = 32;
__global__ __launch_bounds__(kThreads) void resize_px_k(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int OH, int OW, float hs,
                                                       float ws, int half_pixel) {
  __shared__ float tile[kThreads * kResizePxMaxC];
  const long n = blockIdx.y;
  const unsigned p0 = blockIdx.x * kThreads, P = (unsigned)(OH * OW);
  const unsigned p = p0 + threadIdx.x;
  if (p < P) {
    const int oy = (int)(p / (unsigned)OW), ox = (int)(p - (unsigned)oy * (unsigned)OW);
    float dy, dx; int y0, y1, x0, x1;
    interp(oy, hs, half_pixel, H, &dy, &y0, &y1);
    interp(ox, ws, half_pixel, W, &dx, &x0, &x1);
    const float* b = x + n * (long)H * W * C;
    const float* p00 = b + (y0 * W + x0) * C; const float* p10 = b + (y1 * W + x0) * C;
    const float* p01 = b + (y0 * W + x1) * C; const float* p11 = b + (y1 * W + x1) * C;
    for (int c = 0; c < C; c++) tile[threadIdx.x * C + c] = bilerp(p00[c], p10[c], p01[c], p11[c], dy, dx);
  }
  __syncthreads();
  const unsigned valid = min((unsigned)kThreads, P - p0) * (unsigned)C;
  float* o = y + (n * (long)P + p0) * C;
  for (unsigned i = threadIdx.x; i < valid; i += kThreads) o[i] = tile[i];
}

// -------------------------------------------------------------------------------------
// DeepLab tail: RESIZE_BILINEAR (33 → 257, align_corners) + 21-way argmax + "person" test + temporal IIR in one pass
// (lib/libbackscrub.cc:318-332 on the output of the graph's last op).  The 257x257x21 logits — 5.5 MB per frame, written and
// read back once each by the unfused pair resize_px_k + decode_argmax_k — never exist: a workgroup stages the few low-resolution
// pixels its 64 x 4 output pixels interpolate from in LDS and every lane scans the 21 classes of its pixel from there.
// Same interpolation arithmetic (interp / bilerp) and the same first-maximum-wins scan as the unfused kernels.
// -------------------------------------------------------------------------------------
constexpr int kFusedTW = 64, kFusedTH = 16, kFusedMaxSrc = 160;    // source pixels per tile (rows x cols), C <= kResizePxMaxC
template <bool PERSON_ONLY, int CC = 0, int PC = -1>      // PERSON_ONLY: only "is the first maximum the person class?" is formed (two running maxima), not the argmax itself;
                                                          // CC / PC: class count and person class as compile-time constants (the scan then resolves per class: one v_max each)
__global__ __launch_bounds__(kThreads) void resize_argmax_iir_k(const float* __restrict__ x, uint8_t* __restrict__ ofinal, int H, int W, int C_, int OH, int OW,
                                                               float hs, float ws, int half_pixel, int person_, int ntx, int nty, int n_frames) {
  const int C = CC > 0 ? CC : C_;
  int person = CC > 0 ? PC : person_;
  __shared__ float src[kFusedMaxSrc * kResizePxMaxC];
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(ntx * nty), (unsigned)n_frames, &f_, &t_);      // a frame's tiles on one XCD: the 91 KB logits tensor is fetched into ONE L2 (PMC: 2.7x over-fetch in the plain order)
  const long n = f_;
  const int tby = (int)t_ / ntx, tbx = (int)t_ - tby * ntx;
  const int ox0 = tbx * kFusedTW, oy0 = tby * kFusedTH;
  const int tx = threadIdx.x & (kFusedTW - 1), ty = threadIdx.x >> 6;
  // source window of the tile (monotone maps)
  float fr; int a0, a1, sy0, sy1, sx0, sx1;
  interp(oy0, hs, half_pixel, H, &fr, &sy0, &a1);
  interp(min(oy0 + kFusedTH - 1, OH - 1), hs, half_pixel, H, &fr, &a0, &sy1);
  interp(ox0, ws, half_pixel, W, &fr, &sx0, &a1);
  interp(min(ox0 + kFusedTW - 1, OW - 1), ws, half_pixel, W, &fr, &a0, &sx1);
  const int SR = sy1 - sy0 + 1, SC = sx1 - sx0 + 1;
  const float* b = x + n * (long)H * W * C;
  // Every lane owns kFusedTH / 4 pixels of one column (rows ty, ty + 4, ...).  Their previous mask bytes are requested FIRST: the kernel is a chain
  // of memory latencies (source window → barrier → arithmetic → old byte → store), not of arithmetic, and this takes one link out of it.
  constexpr int PPL = kFusedTH / 4;
  const int ox = ox0 + tx;
  uint8_t* obase = ofinal + n * (long)OH * OW + ox;
  uint8_t old[PPL];
#pragma unroll
  for (int k = 0; k < PPL; k++) { const int oy = oy0 + ty + 4 * k; old[k] = (ox < OW && oy < OH) ? obase[(long)oy * OW] : (uint8_t)0; }
  // LDS pixels are padded to CP floats — 24 for the 21 classes: 16-byte aligned, so that a lane reads a corner's classes as six ds_read_b128 instead of
  // 21 ds_read_b32
  const bool generic = person < 0;                                       // (BSX_TAIL_GENERIC: person arrives as -1 - person)
  if (generic) person = -1 - person;
  const int CP = (C <= 24 && !generic) ? 24 : kResizePxMaxC;
  const unsigned cmagic = 0xFFFFFFFFu / (unsigned)C + 1u;                // i / C for i < 2^16
  for (int r = 0; r < SR; r++) {
    const float* row = b + ((long)(sy0 + r) * W + sx0) * C;
    for (int i = threadIdx.x; i < SC * C; i += kThreads) {
      const int px = (int)__umulhi((unsigned)i, cmagic);
      src[(r * SC + px) * CP + (i - px * C)] = row[i];
    }
  }
  __syncthreads();
  if (ox >= OW) return;
  float dx; int x0, x1;
  interp(ox, ws, half_pixel, W, &dx, &x0, &x1);
#pragma unroll
  for (int k = 0; k < PPL; k++) {
    const int oy = oy0 + ty + 4 * k;
    if (oy >= OH) break;
    float dy; int y0, y1;
    interp(oy, hs, half_pixel, H, &dy, &y0, &y1);
    const float* p00 = src + ((y0 - sy0) * SC + (x0 - sx0)) * CP; const float* p10 = src + ((y1 - sy0) * SC + (x0 - sx0)) * CP;
    const float* p01 = src + ((y0 - sy0) * SC + (x1 - sx0)) * CP; const float* p11 = src + ((y1 - sy0) * SC + (x1 - sx0)) * CP;
    float maxval = -10000.f; int maxpos = 0;                         // first maximum wins, as the reference loop
    // PERSON_ONLY: the scan "v > maxval → take it" ends on the person class iff v[person] > max(-10000, v[c < person]) and v[person] >= max(v[c > person])
    float mbefore = -10000.f, mafter = -__builtin_inff(), vperson = 0.f;
    if (CP == 24) {
      // four classes at a time; the arithmetic of bilerp() exactly — ((x * wy) * wx) per corner, summed left to right, nothing contracted
      // (-ffp-contract=off) — on two-wide vectors (v_pk_mul_f32 / v_pk_add_f32)
      const float omdy = 1.f - dy, omdx = 1.f - dx;
      const f2v wy0 = {omdy, omdy}, wy1 = {dy, dy}, wx0 = {omdx, omdx}, wx1 = {dx, dx};
#pragma unroll
      for (int q = 0; q < 6; q++) {
        if (CC > 0 && 4 * q >= CC) break;
        const f4v v00 = *reinterpret_cast<const f4v*>(p00 + 4 * q), v10 = *reinterpret_cast<const f4v*>(p10 + 4 * q);
        const f4v v01 = *reinterpret_cast<const f4v*>(p01 + 4 * q), v11 = *reinterpret_cast<const f4v*>(p11 + 4 * q);
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          if (CC > 0 && 4 * q + 2 * h >= CC) break;                      // padding classes of the last quad: never looked at
          const f2v a = (f2v{v00[2 * h], v00[2 * h + 1]} * wy0) * wx0, bb = (f2v{v10[2 * h], v10[2 * h + 1]} * wy1) * wx0;
          const f2v c2 = (f2v{v01[2 * h], v01[2 * h + 1]} * wy0) * wx1, d = (f2v{v11[2 * h], v11[2 * h + 1]} * wy1) * wx1;
          const f2v sum = ((a + bb) + c2) + d;
          r4[2 * h] = sum.x; r4[2 * h + 1] = sum.y;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = 4 * q + e;
          if (PERSON_ONLY) {
            if (c < C) {                                                 // (uniform conditions)
              if (c < person) mbefore = fmaxf(mbefore, r4[e]);
              else if (c > person) mafter = fmaxf(mafter, r4[e]);
              else vperson = r4[e];
            }
          } else if (c < C && r4[e] > maxval) { maxval = r4[e]; maxpos = c; }
        }
      }
      if (PERSON_ONLY) maxpos = (vperson > mbefore && vperson >= mafter) ? person : person + 1;
    } else {
      for (int c = 0; c < C; c++) { const float v = bilerp(p00[c], p10[c], p01[c], p11[c], dy, dx); if (v > maxval) { maxval = v; maxpos = c; } }
    }
    const uint8_t val = maxpos == person ? 0 : 255;
    obase[(long)oy * OW] = (uint8_t)((val & 0xE0) | (old[k] >> 3));
  }
}

// -------------------------------------------------------------------------------------
// channel concat (up to 4 inputs, channel counts multiples of 4)
// -------------------------------------------------------------------------------------
struct ConcatArgs { const float4* in[4]; int c4[4]; int n_in; };
__global__ __launch_bounds__(kThreads) void concat_k(ConcatArgs a, float4* __restrict__ y, long total4, int C4out) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total4) return;
  int c = (int)(i % C4out);
  long p = i / C4out;
  for (int k = 0; k < a.n_in; k++) {
    if (c < a.c4[k]) { y[i] = a.in[k][p * a.c4[k] + c]; return; }
    c -= a.c4[k];
  }
}

// -------------------------------------------------------------------------------------
// Convolution2DTransposeBias with kernel == stride (no overlap): every output pixel is one
// Cin-long dot product per output channel, accumulated on top of the bias
// (lib/transpose_conv_bias.cc:70-110 initialises the output with the bias, then adds).
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tconv_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, long M, int H, int W, int Cin, int OH, int OW, int Cout, int kh,
                                                   int kw, int act) {
  long p = (long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= M) return;
  int ox = (int)(p % OW);
  long q = p / OW;
  int oy = (int)(q % OH);
  long n = q / OH;
  int iy = oy / kh, fy = oy % kh, ix = ox / kw, fx = ox % kw;
  const float4* xp = reinterpret_cast<const float4*>(x + ((n * H + iy) * W + ix) * (long)Cin);
  const int C4 = Cin >> 2;
  for (int oc = 0; oc < Cout; oc++) {
    const float4* wp = reinterpret_cast<const float4*>(w + ((long)(fy * kw + fx) * Cout + oc) * Cin);
    float acc = bias[oc];
    for (int c = 0; c < C4; c++) {
      float4 xv = xp[c], wv = wp[c];
      acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
    }
    y[p * Cout + oc] = act_fn(acc, act);
  }
}

inline unsigned blocks_for(long total) { return (unsigned)((total + kThreads - 1) / kThreads); }

}  // namespace

// → false when the source window of some tile would not fit the staging area (the caller then keeps the unfused pair)
bool resize_argmax_fusable(const Step& st) {
  if (st.kind != StepKind::Resize || st.Cin > kResizePxMaxC || st.OH < st.H || st.OW < st.W) return false;
  // up-sampling: a 64 x 4 tile touches at most ceil(4 * H / OH) + 2 rows and ceil(64 * W / OW) + 2 columns
  const int rows = (kFusedTH * st.H + st.OH - 1) / st.OH + 2, cols = (kFusedTW * st.W + st.OW - 1) / st.OW + 2;
  return rows * cols <= kFusedMaxSrc;
}
hipError_t launch_resize_argmax_iir(const Step& st, const float* x, uint8_t* ofinal, int n, hipStream_t s, bool generic) {
  float hs = (float)st.H / (float)st.OH, ws = (float)st.W / (float)st.OW;
  if (st.align_corners && st.OH > 1) hs = (float)(st.H - 1) / (float)(st.OH - 1);
  if (st.align_corners && st.OW > 1) ws = (float)(st.W - 1) / (float)(st.OW - 1);
  const int ntx = (st.OW + kFusedTW - 1) / kFusedTW, nty = (st.OH + kFusedTH - 1) / kFusedTH;
  if ((unsigned long long)ntx * nty * (unsigned long long)n >= (1ull << 31)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)(ntx * nty) * (unsigned)n);
  static const bool xcd_on = !(BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 0);      // A/B timing: 0 = plain frame-major workgroup order
  const int nf = xcd_on ? n : 0;
  const int person = 15;                                           // lib/libbackscrub.cc:330 (pascal VOC class 15)
  if (!generic && st.Cin == 21 && person == 15) resize_argmax_iir_k<true, 21, 15><<<grid, kThreads, 0, s>>>(x, ofinal, st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel, person, ntx, nty, nf);   // DeepLab / PASCAL VOC
  else if (!generic && st.Cin <= 24 && person < st.Cin) resize_argmax_iir_k<true><<<grid, kThreads, 0, s>>>(x, ofinal, st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel, person, ntx, nty, nf);
  else resize_argmax_iir_k<false><<<grid, kThreads, 0, s>>>(x, ofinal, st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel, generic ? -1 - person : person, ntx, nty, nf);
  return hipGetLastError();
}

// Dynamic-LDS limits of the fused kernels.  Function attributes belong to the (device, kernel) pair: set for the CURRENT device, once per context
// (bsx_new, under its device guard) — a process-wide "done" flag would leave a second GPU's copies at the 64 KB default.
hipError_t nn_prepare() {
  const int full = 160 * 1024;
  hipError_t e;
#define BSX_ATTR(K) if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, full)) != hipSuccess) return e
#define BSX_ATTR_IR(T, SL) BSX_ATTR((ir_expand_dw_k<T, SL, 32>)); BSX_ATTR((ir_expand_dw_k<T, SL, 24>)); BSX_ATTR((ir_expand_dw_k<T, SL, 16>))
  BSX_ATTR_IR(3, 1); BSX_ATTR_IR(3, 2); BSX_ATTR_IR(3, 3); BSX_ATTR_IR(1, 1); BSX_ATTR_IR(1, 2); BSX_ATTR_IR(1, 3);
#define BSX_ATTR_IR16(T, SL) BSX_ATTR((ir_expand_dw_k<T, SL, 32, true>)); BSX_ATTR((ir_expand_dw_k<T, SL, 24, true>)); BSX_ATTR((ir_expand_dw_k<T, SL, 16, true>))
  BSX_ATTR_IR16(1, 1); BSX_ATTR_IR16(1, 2); BSX_ATTR_IR16(1, 3); BSX_ATTR_IR16(3, 1); BSX_ATTR_IR16(3, 2); BSX_ATTR_IR16(3, 3);
#undef BSX_ATTR_IR16
  BSX_ATTR((pw_chain3_k<kChainWaves, kChainNP, kChainS0, kChainP1, kChainP2>));
  BSX_ATTR(dl_head0_k<false>);
  BSX_ATTR(dl_head0_k<true>);
  BSX_ATTR((ir_expand_dw_k<3, 1, 32, false, 1024>)); BSX_ATTR((ir_expand_dw_k<3, 2, 32, false, 1024>)); BSX_ATTR((ir_expand_dw_k<3, 3, 32, false, 1024>));
#undef BSX_ATTR_IR
#undef BSX_ATTR
  return hipSuccess;
}

hipError_t launch_step(const Step& st, const Plan& plan, float* arena, float* net_in, float* net_out, const float* weights, int n, int n_cap,
                       hipStream_t s, const uint16_t* weights16, int f16_terms, const uint32_t* net_in_u8, float in_scale, float in_offset) {
  auto P = [&](int t) -> float* {
    if (t < 0) return nullptr;
    if (t == plan.input) return net_in;
    if (t == plan.output) return net_out;
    return arena + (size_t)plan.tensor_off[t] * (size_t)n_cap;
  };
  const float* w = weights + st.w_off;
  const float* b = weights + st.b_off;
  switch (st.kind) {
    case StepKind::PwConv: {
      long M = (long)n * st.OH * st.OW;
      dim3 grid(blocks_for(M), st.cout_pad / st.cout_tile);
      int HW = st.OH * st.OW;
      if (st.fused_away && plan.steps[0].fuse_head0) break;          // ran inside dl_head0_k (the planner decides: BSX_NO_HEAD0 is read there)
      if (st.chain_mid >= 0 && chain3_on(plan, st.chain_mid, n, weights16, f16_terms)) break;      // runs inside the chain's launch (at its middle step)
      if (st.chain_first >= 0 && chain3_on(plan, (int)(&st - plan.steps.data()), n, weights16, f16_terms)) {
        const Step& ca = plan.steps[st.chain_first];
        const Step& cc = plan.steps[st.chain_last];
        const _Float16* wsx = reinterpret_cast<const _Float16*>(weights16) + st.chain_w16_off;
        constexpr int kW = kChainWaves, kNP = kChainNP, kPx = kW * 16 * kNP;
        const size_t lds = 2 * (size_t)(4 * kChainP1 + 4) * 1024;
#ifdef BSX_DEBUG_SWITCHES
        static const int form = BSX_DBG_ENV("BSX_CHAIN_FORM") ? atoi(BSX_DBG_ENV("BSX_CHAIN_FORM")) : 0;      // geometry A/B (profiles/r06aa, r06ab): 82 = 8 waves x 32 pixels, 161 / 1610 = 16 waves x 16 pixels, 421 = the default geometry with the DMA at the round top
#define BSX_CHAIN_ALT(WV, NPX, DTX) { constexpr int px = WV * 16 * NPX; static bool once = false; \
          if (!once) { once = true; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_chain3_k<WV, NPX, kChainS0, kChainP1, kChainP2, DTX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
          pw_chain3_k<WV, NPX, kChainS0, kChainP1, kChainP2, DTX><<<(unsigned)((M + px - 1) / px), WV * 64, lds, s>>>(P(ca.in0), wsx, weights + ca.b_off, b, P(st.out_bias), weights + cc.b_off, P(cc.out), M, HW, cc.Cout, ca.act, st.act, cc.act); break; }
        if (form == 82) BSX_CHAIN_ALT(8, 2, true)
        if (form == 161) BSX_CHAIN_ALT(16, 1, true)
        if (form == 421) BSX_CHAIN_ALT(4, 2, true)
        if (form == 1610) BSX_CHAIN_ALT(16, 1, false)
#undef BSX_CHAIN_ALT
#endif
        pw_chain3_k<kW, kNP, kChainS0, kChainP1, kChainP2><<<(unsigned)((M + kPx - 1) / kPx), kW * 64, lds, s>>>(
            P(ca.in0), wsx, weights + ca.b_off, b, P(st.out_bias), weights + cc.b_off, P(cc.out), M, HW, cc.Cout, ca.act, st.act, cc.act);
        break;
      }
      static const bool no_gemm = BSX_DBG_ENV("BSX_NO_PW_GEMM") != nullptr;
      if (st.fuse_dw >= 0 && weights16 && f16_terms > 0) {          // expand 1x1 + depthwise 3x3 of an inverted-residual block in one kernel
        const Step& dws = plan.steps[st.fuse_dw];
        const IrGeom ig = ir_geometry(st.OH, st.OW, st.Cout, dws.OH, dws.sh, dws.dh);
        if (ig.CH == 0) return hipErrorInvalidValue;                 // the planner checked the same function
        const _Float16* w16 = reinterpret_cast<const _Float16*>(weights16) + st.w16_off;
        const size_t lds = (size_t)ir_lds_bytes(ig.rows, st.OW, ig.CH);
        if ((long)ig.rows * st.OW * dws.dh >= 65536) return hipErrorInvalidValue;   // the kernel divides item indices by multiplication
        const dim3 gi((unsigned)(st.Cout / ig.CH) * (unsigned)ig.nbands * (unsigned)n);
        const int slabs = st.k16_pad / 32;
        static const int ir_phases = BSX_DBG_ENV("BSX_IR_PHASES") ? atoi(BSX_DBG_ENV("BSX_IR_PHASES")) : 3;      // timing experiments: 1 = expand only, 2 = depthwise only
        // reduced-precision storage (f16_terms bit 4): only when the depthwise output's single reader is a GEMM that will take the f16 form (same M rule)
        const bool out16 = (f16_terms & 16) && (size_t)st.fuse_dw + 1 < plan.steps.size() && plan.steps[st.fuse_dw + 1].in_from_fused_dw &&
                           plan.steps[st.fuse_dw + 1].in0 == dws.out && (long)n * dws.OH * dws.OW >= 8192 && !no_gemm;
        // 1024-lane workgroups (four waves per SIMD, <= 128 registers, input rows straight into the MFMA layout: no LDS left for sixteen re-order
        // buffers) for the whole-frame 32-channel layers.  Measured per K: 16 input channels (64-byte rows: nothing to coalesce) 149 -> 134 us;
        // 32 channels equal; 48 (two slabs) +5 %; 80 (three slabs) spills at 128 registers, 1.06 -> 1.72 ms.  Default: the 16-channel layers only
        // (BSX_IR_WAVES16 = bit mask over the slab count, for experiments; 0 = never).
        static const int w16_mask = BSX_DBG_ENV("BSX_IR_WAVES16") ? atoi(BSX_DBG_ENV("BSX_IR_WAVES16")) : -1;
        const bool w16_on = w16_mask < 0 ? (slabs == 1 && st.Cin <= 16) : ((w16_mask >> (slabs - 1)) & 1) != 0;
        if ((f16_terms & 15) == 3 && !out16 && ig.CH == 32 && ig.nbands == 1 && slabs >= 1 && slabs <= 3 && w16_on) {
#define BSX_IR16(SL) ir_expand_dw_k<3, SL, 32, false, 1024><<<gi, 1024, lds, s>>>(P(st.in0), w16, b, weights + dws.w_off, weights + dws.b_off, P(dws.out), st.OH, st.OW, st.Cin, st.k16_pad, st.Cout, st.cout_pad, st.act, dws.act, dws.dh, dws.sh, dws.pad_t, dws.pad_l, dws.OH, dws.OW, ig.BH, ig.nbands, ir_phases)
          if (slabs == 1) BSX_IR16(1); else if (slabs == 2) BSX_IR16(2); else BSX_IR16(3);
#undef BSX_IR16
          break;
        }
#define BSX_IR(T, SL, C) { if (out16) ir_expand_dw_k<T, SL, C, true><<<gi, kIrThreads, lds, s>>>(P(st.in0), w16, b, weights + dws.w_off, weights + dws.b_off, P(dws.out), st.OH, st.OW, st.Cin, st.k16_pad, st.Cout, st.cout_pad, st.act, dws.act, dws.dh, dws.sh, dws.pad_t, dws.pad_l, dws.OH, dws.OW, ig.BH, ig.nbands, ir_phases); \
          else ir_expand_dw_k<T, SL, C, false><<<gi, kIrThreads, lds, s>>>(P(st.in0), w16, b, weights + dws.w_off, weights + dws.b_off, P(dws.out), st.OH, st.OW, st.Cin, st.k16_pad, st.Cout, st.cout_pad, st.act, dws.act, dws.dh, dws.sh, dws.pad_t, dws.pad_l, dws.OH, dws.OW, ig.BH, ig.nbands, ir_phases); }
#define BSX_IR_C(T, SL) { if (ig.CH == 32) BSX_IR(T, SL, 32) else if (ig.CH == 24) BSX_IR(T, SL, 24) else BSX_IR(T, SL, 16) }
        if ((f16_terms & 15) == 3) { if (slabs == 1) BSX_IR_C(3, 1) else if (slabs == 2) BSX_IR_C(3, 2) else BSX_IR_C(3, 3) }
        else { if (slabs == 1) BSX_IR_C(1, 1) else if (slabs == 2) BSX_IR_C(1, 2) else BSX_IR_C(1, 3) }
#undef BSX_IR_C
#undef BSX_IR
        break;
      }
      const bool gemm_ok = !no_gemm && M >= 8192 && (st.Cin & 3) == 0 && st.Cin >= 8 && st.cout_pad % 16 == 0 && st.cout_pad >= 16;
      // a step that carries a per-frame out_bias (the folded ASPP pool branch) has two forms only — this one and the GEMMs: between their M ranges
      // (DeepLab at 4-7 streams) and with BSX_NO_PW_GEMM it stays on the lane-per-output form
      if (M <= 4096 || (st.out_bias >= 0 && !gemm_ok)) {
        long total = M * st.Cout;
        pw_small_k<<<blocks_for(total), kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.in_scale), P(st.in2), P(st.out), total, HW, st.Cin, st.Cout, st.cout_pad, st.act, P(st.out_bias));
        break;
      }
      // enough rows and channels to fill 128 x 64 MFMA tiles → the GEMM form (BSX_NO_PW_GEMM=1 keeps the lane-per-pixel form)
      // (even K = 8 / N = 16 layers: the tiles are mostly padding, but A is read once and coalesced — measured faster than the lane-per-pixel form)
      if (gemm_ok) {
        dim3 gg((unsigned)((M + kGemmBM - 1) / kGemmBM), (st.cout_pad + kGemmBN - 1) / kGemmBN);
        if (weights16 && st.k16_pad > 0 && (f16_terms & 15) > 0) {        // split-f16 (3 terms, f32-grade) or plain f16-input (1 term) MFMA
          if (st.out_bias >= 0 && st.residual >= 0) return hipErrorInvalidValue;      // the epilogue carries ONE extra operand per tile (see gemm_store_tile)
          const _Float16* w16 = reinterpret_cast<const _Float16*>(weights16) + st.w16_off;
          // column tiles: 64 channels (NTW = 4), or 80 / 48 in ONE tile where that covers the whole layer (NTW = 5 / 3: the A block is
          // staged and split once instead of once per column tile — the 480 -> 80 and 288 -> 48 project layers)
          static const bool wide_ok = BSX_DBG_ENV("BSX_NO_GEMM_NTW") == nullptr;
          static const int gemm_dbg = BSX_DBG_ENV("BSX_GEMM_DBG") ? atoi(BSX_DBG_ENV("BSX_GEMM_DBG")) : 0;      // timing experiments: 1 = A from one L2-resident block, 2 = no stores
          // (128-column tiles for the 256-channel ASPP layers — A staged twice instead of four times — measured 35-43 % SLOWER: 168 registers, 3 workgroups per CU)
          const int ntw = (wide_ok && st.Cout % 80 == 0) ? 5 : ((wide_ok && st.Cout == 48) ? 3 : 4);
          const unsigned ncol = (unsigned)((st.Cout + ntw * 16 - 1) / (ntw * 16));
          if ((unsigned long long)gg.x * ncol >= (1ull << 31)) return hipErrorInvalidValue;
          const dim3 gw(gg.x * ncol);                              // 1-D: the kernel derives (column tile, row block) XCD-aware
#define BSX_F16S(T, N) pw_gemm_f16s_k<T, N><<<gw, kThreads, 0, s>>>(P(st.in0), w16, b, P(st.residual), P(st.in_scale), P(st.in2), P(st.out), M, HW, st.Cin, st.k16_pad, st.Cout, st.cout_pad, st.act, P(st.out_bias), gemm_dbg)
          if ((f16_terms & 15) == 3) { if (ntw == 5) BSX_F16S(3, 5); else if (ntw == 3) BSX_F16S(3, 3); else BSX_F16S(3, 4); }
          else if ((f16_terms & 16) && st.in_from_fused_dw) {        // its input was stored as f16 by the fused kernel before it (same M rule on both sides)
#define BSX_F16S16(N) pw_gemm_f16s_k<1, N, true><<<gw, kThreads, 0, s>>>(P(st.in0), w16, b, P(st.residual), P(st.in_scale), P(st.in2), P(st.out), M, HW, st.Cin, st.k16_pad, st.Cout, st.cout_pad, st.act, P(st.out_bias))
            if (ntw == 5) BSX_F16S16(5); else if (ntw == 3) BSX_F16S16(3); else BSX_F16S16(4);
#undef BSX_F16S16
          }
          else { if (ntw == 5) BSX_F16S(1, 5); else if (ntw == 3) BSX_F16S(1, 3); else BSX_F16S(1, 4); }
#undef BSX_F16S
          break;
        }
        pw_gemm_mfma_k<<<gg, kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.in_scale), P(st.in2), P(st.out), M, HW, st.Cin, st.Cout, st.cout_pad, st.act, P(st.out_bias));
        break;
      }
      if (st.out_bias >= 0) return hipErrorInvalidValue;          // the planner only folds a concat branch into convs that take one of the forms above
#define BSX_PW(CT) pw_conv_k<CT><<<grid, kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.in_scale), P(st.in2), P(st.out), M, HW, st.Cin, st.Cout, st.cout_pad, st.act)
      if (st.cout_tile == 16) BSX_PW(16); else BSX_PW(32);
#undef BSX_PW
      break;
    }
    case StepKind::Conv: {
      if (st.fuse_head0) {             // stem + depthwise + 1x1 (plan.steps[1], [2]) in one tiled kernel
        const Step& d1 = plan.steps[1];
        const Step& p2 = plan.steps[2];
        const int BH = head0_band_rows(st.W, st.OW), nb = (st.OH + BH - 1) / BH;
        const size_t fl = (size_t)head0_lds_floats(st.W, st.OW, BH);
        static const int h0_phases = BSX_DBG_ENV("BSX_H0_PHASES") ? atoi(BSX_DBG_ENV("BSX_H0_PHASES")) : 15;   // timing experiments
        static const bool h0_xcd = BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 2;      // one-XCD-per-frame band order: measured SLOWER here (0.927 vs 0.908 ms, profiles/r03o) — only with BSX_XCD_TILES=2
        const bool u8_fits = (long)(2 * (BH + 2) + 1) * st.W <= 4 * 3 * kH0Threads;                // three 4-pixel loads per lane cover the band's rows
        if (net_in_u8 && st.in0 == plan.input && !u8_fits) return hipErrorInvalidValue;            // (bsx_api decides with the same rule: head0_u8_ok)
        if (net_in_u8 && st.in0 == plan.input)
          dl_head0_k<true><<<(unsigned)nb * (unsigned)n, kH0Threads, fl * sizeof(float), s>>>(reinterpret_cast<const float*>(net_in_u8), w, b, weights + d1.w_off, weights + d1.b_off,
                                                                                            weights + p2.w_off, weights + p2.b_off, P(p2.out), st.H, st.W, st.OH, st.OW, st.pad_t,
                                                                                            st.pad_l, p2.Cout, p2.cout_pad, st.act, d1.act, p2.act, BH, nb, h0_phases, in_scale, in_offset, h0_xcd ? n : 0);
        else
          dl_head0_k<false><<<(unsigned)nb * (unsigned)n, kH0Threads, fl * sizeof(float), s>>>(P(st.in0), w, b, weights + d1.w_off, weights + d1.b_off, weights + p2.w_off,
                                                                                             weights + p2.b_off, P(p2.out), st.H, st.W, st.OH, st.OW, st.pad_t, st.pad_l, p2.Cout,
                                                                                             p2.cout_pad, st.act, d1.act, p2.act, BH, nb, h0_phases, 0.f, 0.f, h0_xcd ? n : 0);
        break;
      }
      long M = (long)n * st.OH * st.OW;
      ConvGeom g{st.H, st.W, st.Cin, st.OH, st.OW, st.Cout, st.cout_pad, st.kh, st.kw, st.sh, st.sw, st.dh, st.dw, st.pad_t, st.pad_l};
      dim3 grid(blocks_for(M), st.cout_pad / 16);
      conv_k<16><<<grid, kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.out), M, g, st.act);
      break;
    }
    case StepKind::DwConv: {
      if (st.fused_away && &st == &plan.steps[1] && plan.steps[0].fuse_head0) break;      // ran inside dl_head0_k
      if (st.fused_away && weights16 && f16_terms > 0) break;       // ran inside the expand convolution before it (ir_expand_dw_k)
      long total = (long)n * st.OH * st.OW * (st.Cin / 4);
      ConvGeom g{st.H, st.W, st.Cin, st.OH, st.OW, st.Cout, st.cout_pad, st.kh, st.kw, st.sh, st.sw, st.dh, st.dw, st.pad_t, st.pad_l};
      static const bool no_col = BSX_DBG_ENV("BSX_NO_DW_COL") != nullptr;
      if (!no_col && st.kh == 3 && st.kw == 3 && st.sh == 1 && st.sw == 1 && st.dh == st.dw && st.pad_t == st.dh && st.pad_l == st.dw && st.OH == st.H && st.OW == st.W &&
          st.H >= 4 * st.dh) {                                   // SAME 3x3, stride 1: the sliding-window column walk
        const long lanes = (long)n * st.dh * st.W * (st.Cin / 4);
        dw_col_k<<<blocks_for(lanes), kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.out), lanes, st.H, st.W, st.Cin, st.dh, st.act);
        break;
      }
      dw_conv_k<<<blocks_for(total), kThreads, 0, s>>>(P(st.in0), w, b, P(st.residual), P(st.out), total, g, st.act);
      break;
    }
    case StepKind::Gap: {
      if (st.Cin % 4) return hipErrorInvalidValue;
      auto one = [&](int tensor, int C, int c_off, int accumulate) {
        int C4 = C / 4;
        int CG = 1;
        while (CG * 2 <= C4 && CG * 2 <= 64) CG *= 2;  // power of two ≤ 64 so rows = 256/CG is exact
        dim3 grid(n, (C4 + CG - 1) / CG);
        gap_k<<<grid, kThreads, 0, s>>>(P(tensor), P(st.out), st.H * st.W, C, CG, st.Cout / 4, c_off / 4, accumulate);
      };
      if (st.concat_in.empty()) one(st.in0, st.Cin, 0, 0);
      else {
        int off = 0;
        for (size_t k = 0; k < st.concat_in.size(); k++) {
          one(st.concat_in[k], st.concat_c[k], off, st.gap_sum && k > 0);
          if (!st.gap_sum) off += st.concat_c[k];
        }
      }
      break;
    }
    case StepKind::Eltwise: {
      long per_frame = (long)st.H * st.W * st.Cin;
      long total = per_frame * n;
      if (st.Cin % 4 == 0) {
        elt4_k<<<blocks_for(total / 4), kThreads, 0, s>>>((const float4*)P(st.in0), (const float4*)P(st.in1), (const float4*)P(st.in2),
                                                         (float4*)P(st.out), total / 4, per_frame / 4, st.Cin / 4, st.elt, st.bcast1, st.act);
      } else {
        elt1_k<<<blocks_for(total), kThreads, 0, s>>>(P(st.in0), P(st.in1), P(st.in2), P(st.out), total, per_frame, st.Cin, st.elt, st.bcast1, st.act);
      }
      break;
    }
    case StepKind::Resize: {
      float hs = (float)st.H / (float)st.OH, ws = (float)st.W / (float)st.OW;
      if (st.align_corners && st.OH > 1) hs = (float)(st.H - 1) / (float)(st.OH - 1);
      if (st.align_corners && st.OW > 1) ws = (float)(st.W - 1) / (float)(st.OW - 1);
      if (st.Cin % 4 == 0) {
        long total = (long)n * st.OH * st.OW * (st.Cin / 4);
        resize_k<4><<<blocks_for(total), kThreads, 0, s>>>(P(st.in0), P(st.out), total, st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel);
      } else if (st.Cin <= kResizePxMaxC && n <= 65535) {
        resize_px_k<<<dim3(blocks_for((long)st.OH * st.OW), n), kThreads, 0, s>>>(P(st.in0), P(st.out), st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel);
      } else {
        long total = (long)n * st.OH * st.OW * st.Cin;
        resize_k<1><<<blocks_for(total), kThreads, 0, s>>>(P(st.in0), P(st.out), total, st.H, st.W, st.Cin, st.OH, st.OW, hs, ws, st.half_pixel);
      }
      break;
    }
    case StepKind::Concat: {
      if (st.concat_in.size() > 4) return hipErrorInvalidValue;
      ConcatArgs a{};
      a.n_in = (int)st.concat_in.size();
      for (int k = 0; k < a.n_in; k++) { a.in[k] = (const float4*)P(st.concat_in[k]); a.c4[k] = st.concat_c[k] / 4; }
      long total4 = (long)n * st.OH * st.OW * (st.Cout / 4);
      concat_k<<<blocks_for(total4), kThreads, 0, s>>>(a, (float4*)P(st.out), total4, st.Cout / 4);
      break;
    }
    case StepKind::TConv: {
      long M = (long)n * st.OH * st.OW;
      tconv_k<<<blocks_for(M), kThreads, 0, s>>>(P(st.in0), w, b, P(st.out), M, st.H, st.W, st.Cin, st.OH, st.OW, st.Cout, st.kh, st.kw, st.act);
      break;
    }
  }
  return hipGetLastError();
}

}  // namespace bsx
