// mid_prelude.hip — device templates of the SPECIALISED per-frame network program (gfx950).
//
// The interpreter of kernels_frame.hip walks a micro-op table: every op pays a descriptor fetch, a dispatch chain, run-time
// index arithmetic (integer division by tensor widths, address-space tests) and loops whose trip counts the compiler cannot
// see — measured 2-3k cycles per op before the first useful instruction, and bodies 3-6x longer than their arithmetic.
// Here every op of ONE planned graph becomes a call of one of these templates with ALL geometry, LDS offsets, strides,
// activations and operand address spaces as compile-time constants of a traits struct the generator (gen_mid.cpp) emits;
// the kernel is compiled with hipRTC when the context is created (rtc.cpp; cached on disk) and is straight-line code:
// no descriptors, no dispatch, only the bodies this graph uses.  (tools/microbench_icache.hip: cold straight-line code
// costs < 10 % over warm code on gfx950 — code size is not the constraint, dependent latency is.)
//
// Reference: what this executes is the middle of Interpreter::Invoke() (/root/reference/lib/libbackscrub.cc:307) for the
// Meet / MLKit graphs — CONV_2D 1x1, DEPTHWISE_CONV_2D, the squeeze-excite / gate chains (AVERAGE_POOL_2D, FULLY_CONNECTED /
// 1x1 CONV_2D, RELU, LOGISTIC, MUL), RESIZE_BILINEAR — with TFLite's f32 semantics (bias added last, fused activations).
//
// This file is embedded into libbsx.so as a string (build.py) and is ALSO a valid stand-alone HIP translation unit, so that
// `hipcc -fsyntax-only` and the tests can check it without a GPU.
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

namespace bsxm {

// operand address spaces (frame_program.hpp: LocSpace).  SP_GLB16: an ACTIVATION tensor in the arena stored as packed halves — the 16-bit activation
// storage mode (BSX_ACT16: every activation tensor that leaves LDS, i.e. the tensors exchanged with the segment kernels and the ones that do not
// fit; arithmetic stays f32, stores round to nearest even).  Weights, pooled partial sums and gate vectors in the arena are always f32 (SP_GLB).
enum { SP_NONE = 0, SP_LDS = 1, SP_GLB = 2, SP_GLB16 = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 3, ACT_HSWISH = 100, ACT_SIGMOID = 101 };   // tflite_model.hpp: Activation
// geometry of the workgroup (gen_mid.cpp emits both macros in front of this text: Plan::mid_lanes, Plan::lds_zero_off(); the defaults are the stand-alone syntax check's)
#ifndef BSXM_LANES
#define BSXM_LANES 1024
#endif
#ifndef BSXM_ZERO_OFF
#define BSXM_ZERO_OFF (160 * 256 - 4)
#endif
constexpr int kThreads = BSXM_LANES, kWaves = kThreads / 64;
// The last 16 bytes of the LDS block hold 0.0f for the whole kernel (frame_program.hpp: kLdsZeroOff; zeroed in the kernel's prologue, outside every planned block):
// a depthwise tap that falls outside the image is READ from there through an address select — one v_cndmask on the address instead of one per data register
// plus the zeroing of the weights of out-of-image rows (round 5: 130 of the 452 vector instructions of a 5x5 item were those selects).
constexpr int kZeroOff = BSXM_ZERO_OFF;

typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(1))) float glb_f;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f4v lds_v4;
typedef __attribute__((address_space(1))) f4v glb_v4;
typedef __attribute__((address_space(3))) f2v lds_v2;
typedef __attribute__((address_space(1))) f2v glb_v2;
typedef __attribute__((address_space(1))) char glb_c;
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) h4v glb_h4;
typedef __attribute__((address_space(1))) h2v glb_h2;
typedef __attribute__((address_space(1))) _Float16 glb_h;
// uniform base + zero-extended 32-bit BYTE offset (computed in 32 bits: every arena slice is far below 4 GB)
#define BSXM_G(T, g, off) ((T*)((glb_c*)(g) + (unsigned)((off) * 4)))
#define BSXM_H(T, g, off) ((T*)((glb_c*)(g) + (unsigned)((off) * 2)))          // the same tensor (same arena offset) as halves: element `off` at 2 * off bytes

// The lane index as every op reads it.  BSXM_OPAQUE_TID (gen_mid.cpp emits it for the graphs whose plain form SPILLS): through an opaque asm, so that what an op
// derives from it (pixel, channel quad, LDS offsets ...) is recomputed by the next op — a handful of VALU instructions — instead of being kept alive across the
// whole straight-line kernel by common-subexpression elimination.  Left alone the compiler carries dozens of such values through the 100-register depthwise
// bodies: MLKit's kernel sat at 128 registers with 352 bytes of scratch, stores in the first ops and loads all the way down; opaque it needs 92 registers and
// none (round 6: middle kernel -3.5 %).  Where nothing spills the shared values are free and recomputing them only costs issue slots (segm_lite: +6 %), so the
// choice is made per graph from the scratch size of the compiled code object (bsx_api.hip: build_mid_kernel).
#ifdef BSXM_OPAQUE_TID
__device__ __forceinline__ int tid_now() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#else
__device__ __forceinline__ int tid_now() { return (int)threadIdx.x; }
#endif
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(tid_now() >> 6); }

// compile-time address-space loads / stores: one ds_* or global_* instruction, never a flat access and never a branch.  Global offsets are
// UNSIGNED: uniform base + zero-extended 32-bit lane offset is the `global_load … v_off, s[base]` form — a signed offset makes the compiler
// build a 64-bit address pair per access (nine taps = 18 more registers in the depthwise bodies: spills).
template <int SP> __device__ __forceinline__ f4v ld4(const lds_f* l, const glb_f* g, int off) {
  if constexpr (SP == SP_LDS) return *(const lds_v4*)(l + off);
  else if constexpr (SP == SP_GLB16) return __builtin_convertvector(*BSXM_H(const glb_h4, g, off), f4v);
  else return *BSXM_G(const glb_v4, g, off);
}
template <int SP> __device__ __forceinline__ void st4(lds_f* l, glb_f* g, int off, f4v v) {
  if constexpr (SP == SP_LDS) *(lds_v4*)(l + off) = v;
  else if constexpr (SP == SP_GLB16) *BSXM_H(glb_h4, g, off) = __builtin_convertvector(v, h4v);
  else *BSXM_G(glb_v4, g, off) = v;
}
template <int SP> __device__ __forceinline__ f2v ld2(const lds_f* l, const glb_f* g, int off) {
  if constexpr (SP == SP_LDS) return *(const lds_v2*)(l + off);
  else if constexpr (SP == SP_GLB16) return __builtin_convertvector(*BSXM_H(const glb_h2, g, off), f2v);
  else return *BSXM_G(const glb_v2, g, off);
}
template <int SP> __device__ __forceinline__ void st2(lds_f* l, glb_f* g, int off, f2v v) {
  if constexpr (SP == SP_LDS) *(lds_v2*)(l + off) = v;
  else if constexpr (SP == SP_GLB16) *BSXM_H(glb_h2, g, off) = __builtin_convertvector(v, h2v);
  else *BSXM_G(glb_v2, g, off) = v;
}
template <int SP> __device__ __forceinline__ float ld1(const lds_f* l, const glb_f* g, int off) {
  if constexpr (SP == SP_LDS) return l[off];
  else if constexpr (SP == SP_GLB16) return (float)*BSXM_H(const glb_h, g, off);
  else return *BSXM_G(const glb_f, g, off);
}
template <int SP> __device__ __forceinline__ void st1(lds_f* l, glb_f* g, int off, float v) {
  if constexpr (SP == SP_LDS) l[off] = v;
  else if constexpr (SP == SP_GLB16) *BSXM_H(glb_h, g, off) = (_Float16)v;
  else *BSXM_G(glb_f, g, off) = v;
}

// activations: hardware exp2 / rcp (≈1 ulp), the same forms as the interpreter (kernels_frame.hip: fp_act)
template <int ACT> __device__ __forceinline__ float act1(float v) {
  if constexpr (ACT == ACT_NONE) return v;
  else if constexpr (ACT == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
  else if constexpr (ACT == ACT_HSWISH) return v * __builtin_fminf(6.f, __builtin_fmaxf(0.f, v + 3.f)) * 0.16666667163372040f;
  else if constexpr (ACT == ACT_RELU) return __builtin_fmaxf(v, 0.f);
  else return __builtin_fminf(__builtin_fmaxf(v, 0.f), 6.f);
}
template <int ACT> __device__ __forceinline__ f4v act4(f4v v) { f4v r = {act1<ACT>(v.x), act1<ACT>(v.y), act1<ACT>(v.z), act1<ACT>(v.w)}; return r; }
template <int ACT> __device__ __forceinline__ f2v act2(f2v v) { f2v r = {act1<ACT>(v.x), act1<ACT>(v.y)}; return r; }

// ---- DPP helpers ---------------------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRor4 = 0x124, kRor8 = 0x128, kHalfMirror = 0x141;
// float4 per lane → wave total of component (lane & 3) in every lane (kernels_frame.hip: wave_total_scatter)
__device__ __forceinline__ float wave_total_scatter(f4v a, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  const float klo = (b0 ? a.y : a.x) + dpp<kQuadXor1>(b0 ? a.x : a.y);
  const float khi = (b0 ? a.w : a.z) + dpp<kQuadXor1>(b0 ? a.z : a.w);
  float v = (b1 ? khi : klo) + dpp<kQuadXor2>(b1 ? klo : khi);
  v += dpp<kRor4>(v);
  v += dpp<kRor8>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ---- weight staging: asynchronous global → LDS DMA (global_load_lds_dwordx4, tracked by vmcnt) ---------------------------------
// One wave instruction moves 64 lanes x 16 B; LDS address = M0 (wave-uniform) + lane * 16.  FLOATS and both addresses are compile-time.
template <int FLOATS> __device__ __forceinline__ void stage(const glb_f* src, lds_f* dst) {
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  if constexpr (FLOATS > 0) {
    const int lane4 = (tid_now() & 63) * 4, w = wave_id();
    constexpr int CH = (FLOATS + 255) / 256;                      // 1 KiB chunks
#pragma unroll
    for (int i = 0; i < (CH + kWaves - 1) / kWaves; i++) {
      const int c0 = (w + i * kWaves) * 256;
      if (c0 < FLOATS && c0 + lane4 < FLOATS) __builtin_amdgcn_global_load_lds((glb_vp)(src + c0 + lane4), (lds_vp)(dst + c0), 16, 0, 0);
    }
  }
}
// every op starts here: the previous op's LDS writes and this op's staged weights (DMA issued an op earlier) are visible
__device__ __forceinline__ void op_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---- 1x1 convolution on v_mfma_f32_16x16x4_f32 (exact f32) ----------------------------------------------------------------------
// Wave = 16-pixel x 16-channel tile.  Operand maps: A lane (li, g) holds x[pixel m0 + li][16 j + 4 g .. +3] (one 16-byte load per
// 16 input channels, its four components feed four successive MFMAs); B lane holds w[16 j + 4 g + r][n0 + li]; D lane holds pixels
// m0 + 4 g + r of channel n0 + li.  A K tail of 8 (K = 24, 72, 88 …) runs as TWO MFMAs with lane g holding k = base + 2 g + r instead
// of four half-empty ones.  All operands of a tile are requested before the first MFMA; two accumulator chains alternate (a
// dependent f32 MFMA has 40 cycles of latency against 32 of issue).
// Traits: P CIN COUT CPAD ACT | X_SP X_OFF X_ST | Y_SP Y_OFF Y_ST | R_SP R_OFF R_ST | S_OFF (LDS, -1: none) | D_SP D_OFF D_ST | W_LDS B_LDS
template <class T> __device__ __forceinline__ void op_pw(lds_f* L, glb_f* A) {
  // N0 / NCOLS: the column range [N0, N0 + NCOLS) of the layer this call computes (the whole layer, or one channel chunk written straight into the LDS
  // workspace of the depthwise that consumes it — YSUB = N0 then rebases the stored channel index); CPAD stays the row stride of the weight block
  constexpr int P = T::P, CIN = T::CIN, CPAD = T::CPAD, MT = (P + 15) / 16, NT = T::NCOLS / 16, TILES = MT * NT;
  constexpr int NJ = CIN / 16, TAIL = CIN % 16, TM = TAIL / 4;      // TM = MFMAs of the tail (0..3)
  static_assert(CIN % 4 == 0 && CPAD % 16 == 0, "op_pw: channel counts");
  const int lane = tid_now() & 63, li = lane & 15, g = lane >> 4, wave = wave_id();
  const lds_f* wl = L + T::W_LDS;
  const lds_f* bl = L + T::B_LDS;
#pragma unroll
  for (int it = 0; it < (TILES + kWaves - 1) / kWaves; it++) {
    const int wi = wave + it * kWaves;
    if (wi >= TILES) break;
    // tile order.  LDS output: M-tile fastest (the 16 waves of a round share one B tile).  Output in the ARENA (a tensor that does not fit LDS): N-tile
    // fastest (T::NFAST) — the waves of a round then write ADJACENT 64-byte pieces of the same 16 pixel rows at the same time, which the L2 merges into
    // whole lines, instead of sixteen isolated 64-byte pieces 4 * COUT bytes apart each; the shared A rows are read once per round.
    const int tn = T::NFAST ? wi % NT : wi / MT, tm = T::NFAST ? wi / NT : wi - tn * MT, m0 = tm << 4, n0 = T::N0 + (tn << 4);
    const int arow = (P % 16 == 0) ? m0 + li : min(m0 + li, P - 1);      // rows past the end read a valid pixel; results are dropped
    const int xo = arow * T::X_ST + 4 * g;
    const lds_f* bp = wl + (4 * g) * CPAD + n0 + li;
    f4v a[NJ > 0 ? NJ : 1];
    float b[NJ > 0 ? NJ : 1][4];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      a[j] = ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, xo + 16 * j);
      const lds_f* br = bp + (16 * j) * CPAD;
      b[j][0] = br[0]; b[j][1] = br[CPAD]; b[j][2] = br[2 * CPAD]; b[j][3] = br[3 * CPAD];
    }
    // K tail: lane g holds k = 16 NJ + TM g + r, r < TM
    float ta[3] = {0.f, 0.f, 0.f}, tb[3] = {0.f, 0.f, 0.f};
    if constexpr (TM > 0) {
      const int k0 = 16 * NJ + TM * g;
#pragma unroll
      for (int r = 0; r < TM; r++) { ta[r] = ld1<T::X_SP>(L + T::X_OFF, A + T::X_OFF, arow * T::X_ST + k0 + r); tb[r] = wl[(k0 + r) * CPAD + n0 + li]; }
    }
    if constexpr (T::S_OFF >= 0) {                                   // squeeze-excite scale on the input channels (always an LDS vector)
#pragma unroll
      for (int j = 0; j < NJ; j++) { const f4v sv = *(const lds_v4*)(L + T::S_OFF + 16 * j + 4 * g); a[j].x = __fmul_rn(a[j].x, sv.x); a[j].y = __fmul_rn(a[j].y, sv.y); a[j].z = __fmul_rn(a[j].z, sv.z); a[j].w = __fmul_rn(a[j].w, sv.w); }
      if constexpr (TM > 0) {
#pragma unroll
        for (int r = 0; r < TM; r++) ta[r] = __fmul_rn(ta[r], L[T::S_OFF + 16 * NJ + TM * g + r]);
      }
    }
    if constexpr (T::D_SP != SP_NONE) {                              // x * s + d  (gate * skip + up-sampled tensor)
#pragma unroll
      for (int j = 0; j < NJ; j++) { const f4v dv = ld4<T::D_SP>(L + T::D_OFF, A + T::D_OFF, arow * T::D_ST + 16 * j + 4 * g); a[j].x = __fadd_rn(a[j].x, dv.x); a[j].y = __fadd_rn(a[j].y, dv.y); a[j].z = __fadd_rn(a[j].z, dv.z); a[j].w = __fadd_rn(a[j].w, dv.w); }
      if constexpr (TM > 0) {
#pragma unroll
        for (int r = 0; r < TM; r++) ta[r] = __fadd_rn(ta[r], ld1<T::D_SP>(L + T::D_OFF, A + T::D_OFF, arow * T::D_ST + 16 * NJ + TM * g + r));
      }
    }
    f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#if defined(BSXM_EXP_MFMA_QUARTER)
    // TIMING EXPERIMENT ONLY (BSX_RTC_EXP_MFMA=1, results are wrong): one of every four MFMAs of the K loop, every load / scale / epilogue unchanged — an UPPER bound on
    // what a split-f16 v_mfma_f32_16x16x32_f16 form of this op (3 f16 MFMAs where 8 f32 ones are, plus the operand split) could gain
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][0] + b[j][1] + b[j][2] + b[j][3], a[j].x + a[j].y + a[j].z + a[j].w, acc0, 0, 0, 0);
      if (j + 1 < NJ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j + 1 < NJ ? j + 1 : j][0] + b[j + 1 < NJ ? j + 1 : j][1] + b[j + 1 < NJ ? j + 1 : j][2] + b[j + 1 < NJ ? j + 1 : j][3],
                                                                  a[j + 1 < NJ ? j + 1 : j].x + a[j + 1 < NJ ? j + 1 : j].y + a[j + 1 < NJ ? j + 1 : j].z + a[j + 1 < NJ ? j + 1 : j].w, acc1, 0, 0, 0);
    }
#else
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][0], a[j].x, acc0, 0, 0, 0);
      if (j + 1 < NJ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j + 1 < NJ ? j + 1 : j][0], a[j + 1 < NJ ? j + 1 : j].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][1], a[j].y, acc0, 0, 0, 0);
      if (j + 1 < NJ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j + 1 < NJ ? j + 1 : j][1], a[j + 1 < NJ ? j + 1 : j].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][2], a[j].z, acc0, 0, 0, 0);
      if (j + 1 < NJ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j + 1 < NJ ? j + 1 : j][2], a[j + 1 < NJ ? j + 1 : j].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][3], a[j].w, acc0, 0, 0, 0);
      if (j + 1 < NJ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j + 1 < NJ ? j + 1 : j][3], a[j + 1 < NJ ? j + 1 : j].w, acc1, 0, 0, 0);
    }
#endif
    if constexpr (TM > 0) {
#pragma unroll
      for (int r = 0; r < TM; r++) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tb[r], ta[r], acc1, 0, 0, 0);
    }
    const f4v acc = acc0 + acc1;
    // epilogue: the weights are the MFMA's A operand (rows = output channels), the activations its B operand (columns = pixels), so the accumulator of lane
    // (li, g) IS channels n0 + 4 g .. +3 of pixel m0 + li → one 16-byte bias load, residual load, store, and no 4x4 transpose inside lane quads
    const int c0 = n0 + 4 * g, pix = m0 + li;
    f4v v = acc;
    if ((T::COUT % 16 == 0 || c0 < T::COUT) && (P % 16 == 0 || pix < P)) {
      v = act4<T::ACT>(v + *(const lds_v4*)(bl + c0));
      if constexpr (T::R_SP != SP_NONE) v += ld4<T::R_SP>(L + T::R_OFF, A + T::R_OFF, pix * T::R_ST + c0);
      st4<T::Y_SP>(L + T::Y_OFF, A + T::Y_OFF, pix * T::Y_ST + c0 - T::YSUB, v);
    }
  }
}

// ---- channels [C0, C0 + CK) of an arena tensor [P][CFULL] → LDS workspace [P][CK + 4] (one coalesced pass: CK / 4 lanes read one pixel's 4 CK contiguous bytes) ----
// A depthwise whose input lives in the arena reads every element ~10 times (K output rows x overlapping strips) through an L2 that 32 frames share; with the
// chunk staged here each element leaves memory ONCE and the taps run from LDS.  All of a lane's loads are in flight before its first LDS store.
template <int SP, int X_OFF, int P, int CFULL, int CK, int C0, int WS, int WST = CK + 4>      // WST: row stride of the workspace (the full chunk's CK + 4 also for a ragged last chunk)
__device__ __forceinline__ void load_chunk(lds_f* L, const glb_f* A) {
  constexpr int Q = CK / 4, TOTAL = P * Q, IT = (TOTAL + kThreads - 1) / kThreads;
  f4v v[IT];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int i = min(tid_now() + it * kThreads, TOTAL - 1), px = i / Q, q = i - px * Q;
    v[it] = ld4<SP>(L, A + X_OFF, px * CFULL + C0 + 4 * q);
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int i = tid_now() + it * kThreads, px = i / Q, q = i - px * Q;
    if (i < TOTAL) *(lds_v4*)(L + WS + px * WST + 4 * q) = v[it];
  }
}

// ---- depthwise k x k, register-strip form -----------------------------------------------------------------------------------------
// lane = (channel pair / quad, strip of TX consecutive output columns, output row): per filter row the lane loads the (TX-1) S + K
// inputs its strip touches and the K weights once and forms TX outputs from registers (kernels_frame.hip: dw_strip).  Everything
// — the item decomposition included — is compile-time here; rows and columns outside the image contribute exact zeros.
// FMA order per output: fy, fx ascending, bias last (TFLite reference order).
// Traits: K S H W OH OW PT PL C ACT | X_SP X_OFF X_ST | Y_SP Y_OFF Y_ST | R_SP R_OFF R_ST | W_SP (LDS staged / GLB) W_OFF B_OFF | V TX | CW YC0
// (CW = channels of the WHOLE layer = row stride of the weight block, YC0 = first output channel: C < CW when the op is one channel chunk of a layer whose input
//  is staged through LDS chunk by chunk — load_chunk below)
template <class T> __device__ __forceinline__ void op_dw(lds_f* L, glb_f* A, const glb_f* Wg) {
  constexpr int K = T::K, S = T::S, TX = T::TX, V = T::V, C = T::C, CV = C / V, NIN = (TX - 1) * S + K;
  constexpr int NSTRIPS = (T::OW + TX - 1) / TX, TOTAL = CV * NSTRIPS * T::OH;
  typedef float vec_t __attribute__((ext_vector_type(V)));
  // ZC: out-of-image taps are read from the zero cell — in the fully unrolled LDS form, where the generator found register room for it (T::ZC: gen_mid.cpp)
  constexpr bool ZC = T::ZC && T::X_SP == SP_LDS;
  auto ldx = [&](int off) -> vec_t { if constexpr (V == 4) return ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, off); else return ld2<T::X_SP>(L + T::X_OFF, A + T::X_OFF, off); };
  auto ldw = [&](int off) -> vec_t { if constexpr (V == 4) return ld4<T::W_SP>(L, Wg, off); else return ld2<T::W_SP>(L, Wg, off); };
#pragma unroll
  for (int it = 0; it < (TOTAL + kThreads - 1) / kThreads; it++) {
    const int item = tid_now() + it * kThreads;
    if (item >= TOTAL) break;
    const int t = item / CV, cq = item - t * CV, oy = t / NSTRIPS, sx = t - oy * NSTRIPS;
    const int ch = cq * V, ox0 = sx * TX, ix0 = ox0 * S - T::PL;
    vec_t acc[TX];
#pragma unroll
    for (int k = 0; k < TX; k++) acc[k] = (vec_t)(0.f);
    // one filter row: the (TX-1) S + K inputs of the strip (clamped addresses, zeroed outside the image) and the K weights (zeroed when the row is outside)
    auto load_row = [&](int fy, vec_t (&xin)[NIN], vec_t (&wv)[K]) {
      const int iy = oy * S - T::PT + fy;
      const bool vy = iy >= 0 && iy < T::H;
      if constexpr (ZC) {
        // LDS input: an out-of-image tap reads the zero cell (same +0.0f the select on the data produced: bit-identical); nothing is clamped, nothing zeroed afterwards
        // (one base per filter row; the column step j * X_ST is a compile-time term the ds_read carries as its immediate offset — the select picks between the row's
        //  base and "zero cell minus that term", so a tap costs one v_cndmask and no address arithmetic)
        const int rowo = (iy * T::W + ix0) * T::X_ST + ch;
#pragma unroll
        for (int j = 0; j < NIN; j++) { const int ix = ix0 + j; xin[j] = ldx(((vy && ix >= 0 && ix < T::W) ? rowo : kZeroOff - T::X_OFF - j * T::X_ST) + j * T::X_ST); }
#pragma unroll
        for (int fx = 0; fx < K; fx++) wv[fx] = ldw(T::W_OFF + (fy * K + fx) * T::CW + ch);
      } else {
        const int rowo = min(max(iy, 0), T::H - 1) * T::W * T::X_ST + ch;
#pragma unroll
        for (int j = 0; j < NIN; j++) xin[j] = ldx(rowo + min(max(ix0 + j, 0), T::W - 1) * T::X_ST);
#pragma unroll
        for (int fx = 0; fx < K; fx++) { wv[fx] = ldw(T::W_OFF + (fy * K + fx) * T::CW + ch); if (!vy) wv[fx] = (vec_t)(0.f); }
      }
    };
    auto fma_row = [&](vec_t (&xin)[NIN], const vec_t (&wv)[K]) {
      if constexpr (!ZC) {
#pragma unroll
        for (int j = 0; j < NIN; j++) { const int ix = ix0 + j; if (ix < 0 || ix >= T::W) xin[j] = (vec_t)(0.f); }
      }
#pragma unroll
      for (int k = 0; k < TX; k++) {
#pragma unroll
        for (int fx = 0; fx < K; fx++) acc[k] = __builtin_elementwise_fma(xin[k * S + fx], wv[fx], acc[k]);
      }
    };
    if constexpr (T::X_SP == SP_GLB || T::X_SP == SP_GLB16) {
      // global input (a tensor that does not fit LDS): exactly ONE row ahead in flight, in a real loop with a register copy per row.  Measured
      // alternatives: two alternating register sets without copies (100-121 registers: spills in the whole kernel, segm_full program 213 -> 404 us),
      // every row requested up front over a short strip (126 registers, spills everywhere).  These ops are bound by the HBM round trip per row;
      // the real fix is not to have such tensors in HBM (a tiled level-3 kernel — DESIGN.md, "next").
      vec_t xa[NIN], wa[K];
      load_row(0, xa, wa);
#pragma unroll 1
      for (int fy = 0; fy < K; fy++) {
        vec_t xb[NIN], wb[K];
        load_row(fy + 1 < K ? fy + 1 : fy, xb, wb);               // the last trip re-requests its own row (cached) instead of branching
        fma_row(xa, wa);
#pragma unroll
        for (int j = 0; j < NIN; j++) xa[j] = xb[j];
#pragma unroll
        for (int fx = 0; fx < K; fx++) wa[fx] = wb[fx];
      }
    } else if constexpr (K * (NIN + K) * V <= 144) {
      // LDS input: fully unrolled — the scheduler overlaps the next rows' ds_reads with this row's FMAs (the 5x5 stride-2 form would hold
      // 5 x 16 x 2 registers that way and spills: it takes the one-row-ahead loop below)
#pragma unroll
      for (int fy = 0; fy < K; fy++) { vec_t xin[NIN], wv[K]; load_row(fy, xin, wv); fma_row(xin, wv); }
    } else {
      // exactly ONE row ahead in flight (all K rows at once — what plain unrolling turns into — needs K (NIN + K) V registers: spills).
      // Two register sets alternate by filter-row parity.
      vec_t x0[NIN], w0[K], x1[NIN], w1[K];
      static_assert(K % 2 == 1, "op_dw: odd filter sizes");
      load_row(0, x0, w0);
#pragma unroll 1
      for (int fy = 0; fy + 1 < K; fy += 2) {                       // a real loop: nothing is hoisted across its back edge, no register copies
        load_row(fy + 1, x1, w1);
        fma_row(x0, w0);
        load_row(fy + 2, x0, w0);
        fma_row(x1, w1);
      }
      fma_row(x0, w0);
    }
    const vec_t bq = ldw(T::B_OFF + ch);
#pragma unroll
    for (int k = 0; k < TX; k++) {
      const int ox = ox0 + k;
      if (T::OW % TX == 0 || ox < T::OW) {
        const int pix = oy * T::OW + ox, cho = ch + T::YC0;
        vec_t v = acc[k] + bq;
        if constexpr (V == 4) v = act4<T::ACT>(v); else v = act2<T::ACT>(v);
        if constexpr (T::R_SP != SP_NONE) { if constexpr (V == 4) v += ld4<T::R_SP>(L + T::R_OFF, A + T::R_OFF, pix * T::R_ST + ch); else v += ld2<T::R_SP>(L + T::R_OFF, A + T::R_OFF, pix * T::R_ST + ch); }
        if constexpr (V == 4) st4<T::Y_SP>(L + T::Y_OFF, A + T::Y_OFF, pix * T::Y_ST + cho, v); else st2<T::Y_SP>(L + T::Y_OFF, A + T::Y_OFF, pix * T::Y_ST + cho, v);
      }
    }
  }
}

// ---- global average pool of one part into mean[COFF .. COFF + C): wave-owned channel quads, no LDS scratch, no barrier inside ------
// rows = pixels of a tensor, or per-tile partial sums written by a segment kernel ([ROWS][C] floats, mean = sum / HW)
template <int SP, int OFF, int ST, int ROWS, int C, int HW, int COFF, bool ACCUM, int MEAN_OFF>
__device__ __forceinline__ void gap_part(lds_f* L, glb_f* A) {
  constexpr int C4 = C / 4;
  const int lane = tid_now() & 63, wave = wave_id();
#pragma unroll
  for (int it = 0; it < (C4 + kWaves - 1) / kWaves; it++) {
    const int cq = wave + it * kWaves;
    if (cq >= C4) break;
    f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += 128) {
      const int r = r0 + lane;
      if (r < ROWS) a0 += ld4<SP>(L + OFF, A + OFF, r * ST + cq * 4);
      if (r0 + 64 < ROWS && r + 64 < ROWS) a1 += ld4<SP>(L + OFF, A + OFF, (r + 64) * ST + cq * 4);
    }
    a0 += a1;
    float t = wave_total_scatter(a0, lane);
    if (lane < 4) {                                                 // lane e holds the total of channel 4 cq + e
      t /= (float)HW;
      if (ACCUM) t += L[MEAN_OFF + COFF + cq * 4 + lane];            // GAP(a + b) as GAP(a) + GAP(b): the same lane wrote the first part
      L[MEAN_OFF + COFF + cq * 4 + lane] = t;
    }
  }
}

// ---- fully connected layer of a squeeze-excite / gate chain: each output = dot product over LK aligned lanes + DPP reduction ------
// weights [co][ci] rows with the bias in front, either staged in LDS (W_SP = LDS: offsets into the LDS block) or in the weight arena.
// Two halves: fc_load requests a lane's weight slice (registers) — called at the TOP of the squeeze-excite op, so that the L2 round trip of
// both layers' weights hides behind the pooling and its barrier — and fc_apply does the arithmetic once the input vector is complete.
// Eight lanes per output whenever the input splits into equal slices of whole float4s (KLEN = 4 ceil(CIN / 32): 72 = 6 x 12, 24 = 6 x 4 — the
// lanes past the last slice add zeros), otherwise the largest power-of-two split.  (A 72-channel layer on 2 lanes per output kept 144 of the
// 1024 lanes busy with 36 MACs each.)
template <int CIN, int COUT> struct FcGeom {
  static constexpr int K8 = 4 * ((CIN + 31) / 32);
  static constexpr bool EIGHT = CIN % K8 == 0;
  static constexpr int LK = EIGHT ? 8 : ((CIN % 16 == 0) ? 4 : ((CIN % 8 == 0) ? 2 : 1)), KLEN = EIGHT ? K8 : CIN / LK, NSUB = CIN / KLEN;   // NSUB <= LK active lanes
  static constexpr int PER = kThreads / LK, IT = (COUT + PER - 1) / PER, Q = KLEN / 4;
};
template <int CIN, int COUT> struct FcRegs { f4v w[FcGeom<CIN, COUT>::IT][FcGeom<CIN, COUT>::Q]; float b[FcGeom<CIN, COUT>::IT]; };
template <int CIN, int COUT, int W_SP, int W_OFF, int B_OFF>
__device__ __forceinline__ void fc_load(const lds_f* L, const glb_f* Wg, FcRegs<CIN, COUT>& r) {
  typedef FcGeom<CIN, COUT> G;
  const int sub = tid_now() & (G::LK - 1), co0 = tid_now() / G::LK;
  // The weights are read-only memory: left alone, the compiler hoists these loads above the barrier into the op in front — wherever IT likes,
  // e.g. into a 91-register depthwise body, which then spills.  Where the loads go is the generator's decision (gen_mid.cpp: `early`): the
  // base pointer passes through an opaque asm here, so they stay behind whatever precedes this call.
  asm volatile("" : "+s"(Wg));
#pragma unroll
  for (int it = 0; it < G::IT; it++) {
    const int co = min(co0 + it * G::PER, COUT - 1);               // lanes past the last output re-read a valid row; their result is dropped
    const int ks = min(sub, G::NSUB - 1) * G::KLEN;                // lanes past the last slice re-read the last one; fc_apply zeroes their input
#pragma unroll
    for (int q = 0; q < G::Q; q++) r.w[it][q] = ld4<W_SP>(L, Wg, W_OFF + co * CIN + ks + 4 * q);
    r.b[it] = ld1<W_SP>(L, Wg, B_OFF + co);
  }
}
template <int CIN, int COUT, int ACT, int X_OFF, int Y_SP, int Y_OFF>
__device__ __forceinline__ void fc_apply(lds_f* L, glb_f* A, const FcRegs<CIN, COUT>& r) {
  typedef FcGeom<CIN, COUT> G;
  const int sub = tid_now() & (G::LK - 1), co0 = tid_now() / G::LK;
  f4v xv[G::Q];
#pragma unroll
  for (int q = 0; q < G::Q; q++) {
    xv[q] = *(const lds_v4*)(L + X_OFF + min(sub, G::NSUB - 1) * G::KLEN + 4 * q);
    if (G::NSUB < G::LK && sub >= G::NSUB) xv[q] = (f4v)(0.f);
  }
#pragma unroll
  for (int it = 0; it < G::IT; it++) {
    const int co = co0 + it * G::PER;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < G::Q; q++) {
      a0 = fmaf(xv[q].x, r.w[it][q].x, a0); a1 = fmaf(xv[q].y, r.w[it][q].y, a1); a0 = fmaf(xv[q].z, r.w[it][q].z, a0); a1 = fmaf(xv[q].w, r.w[it][q].w, a1);
    }
    float acc = a0 + a1;
    if constexpr (G::LK >= 2) acc += dpp<kQuadXor1>(acc);
    if constexpr (G::LK >= 4) acc += dpp<kQuadXor2>(acc);
    if constexpr (G::LK >= 8) acc += dpp<kHalfMirror>(acc);
    if (sub == 0 && co < COUT) st1<Y_SP>(L + Y_OFF, A + Y_OFF, co, act1<ACT>(acc + r.b[it]));
  }
}

// ---- bilinear resize (TFLite reference association) ------------------------------------------------------------------------------------
// Traits: H W OH OW C HALF_PIXEL ALIGN | X_SP X_OFF X_ST | Y_SP Y_OFF Y_ST
__device__ __forceinline__ void interp(int o, float scale, bool half_pixel, int in_size, float* frac, int* lo, int* hi) {
  const float v = half_pixel ? __fadd_rn(__fmul_rn((float)o + 0.5f, scale), -0.5f) : __fmul_rn((float)o, scale);
  const float fl = __builtin_floorf(v);
  *lo = max((int)fl, 0);
  *hi = min((int)__builtin_ceilf(v), in_size - 1);
  *frac = v - (float)*lo;
}
__device__ __forceinline__ float bilerp(float x00, float x10, float x01, float x11, float dy, float dx) {
  const float a = __fmul_rn(__fmul_rn(x00, 1.f - dy), 1.f - dx), b = __fmul_rn(__fmul_rn(x10, dy), 1.f - dx);
  const float c = __fmul_rn(__fmul_rn(x01, 1.f - dy), dx), d = __fmul_rn(__fmul_rn(x11, dy), dx);
  return __fadd_rn(__fadd_rn(__fadd_rn(a, b), c), d);
}
template <class T> __device__ __forceinline__ void op_resize(lds_f* L, glb_f* A) {
  constexpr int CV = T::C / 4, TOTAL = T::OH * T::OW * CV;
  const float hs = (T::ALIGN && T::OH > 1) ? (float)(T::H - 1) / (float)(T::OH - 1) : (float)T::H / (float)T::OH;
  const float ws = (T::ALIGN && T::OW > 1) ? (float)(T::W - 1) / (float)(T::OW - 1) : (float)T::W / (float)T::OW;
#pragma unroll
  for (int it = 0; it < (TOTAL + kThreads - 1) / kThreads; it++) {
    const int item = tid_now() + it * kThreads;
    if (item >= TOTAL) break;
    const int p = item / CV, ch = (item - p * CV) * 4, oy = p / T::OW, ox = p - oy * T::OW;
    float dy, dx; int y0, y1, x0, x1;
    interp(oy, hs, T::HALF_PIXEL, T::H, &dy, &y0, &y1);
    interp(ox, ws, T::HALF_PIXEL, T::W, &dx, &x0, &x1);
    const f4v a = ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, (y0 * T::W + x0) * T::X_ST + ch), b = ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, (y1 * T::W + x0) * T::X_ST + ch);
    const f4v c = ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, (y0 * T::W + x1) * T::X_ST + ch), d = ld4<T::X_SP>(L + T::X_OFF, A + T::X_OFF, (y1 * T::W + x1) * T::X_ST + ch);
    const f4v v = {bilerp(a.x, b.x, c.x, d.x, dy, dx), bilerp(a.y, b.y, c.y, d.y, dy, dx), bilerp(a.z, b.z, c.z, d.z, dy, dx), bilerp(a.w, b.w, c.w, d.w, dy, dx)};
    st4<T::Y_SP>(L + T::Y_OFF, A + T::Y_OFF, p * T::Y_ST + ch, v);
  }
}

}  // namespace bsxm
