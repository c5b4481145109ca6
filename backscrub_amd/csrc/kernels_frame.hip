// kernels_frame.hip — the per-frame network program (see frame_program.hpp).
//
// One 1024-lane workgroup = one camera frame = one CU.  The workgroup interprets the fused step
// list; tensors live in the CU's LDS ([pixel][C+pad] rows) unless the planner spilled them to the
// frame's private slice of the HBM arena.
//
// gfx950 notes that shaped this file (all measured with tools/program_timeline.py):
//  * generic ("flat") loads that land in LDS run at a small fraction of the ds_read rate, so every
//    operand is accessed through an address-space-typed reference (`Ref`): ds_read/ds_write for
//    LDS tensors, global_load/store for spilled ones, chosen by a wave-uniform branch;
//  * a layer's weights are read by all 16 waves — they are staged ONCE per op into an LDS slot the
//    planner reserved, by an asynchronous global→LDS DMA (global_load_lds_dwordx4) issued while the
//    PREVIOUS op runs; only layers whose block gets no slot use wave-uniform scalar loads (constant
//    address space → s_load + SGPR-operand FMAs);
//  * hot loops that gather several operands are templated on the operand's address space: a run-time
//    LDS-or-global test per load keeps the compiler from issuing the loads back to back, and at 4 waves
//    per SIMD every exposed memory round trip is paid in full;
//  * 1x1 convolutions run on v_mfma_f32_16x16x4_f32 (exact f32); the accumulator tile is transposed
//    inside lane quads (mfma_tile.hpp) so that every lane stores 16 contiguous bytes;
//  * the micro-op table itself is read through the constant address space so that dims/offsets stay
//    in SGPRs and every branch on them is scalar.
//
// Numerics: f32 FMA chains with the bias added last (the MFMA forms sum k in a fixed, different order
// than ci-ascending), un-contracted bilinear taps; the single-pixel GEMV steps and the global average
// pools use tree reductions.  Measured against the oracle: max relative logit error < 1e-5.
#include "debug_switches.hpp"
#include <cstdlib>

#include "frame_program.hpp"
#include "kernels.hpp"
#include "mfma_tile.hpp"

namespace bsx {
namespace {

extern __shared__ __attribute__((aligned(16))) float smem[];

typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(1))) float glb_f;
// HIP's float4 is a class; address-space-qualified accesses use the builtin vector type underneath
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4v lds_v4;
typedef __attribute__((address_space(1))) f4v glb_v4;
typedef __attribute__((address_space(4))) const float cfloat_t;
typedef __attribute__((address_space(4))) const MicroOp cop_t;
typedef __attribute__((address_space(4))) const Loc cloc_t;

__device__ __forceinline__ cfloat_t* as_const(const float* p) { return (cfloat_t*)p; }
__device__ __forceinline__ lds_f* lds_base() { return (lds_f*)smem; }
__device__ __forceinline__ float4 ld_lds4(const lds_f* p) { f4v v = *(const lds_v4*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ld_glb4(const glb_f* p) { f4v v = *(const glb_v4*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_lds4(lds_f* p, float4 v) { f4v t = {v.x, v.y, v.z, v.w}; *(lds_v4*)p = t; }
__device__ __forceinline__ void st_glb4(glb_f* p, float4 v) { f4v t = {v.x, v.y, v.z, v.w}; *(glb_v4*)p = t; }

struct FrameCtx {
  float* arena;        // this frame's private slice
  float* net_in;       // batch-major network input  [n][inH][inW][inC]
  float* net_out;      // batch-major network output [n][outH][outW][outC]
  const float* weights;
  int frame;
  unsigned long long* tl;   // debug timeline (nullptr normally); slots 256.. are free for sub-phase accumulators
};

// Address-space-typed tensor reference.  `lds` is wave-uniform, so `ld4`/`st4` compile to one scalar
// branch around a ds_* or a global_* instruction — never a flat access.
struct Ref {
  lds_f* l;
  glb_f* g;
  int stride;
  bool lds, valid;
};

__device__ __forceinline__ Ref make_ref(cloc_t& loc, const FrameCtx& c) {
  Ref r;
  const int space = loc.space, off = loc.off;
  r.stride = loc.stride;
  r.valid = space != kLocNone;
  r.lds = space == kLocLds;
  r.l = lds_base() + off;
  float* g = c.arena + off;
  if (space == kLocInput) g = c.net_in + (size_t)c.frame * (size_t)loc.elems;
  if (space == kLocOutput) g = c.net_out + (size_t)c.frame * (size_t)loc.elems;
  r.g = (glb_f*)g;
  return r;
}
__device__ __forceinline__ float4 ld4(const Ref& r, int off) {
  if (r.lds) return ld_lds4(r.l + off);
  return ld_glb4(r.g + off);
}
__device__ __forceinline__ void st4(const Ref& r, int off, float4 v) {
  if (r.lds) st_lds4(r.l + off, v); else st_glb4(r.g + off, v);
}
// weight fetch: staged LDS copy or the global original (wave-uniform choice)
__device__ __forceinline__ float4 ldw4(bool staged, const lds_f* l, const glb_f* g, int off) {
  if (staged) return ld_lds4(l + off);
  return ld_glb4(g + off);
}
__device__ __forceinline__ float ld1(const Ref& r, int off) { return r.lds ? r.l[off] : r.g[off]; }
__device__ __forceinline__ void st1(const Ref& r, int off, float v) { if (r.lds) r.l[off] = v; else r.g[off] = v; }

// Activations use the hardware exp2/rcp instructions (≈1 ulp) instead of the libm / IEEE-divide expansions: this
// kernel runs every op exactly once per CU, so its speed floor is instruction-cache misses — code size matters more
// than ALU work.  (The per-launch kernels in kernels_nn.hip keep the exact forms; both meet the 1e-4 logit bar.)
__device__ __forceinline__ float fp_act(float v, int act) {
  if (act == kActNone) return v;
  if (act == kActSigmoid) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
  const float hi = act == kActRelu ? 3.0e38f : 6.f;
  if (act == kActHswish) return v * fminf(hi, fmaxf(0.f, v + 3.f)) * 0.16666667163372040f;
  return fminf(fmaxf(v, 0.f), hi);
}
__device__ __forceinline__ float4 fp_act4(float4 v, int act) {
  return make_float4(fp_act(v.x, act), fp_act(v.y, act), fp_act(v.z, act), fp_act(v.w, act));
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- 1x1 convolution: wave = (64-pixel chunk, CT-channel tile), lane = pixel ------------------------------
template <int CT, bool STAGED>
__device__ __forceinline__ void pw_quad(float (&acc)[CT], const float4 xv, const lds_f* wl, cfloat_t* wc, int cout_pad) {
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if constexpr (STAGED) {
      float wr[CT];
      if constexpr (CT >= 4) {
#pragma unroll
        for (int t = 0; t < CT; t += 4) {
          float4 v = ld_lds4(wl + r * cout_pad + t);
          wr[t] = v.x; wr[t + 1] = v.y; wr[t + 2] = v.z; wr[t + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < CT; t++) wr[t] = wl[r * cout_pad + t];
      }
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xs[r], wr[t], acc[t]);
    } else {
      cfloat_t* w0 = wc + (size_t)r * cout_pad;     // wave-uniform → s_load, SGPR-operand FMAs
#pragma unroll
      for (int t = 0; t < CT; t++) acc[t] = fmaf(xs[r], w0[t], acc[t]);
    }
  }
}

template <int CT, bool STAGED, bool XL>
__device__ __forceinline__ void pw_body(cop_t& op, const FrameCtx& c) {
  constexpr int kPF = 2;
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int P = op.OH * op.OW, chunks = (P + 63) >> 6, tiles = op.cout_pad / CT;
  const int lane = threadIdx.x & 63, nw = kFrameThreads >> 6;
  const int Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act, nq = Cin >> 2;
  const int wfloats = Cin * cout_pad;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + wfloats;
  const bool has_sc = sc.valid;          // SE scale vectors always live in LDS (planner)
  for (int wi = wave_id(); wi < chunks * tiles; wi += nw) {
    const int tile = wi / chunks, chunk = wi - tile * chunks;
    const int p = (chunk << 6) + lane;
    if (p >= P) continue;
    const int co0 = tile * CT;
    const int xo = p * x.stride;
    float acc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) acc[t] = 0.f;
    float4 xb[kPF];
#pragma unroll
    for (int j = 0; j < kPF; j++) {
      xb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < nq) {
        if constexpr (XL) xb[j] = ld_lds4(x.l + xo + 4 * j); else xb[j] = ld_glb4(x.g + xo + 4 * j);
        if (has_sc) { float4 sv = ld_lds4(sc.l + 4 * j); xb[j].x = __fmul_rn(xb[j].x, sv.x); xb[j].y = __fmul_rn(xb[j].y, sv.y); xb[j].z = __fmul_rn(xb[j].z, sv.z); xb[j].w = __fmul_rn(xb[j].w, sv.w); }
        if (ad.valid) { float4 av = ld4(ad, p * ad.stride + 4 * j); xb[j].x = __fadd_rn(xb[j].x, av.x); xb[j].y = __fadd_rn(xb[j].y, av.y); xb[j].z = __fadd_rn(xb[j].z, av.z); xb[j].w = __fadd_rn(xb[j].w, av.w); }
      }
    }
    for (int q0 = 0; q0 < nq; q0 += kPF) {
#pragma unroll
      for (int j = 0; j < kPF; j++) {
        const int q = q0 + j;
        if (q < nq) {
          const float4 xv = xb[j];
          if (q + kPF < nq) {
            if constexpr (XL) xb[j] = ld_lds4(x.l + xo + 4 * (q + kPF)); else xb[j] = ld_glb4(x.g + xo + 4 * (q + kPF));
            if (has_sc) { float4 sv = ld_lds4(sc.l + 4 * (q + kPF)); xb[j].x = __fmul_rn(xb[j].x, sv.x); xb[j].y = __fmul_rn(xb[j].y, sv.y); xb[j].z = __fmul_rn(xb[j].z, sv.z); xb[j].w = __fmul_rn(xb[j].w, sv.w); }
            if (ad.valid) { float4 av = ld4(ad, p * ad.stride + 4 * (q + kPF)); xb[j].x = __fadd_rn(xb[j].x, av.x); xb[j].y = __fadd_rn(xb[j].y, av.y); xb[j].z = __fadd_rn(xb[j].z, av.z); xb[j].w = __fadd_rn(xb[j].w, av.w); }
          }
          pw_quad<CT, STAGED>(acc, xv, wl + (q * 4) * cout_pad + co0, as_const(w + (size_t)(q * 4) * cout_pad + co0), cout_pad);
        }
      }
    }
    float bv[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) { if constexpr (STAGED) bv[t] = bl[co0 + t]; else bv[t] = as_const(bias)[co0 + t]; }
    const int yo = p * y.stride + co0, ro = p * res.stride + co0;
    if (CT >= 4 && (Cout & 3) == 0) {
#pragma unroll
      for (int t = 0; t + 3 < CT; t += 4) {
        if (co0 + t < Cout) {
          float4 v = fp_act4(make_float4(acc[t] + bv[t], acc[t + 1] + bv[t + 1], acc[t + 2] + bv[t + 2], acc[t + 3] + bv[t + 3]), act);
          if (res.valid) { float4 r = ld4(res, ro + t); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
          st4(y, yo + t, v);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < CT; t++) {
        if (co0 + t < Cout) {
          float v = fp_act(acc[t] + bv[t], act);
          if (res.valid) v += ld1(res, ro + t);
          st1(y, yo + t, v);
        }
      }
    }
  }
}

// ---- MFMA tile epilogue (quad_transpose: mfma_tile.hpp) -------------------------------------------------------------------------
// out[pix][n0 + 4k .. +3] = act(acc + bias) (+ residual); pix = m0 + 4g + q
__device__ __forceinline__ void mfma_store_tile(const f4acc acc, int m0, int n0, int P, int Cout, const lds_f* bl, int act, bool has_res,
                                                const Ref& res, const Ref& y, int li, int g) {
  const int q = li & 3, c0 = n0 + (li & ~3), pix = m0 + 4 * g + q;
  float4 v = quad_transpose(acc, q);
  if (c0 < Cout && pix < P) {
    const float4 bv = ld_lds4(bl + c0);
    v = fp_act4(make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w), act);
    if (has_res) { const float4 rv = ld4(res, pix * res.stride + c0); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
    st4(y, pix * y.stride + c0, v);
  }
}

// ---- 1x1 convolution on the matrix cores -----------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32: exact f32 FMA chains at the f32 vector rate, but ONE instruction per 1024 MACs — the VALU
// form of these skinny GEMMs (K, N = 16..128) was bound by instruction issue and LDS→FMA latency, not by math.
// Wave = one 16-pixel x 16-channel tile.  Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds x[pixel m0 + (l&15)][k-group l>>4]   → each lane reads ONE float4 (4 consecutive channels of its
//      group) per 16 input channels and feeds .x/.y/.z/.w to four successive MFMAs,
//   B: lane l holds w[k-group l>>4][channel n0 + (l&15)] → one ds_read_b32 per MFMA from the staged weight block,
//   D: lane l holds 4 pixels m0 + 4*(l>>4) + r of channel n0 + (l&15).
// The k order inside the chain is (j, t, group) instead of ascending ci — a different but fixed f32 summation order.

template <bool XL>
__device__ __forceinline__ void pw_mfma(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c);
  const glb_f* w = (const glb_f*)(c.weights + op.w_off);
  const glb_f* bias = (const glb_f*)(c.weights + op.b_off);
  const int P = op.OH * op.OW, Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act;
  const int ws = cout_pad;                             // staged row stride (a 2-way bank conflict on the B reads is
                                                       // invisible next to the 32-cycle MFMA; padding would cost LDS)
  const lds_f* wl = lds_base() + op.w_lds;             // staged by the main loop: weights, then bias
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  (void)w; (void)bias;
  const int mt = (P + 15) >> 4, nt = cout_pad >> 4, nw = kFrameThreads >> 6;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const bool has_sc = sc.valid, has_res = res.valid, has_add = ad.valid;
  const int nj = (Cin + 15) >> 4;
  for (int wi = wave_id(); wi < mt * nt; wi += nw) {
    const int tn = wi / mt, tm = wi - tn * mt;          // consecutive waves share the weight tile, differ in pixels
    const int m0 = tm << 4, n0 = tn << 4;
    const int arow = min(m0 + li, P - 1);               // rows past the end read a valid pixel; their results are dropped
    const int xo = arow * x.stride + 4 * g;
    const lds_f* bp = wl + (4 * g) * ws + n0 + li;
    f4acc acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nj; j++) {
      const int k0 = 16 * j + 4 * g;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      if (k0 < Cin) {                                    // Cin % 4 == 0: a k-group is either fully valid or absent
        if constexpr (XL) a = ld_lds4(x.l + xo + 16 * j); else a = ld_glb4(x.g + xo + 16 * j);
        if (has_sc) { const float4 sv = ld_lds4(sc.l + k0); a.x = __fmul_rn(a.x, sv.x); a.y = __fmul_rn(a.y, sv.y); a.z = __fmul_rn(a.z, sv.z); a.w = __fmul_rn(a.w, sv.w); }
        if (has_add) { const float4 av = ld4(ad, arow * ad.stride + k0); a.x = __fadd_rn(a.x, av.x); a.y = __fadd_rn(a.y, av.y); a.z = __fadd_rn(a.z, av.z); a.w = __fadd_rn(a.w, av.w); }
        const lds_f* br = bp + (16 * j) * ws;
        b0 = br[0]; b1 = br[ws]; b2 = br[2 * ws]; b3 = br[3 * ws];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc, 0, 0, 0);
    }
    mfma_store_tile(acc, m0, n0, P, Cout, bl, act, has_res, res, y, li, g);
  }
}

// LDS input, all of a tile's operands requested up front (NJ = upper bound of ceil(Cin / 16), compile time), two interleaved
// accumulator chains (even / odd k-steps: a dependent f32 MFMA has 40 cycles of latency against 32 of issue).  The measured
// critical path of the small pointwise ops was LDS-read → MFMA → LDS-read … round trips, not arithmetic.
template <int NJ>
__device__ __forceinline__ void pw_mfma_lds(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c);
  const int P = op.OH * op.OW, Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act;
  const int ws = cout_pad;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  const int mt = (P + 15) >> 4, nt = cout_pad >> 4, nw = kFrameThreads >> 6;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const bool has_sc = sc.valid, has_res = res.valid, has_add = ad.valid;
  for (int wi = wave_id(); wi < mt * nt; wi += nw) {
    const int tn = wi / mt, tm = wi - tn * mt;
    const int m0 = tm << 4, n0 = tn << 4;
    const int arow = min(m0 + li, P - 1);
    const int xo = arow * x.stride + 4 * g;
    const lds_f* bp = wl + (4 * g) * ws + n0 + li;
    float4 a[NJ];
    float b[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const bool valid = 16 * j + 4 * g < Cin;
      a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      b[j][0] = b[j][1] = b[j][2] = b[j][3] = 0.f;
      if (valid) {
        a[j] = ld_lds4(x.l + xo + 16 * j);
        const lds_f* br = bp + (16 * j) * ws;
        b[j][0] = br[0]; b[j][1] = br[ws]; b[j][2] = br[2 * ws]; b[j][3] = br[3 * ws];
      }
    }
    if (has_sc || has_add) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int k0 = 16 * j + 4 * g;
        if (k0 < Cin) {
          if (has_sc) { const float4 sv = ld_lds4(sc.l + k0); a[j].x = __fmul_rn(a[j].x, sv.x); a[j].y = __fmul_rn(a[j].y, sv.y); a[j].z = __fmul_rn(a[j].z, sv.z); a[j].w = __fmul_rn(a[j].w, sv.w); }
          if (has_add) { const float4 av = ld4(ad, arow * ad.stride + k0); a[j].x = __fadd_rn(a[j].x, av.x); a[j].y = __fadd_rn(a[j].y, av.y); a[j].z = __fadd_rn(a[j].z, av.z); a[j].w = __fadd_rn(a[j].w, av.w); }
        }
      }
    }
    f4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      if (16 * j < Cin) {                                   // wave-uniform
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b[j][0], acc0, 0, 0, 0);
        if (j + 1 < NJ && 16 * (j + 1) < Cin) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1 < NJ ? j + 1 : j].x, b[j + 1 < NJ ? j + 1 : j][0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b[j][1], acc0, 0, 0, 0);
        if (j + 1 < NJ && 16 * (j + 1) < Cin) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1 < NJ ? j + 1 : j].y, b[j + 1 < NJ ? j + 1 : j][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b[j][2], acc0, 0, 0, 0);
        if (j + 1 < NJ && 16 * (j + 1) < Cin) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1 < NJ ? j + 1 : j].z, b[j + 1 < NJ ? j + 1 : j][2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b[j][3], acc0, 0, 0, 0);
        if (j + 1 < NJ && 16 * (j + 1) < Cin) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1 < NJ ? j + 1 : j].w, b[j + 1 < NJ ? j + 1 : j][3], acc1, 0, 0, 0);
      }
    }
    const f4acc acc = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
    mfma_store_tile(acc, m0, n0, P, Cout, bl, act, has_res, res, y, li, g);
  }
}

// Global-memory input: the op is bound by load round trips, not math (one 16-pixel tile = 4..32 MFMAs against a ~2 us HBM/L2
// latency with only 4 waves per SIMD to hide it).  So a wave (1) reads each A tile ONCE and runs all of its channel tiles from
// registers, and (2) requests the A operands of B consecutive pixel tiles before touching any of them.  FMA order per
// output is unchanged (j ascending), i.e. bit-identical to pw_mfma<false>.
template <int NJ, int B>
__device__ __forceinline__ void pw_mfma_glb(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c);
  const int P = op.OH * op.OW, Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act;
  const int ws = cout_pad;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  const int mt = (P + 15) >> 4, nt = cout_pad >> 4, nw = kFrameThreads >> 6;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const bool has_sc = sc.valid, has_res = res.valid, has_add = ad.valid;
  for (int t0 = wave_id() * B; t0 < mt; t0 += nw * B) {
    float4 a[B][NJ];
#pragma unroll
    for (int i = 0; i < B; i++) {
      const int arow = min(((t0 + i) << 4) + li, P - 1);      // tiles / rows past the end read a valid pixel; results are dropped
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int k0 = 16 * j + 4 * g;
        a[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 < Cin) a[i][j] = ld_glb4(x.g + arow * x.stride + k0);
      }
    }
    if (has_sc || has_add) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int k0 = 16 * j + 4 * g;
        if (k0 < Cin) {
          float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
          if (has_sc) sv = ld_lds4(sc.l + k0);
#pragma unroll
          for (int i = 0; i < B; i++) {
            float4 v = a[i][j];
            if (has_sc) { v.x = __fmul_rn(v.x, sv.x); v.y = __fmul_rn(v.y, sv.y); v.z = __fmul_rn(v.z, sv.z); v.w = __fmul_rn(v.w, sv.w); }
            if (has_add) {
              const int arow = min(((t0 + i) << 4) + li, P - 1);
              const float4 av = ld4(ad, arow * ad.stride + k0);
              v.x = __fadd_rn(v.x, av.x); v.y = __fadd_rn(v.y, av.y); v.z = __fadd_rn(v.z, av.z); v.w = __fadd_rn(v.w, av.w);
            }
            a[i][j] = v;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B; i++) {
      const int m0 = (t0 + i) << 4;
      if (m0 >= P) break;
      for (int tn = 0; tn < nt; tn++) {
        const int n0 = tn << 4;
        const lds_f* bp = wl + (4 * g) * ws + n0 + li;
        f4acc acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int k0 = 16 * j + 4 * g;
          float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
          if (k0 < Cin) { const lds_f* br = bp + (16 * j) * ws; b0 = br[0]; b1 = br[ws]; b2 = br[2 * ws]; b3 = br[3 * ws]; }
          if (16 * j < Cin) {                                 // wave-uniform: skip k-steps beyond Cin entirely
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j].x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j].y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j].z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j].w, b3, acc, 0, 0, 0);
          }
        }
        mfma_store_tile(acc, m0, n0, P, Cout, bl, act, has_res, res, y, li, g);
      }
    }
  }
}

// ---- fused decoder tail: z = act(pw(x*s + a)) → t = z + act2(dw3x3(z)) → out = act3(tconv2x2(t)) ------------------------------
// The three full-resolution ops of the decoder's last level used to round-trip z and t through HBM (the tensors are too
// big for LDS).  Here the frame is processed in bands of R rows: phase A computes the z rows of the band (+1 halo row on
// each side, zero outside the image) on the matrix cores straight into an LDS band; phase B has one lane per pixel form
// t = z + act2(dw(z)) in registers and emit its 2x2 block of transpose-conv outputs.  Arithmetic (order of every FMA
// chain) is exactly that of pw_mfma / dw_body / mo_tconv, so the result is bit-identical to the unfused program.
template <int C, bool XL, bool AL>       // XL / AL: x / the add operand live in LDS (else HBM) — fixed at compile time so that
__device__ __forceinline__ void tail_body(cop_t& op, const FrameCtx& c) {   // the batched loads below really are issued back to back
  constexpr int ZS = C + 4;                                // LDS row stride of a z pixel
  const Ref x = make_ref(op.in0, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c), out = make_ref(op.out, c);
  const int H = op.H, W = op.W, R = op.band_rows, act = op.act, act2 = op.act2, act3 = op.act3, Co = op.C2, OW = op.OW;
  const unsigned mw = op.magic_w;                          // n / W == __umulhi(n, mw)
  const glb_f* gw = (const glb_f*)c.weights;
  // stage the three weight blocks: [pw C x C | pw bias C | dw 9 x C | dw bias C | tconv 4 x Co x C | tconv bias Co]
  // (scalar-load weights in phase B were measured slower than these uniform LDS reads)
  lds_f* wpw = lds_base();
  lds_f* bpw = wpw + C * C;
  lds_f* wdw = bpw + C;
  lds_f* bdw = wdw + 9 * C;
  lds_f* wtc = bdw + C;
  lds_f* btc = wtc + 4 * Co * C;
  for (int i = threadIdx.x; i < C * C; i += kFrameThreads) wpw[i] = gw[op.w_off + i];
  for (int i = threadIdx.x; i < C; i += kFrameThreads) { bpw[i] = gw[op.b_off + i]; bdw[i] = gw[op.b3_off + i]; }
  for (int i = threadIdx.x; i < 9 * C; i += kFrameThreads) wdw[i] = gw[op.w3_off + i];
  for (int i = threadIdx.x; i < 4 * Co * C; i += kFrameThreads) wtc[i] = gw[op.w4_off + i];
  for (int i = threadIdx.x; i < Co; i += kFrameThreads) btc[i] = gw[op.b4_off + i];
  lds_f* zb = lds_base() + op.ws_off;                     // z band: (R+2) rows, stride ZS
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4, nw = kFrameThreads >> 6;
  const bool has_sc = sc.valid, has_add = ad.valid;
  constexpr int NT = C / 16, NJ = C / 16;
  const bool dbg = c.tl && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tA = 0, tB = 0, tS = 0, t0 = 0;
  for (int r0 = 0; r0 < H; r0 += R) {
    const int zrows = R + 2, npix = zrows * W, mt = (npix + 15) >> 4;
    if (dbg) t0 = wall_clock64();
    __syncthreads();                                       // previous band consumed (and weights staged)
    if (dbg) { const unsigned long long t1 = wall_clock64(); tS += t1 - t0; t0 = t1; }
    // ---- phase A: z band rows r0-1 .. r0+R on the matrix cores (A operand = x*s + a straight from HBM / LDS).
    // A wave requests the operands of ALL its tiles of the band (<= TB) before the first MFMA: one memory round trip per band.
    constexpr int TB = 4;
    for (int w0 = wave_id(); w0 < mt * NT; w0 += nw * TB) {
      float4 a[TB][NJ];
#pragma unroll
      for (int t = 0; t < TB; t++) {
        const int wi = w0 + t * nw;
        const int tm = wi % mt, bp = min((tm << 4) + li, npix - 1);
        const int brow = (int)__umulhi((unsigned)bp, mw), bx = bp - brow * W, iy = r0 - 1 + brow;
        const bool rv = wi < mt * NT && iy >= 0 && iy < H;
        const int pix = (rv ? iy : 0) * W + bx;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int k0 = 16 * j + 4 * g;
          a[t][j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rv) {
            float4 v, av = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XL) v = ld_lds4(x.l + pix * x.stride + k0); else v = ld_glb4(x.g + pix * x.stride + k0);
            if (has_add) { if constexpr (AL) av = ld_lds4(ad.l + pix * ad.stride + k0); else av = ld_glb4(ad.g + pix * ad.stride + k0); }
            if (has_sc) { const float4 sv = ld_lds4(sc.l + k0); v.x = __fmul_rn(v.x, sv.x); v.y = __fmul_rn(v.y, sv.y); v.z = __fmul_rn(v.z, sv.z); v.w = __fmul_rn(v.w, sv.w); }
            if (has_add) { v.x = __fadd_rn(v.x, av.x); v.y = __fadd_rn(v.y, av.y); v.z = __fadd_rn(v.z, av.z); v.w = __fadd_rn(v.w, av.w); }
            a[t][j] = v;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TB; t++) {
        const int wi = w0 + t * nw;
        if (wi >= mt * NT) break;
        const int tn = wi / mt, tm = wi - tn * mt, m0 = tm << 4, n0 = tn << 4;
        f4acc acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const lds_f* br = wpw + (16 * j + 4 * g) * C + n0 + li;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].x, br[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].y, br[C], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].z, br[2 * C], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].w, br[3 * C], acc, 0, 0, 0);
        }
        // quad-transposed epilogue: this lane owns pixel m0 + 4g + (li&3), channels 4*(li>>2) .. +3 → one 16-byte LDS store
        const int q = li & 3, c0 = n0 + (li & ~3), p2 = m0 + 4 * g + q;
        float4 v = quad_transpose(acc, q);
        if (p2 < npix) {
          const int row2 = (int)__umulhi((unsigned)p2, mw), iy2 = r0 - 1 + row2;
          const float4 bv = ld_lds4(bpw + c0);
          v = fp_act4(make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w), act);
          if (!(iy2 >= 0 && iy2 < H)) v = make_float4(0.f, 0.f, 0.f, 0.f);          // rows outside the image are zero padding
          st_lds4(zb + p2 * ZS + c0, v);
        }
      }
    }
    __syncthreads();
    if (dbg) { const unsigned long long t1 = wall_clock64(); tA += t1 - t0; t0 = t1; }
    // ---- phase B: one lane per pixel of the R band rows
    const int rows_here = min(R, H - r0);
    for (int p = threadIdx.x; p < rows_here * W; p += kFrameThreads) {
      const int ry = (int)__umulhi((unsigned)p, mw), ix = p - ry * W, iy = r0 + ry;
      float t[C];
#pragma unroll
      for (int q = 0; q < C / 4; q++) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int fy = 0; fy < 3; fy++) {
#pragma unroll 1
          for (int fx = 0; fx < 3; fx++) {
            const int xx = ix - 1 + fx;
            const bool v = xx >= 0 && xx < W;                                 // rows are handled by the zero band rows
            const float4 zv = ld_lds4(zb + ((ry + fy) * W + min(max(xx, 0), W - 1)) * ZS + 4 * q);
            float4 wv = ld_lds4(wdw + (fy * 3 + fx) * C + 4 * q);
            if (!v) wv = make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x = fmaf(zv.x, wv.x, acc.x); acc.y = fmaf(zv.y, wv.y, acc.y); acc.z = fmaf(zv.z, wv.z, acc.z); acc.w = fmaf(zv.w, wv.w, acc.w);
          }
        }
        const float4 b = ld_lds4(bdw + 4 * q), zc = ld_lds4(zb + ((ry + 1) * W + ix) * ZS + 4 * q);
        const float4 d = fp_act4(make_float4(acc.x + b.x, acc.y + b.y, acc.z + b.z, acc.w + b.w), act2);
        t[4 * q] = d.x + zc.x; t[4 * q + 1] = d.y + zc.y; t[4 * q + 2] = d.z + zc.z; t[4 * q + 3] = d.w + zc.w;   // dw epilogue: act, then + residual
      }
      // the 2x2xCo outputs of this input pixel form two contiguous runs of 2*Co floats (one per output row): with Co == 2
      // each run is ONE 16-byte store, and consecutive lanes write consecutive 16-byte chunks
#pragma unroll 1
      for (int fy = 0; fy < 2; fy++) {
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int fx = 0; fx < 2; fx++) {
          for (int oc = 0; oc < Co; oc++) {
            const lds_f* wr = wtc + ((fy * 2 + fx) * Co + oc) * C;
            float acc = btc[oc];
#pragma unroll
            for (int k = 0; k < C; k += 4) {
              const float4 wv = ld_lds4(wr + k);
              acc = fmaf(t[k], wv.x, acc); acc = fmaf(t[k + 1], wv.y, acc); acc = fmaf(t[k + 2], wv.z, acc); acc = fmaf(t[k + 3], wv.w, acc);
            }
            const float v = fp_act(acc, act3);
            if (Co == 2) { if (fx == 0) { if (oc == 0) o4[0] = v; else o4[1] = v; } else { if (oc == 0) o4[2] = v; else o4[3] = v; } }
            else st1(out, ((2 * iy + fy) * OW + 2 * ix + fx) * out.stride + oc, v);
          }
        }
        if (Co == 2) st4(out, ((2 * iy + fy) * OW + 2 * ix) * out.stride, make_float4(o4[0], o4[1], o4[2], o4[3]));
      }
    }
    if (dbg) { const unsigned long long t1 = wall_clock64(); tB += t1 - t0; t0 = t1; }
  }
  if (dbg) { c.tl[256] = tS; c.tl[257] = tA; c.tl[258] = tB; }
}
__device__ __forceinline__ void mo_tail(cop_t& op, const FrameCtx& c) {
  const bool xl = op.in0.space == kLocLds, al = op.in2.space == kLocLds;   // the planner only forms this micro-op for 16-channel tails
  if (xl) { if (al) tail_body<16, true, true>(op, c); else tail_body<16, true, false>(op, c); }
  else { if (al) tail_body<16, false, true>(op, c); else tail_body<16, false, false>(op, c); }
}

__device__ __forceinline__ void mo_pw(cop_t& op, const FrameCtx& c) {
  const bool xl = op.in0.space == kLocLds;
  if (op.mfma) {
    if (xl) {
      const int njl = (op.Cin + 15) >> 4;
      if (njl <= 2) pw_mfma_lds<2>(op, c);
      else if (njl <= 6) pw_mfma_lds<6>(op, c);
      else if (njl <= 8) pw_mfma_lds<8>(op, c);
      else pw_mfma<true>(op, c);
      return;
    }
    const int nj = (op.Cin + 15) >> 4;
    if (nj == 1) pw_mfma_glb<1, 4>(op, c);
    else if (nj <= 2) pw_mfma_glb<2, 2>(op, c);
    else pw_mfma<false>(op, c);
    return;
  }
  if (xl) pw_body<16, false, true>(op, c); else pw_body<16, false, false>(op, c);   // weight block too large to stage: SGPR-fed VALU form
}

// ---- 1x1 convolution on <= 4 pixels (SE / gate FCs): lane = (output channel, K-slice) ----------------------------
// [ci][co] weight rows are read coalesced across lanes, all of a lane's loads in flight at once; the KS partial
// sums per output meet in the LDS scratch.  Inputs/outputs of these steps are [1,1,C] vectors → always LDS.
__device__ __forceinline__ void mo_gemv(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c), sc = make_ref(op.scale, c), ad = make_ref(op.in2, c);
  const glb_f* w = (const glb_f*)(c.weights + op.w_off);       // [ci][cout_pad]
  const glb_f* bias = (const glb_f*)(c.weights + op.b_off);
  const int P = op.OH * op.OW, Cin = op.Cin, Cout = op.Cout, cout_pad = op.cout_pad;
  int KS = 1;
  while (KS * 2 * P * Cout <= kFrameThreads && KS * 2 <= 16 && (Cin % (KS * 2)) == 0) KS *= 2;
  const int items = P * Cout * KS;            // planner guarantees P*Cout*16 <= kLdsScratchFloats
  lds_f* scratch = lds_base();
  const int klen = Cin / KS;
  for (int i = threadIdx.x; i < items; i += kFrameThreads) {
    const int co = i % Cout, r = i / Cout, p = r % P, ks = r / P;
    const int xo = p * x.stride + ks * klen;
    const glb_f* wr = w + (size_t)(ks * klen) * cout_pad + co;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < klen; k++) {
      float xv = ld1(x, xo + k);
      if (sc.valid) xv *= ld1(sc, ks * klen + k);
      acc = fmaf(xv, wr[(size_t)k * cout_pad], acc);
    }
    scratch[i] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P * Cout; i += kFrameThreads) {
    const int co = i % Cout, p = i / Cout;
    float acc = 0.f;
    for (int ks = 0; ks < KS; ks++) acc += scratch[(ks * P + p) * Cout + co];
    float v = fp_act(acc + bias[co], op.act);
    if (res.valid) v += ld1(res, p * res.stride + co);
    st1(y, p * y.stride + co, v);
  }
}

// ---- dense k x k convolution (network stems): wave = (chunk, 16-channel tile), lane = output pixel ---------------
__device__ __forceinline__ void mo_conv(cop_t& op, const FrameCtx& c) {
  constexpr int CT = 16;
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int P = op.OH * op.OW, chunks = (P + 63) >> 6, tiles = op.cout_pad / CT;
  const int lane = threadIdx.x & 63, nw = kFrameThreads >> 6;
  const bool staged = op.stage_floats > 0;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
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
      const int iy = oy * op.sh - op.pt + fy * op.dh;
      const bool vy = iy >= 0 && iy < op.H;
      const int cy = min(max(iy, 0), op.H - 1);
      for (int fx = 0; fx < op.kw; fx++) {
        const int ix = ox * op.sw - op.pl + fx * op.dw;
        const float m = (vy && ix >= 0 && ix < op.W) ? 1.f : 0.f;    // zero-padding as a multiplier: branch-free taps
        const int cx = min(max(ix, 0), op.W - 1);
        const int xo = (cy * op.W + cx) * x.stride;
        const int wo = (fy * op.kw + fx) * op.Cin * op.cout_pad + co0;
        for (int ci = 0; ci < op.Cin; ci++) {
          const float xv = ld1(x, xo + ci) * m;
          if (staged) {
#pragma unroll
            for (int t = 0; t < CT; t += 4) {
              float4 wv = ld_lds4(wl + wo + ci * op.cout_pad + t);
              acc[t] = fmaf(xv, wv.x, acc[t]); acc[t + 1] = fmaf(xv, wv.y, acc[t + 1]);
              acc[t + 2] = fmaf(xv, wv.z, acc[t + 2]); acc[t + 3] = fmaf(xv, wv.w, acc[t + 3]);
            }
          } else {
            cfloat_t* w0 = as_const(w + wo + (size_t)ci * op.cout_pad);
#pragma unroll
            for (int t = 0; t < CT; t++) acc[t] = fmaf(xv, w0[t], acc[t]);
          }
        }
      }
    }
    const int yo = p * y.stride + co0;
#pragma unroll
    for (int t = 0; t < CT; t++) {
      if (co0 + t < op.Cout) {
        float v = fp_act(acc[t] + (staged ? bl[co0 + t] : as_const(bias)[co0 + t]), op.act);
        if (res.valid) v += ld1(res, p * res.stride + co0 + t);
        st1(y, yo + t, v);
      }
    }
  }
}

// ---- dense k x k convolution on the matrix cores (network stems) ---------------------------------------------------------
// The input (HBM, dense rows) is streamed through an LDS band of (band_rows-1)*stride + k input rows with one coalesced
// cooperative copy per band; a wave then owns a tile of 16 consecutive output pixels of one row x 16 output channels and
// runs ceil(K/4) MFMA steps over the im2col axis K = kh*kw*Cin (27 for the RGB stems): operand A is gathered from the
// band through a per-k offset table, operand B comes from the staged [K][cout_pad] weight block.  Out-of-image taps
// (SAME padding) and the K tail contribute an exact 0.
__device__ __forceinline__ void mo_conv_mfma(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c);
  const int H = op.H, W = op.W, Cin = op.Cin, OH = op.OH, OW = op.OW, Cout = op.Cout, cout_pad = op.cout_pad, act = op.act;
  const int kh = op.kh, kw = op.kw, sh = op.sh, sw = op.sw, pt = op.pt, pl = op.pl;
  const int K = kh * kw * Cin, nsteps = (K + 3) >> 2;
  const int rowf = W * Cin;                                      // floats per input row
  const int band_in = (op.band_rows - 1) * sh + kh;              // input rows per band
  lds_f* band = lds_base() + op.ws_off;
  lds_f* ktab = band + band_in * rowf;                           // per-k table: offset inside the band, fy, fx  (3 ints per k)
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  for (int k = threadIdx.x; k < nsteps * 4; k += kFrameThreads) {
    int fy = 0, fx = 0, ci = 0, valid = k < K;
    if (valid) { fy = k / (kw * Cin); const int r = k - fy * kw * Cin; fx = r / Cin; ci = r - fx * Cin; }
    ((__attribute__((address_space(3))) int*)ktab)[3 * k] = valid ? (fy * W + fx) * Cin + ci : 0;
    ((__attribute__((address_space(3))) int*)ktab)[3 * k + 1] = valid ? fy : -100000;     // invalid k → never inside the image
    ((__attribute__((address_space(3))) int*)ktab)[3 * k + 2] = fx;
  }
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4, nw = kFrameThreads >> 6;
  const int mtx = (OW + 15) >> 4, nt = cout_pad >> 4;
  // band rows are double-buffered through registers: the next band's HBM loads are in flight while this band's tiles run
  constexpr int kBandRegs = 2;                                   // planner keeps a band <= 8192 floats = 2 float4 per lane
  f4v breg[kBandRegs];
  auto band_load = [&](int oy0) {
    const int iy0 = oy0 * sh - pt;
#pragma unroll
    for (int q = 0; q < kBandRegs; q++) {
      const int i = (q * kFrameThreads + (int)threadIdx.x) * 4;
      f4v v = {0.f, 0.f, 0.f, 0.f};
      if (i < band_in * rowf) {
        if ((rowf & 3) == 0) {
          const int iy = iy0 + i / rowf;
          if (iy >= 0 && iy < H) v = *(const glb_v4*)(x.g + (long)iy0 * rowf + i);
        } else {
          for (int e = 0; e < 4; e++) {
            const int ii = i + e, yy = iy0 + ii / rowf;
            if (ii < band_in * rowf && yy >= 0 && yy < H) v[e] = x.g[(long)iy0 * rowf + ii];
          }
        }
      }
      breg[q] = v;
    }
  };
  band_load(0);
  // the (<= 8) im2col entries this lane feeds — k = 4*step + g — live in registers for the whole op
  constexpr int kKReg = 8;
  int koff[kKReg], kfyx[kKReg];                                  // fy and fx packed as (fy << 8) | fx; invalid k has fy = -100000
  __syncthreads();
#pragma unroll
  for (int s4 = 0; s4 < kKReg; s4++) {
    const int k = min(4 * s4 + g, nsteps * 4 - 1);
    koff[s4] = ((const __attribute__((address_space(3))) int*)ktab)[3 * k];
    kfyx[s4] = ((const __attribute__((address_space(3))) int*)ktab)[3 * k + 1] * 256 + ((const __attribute__((address_space(3))) int*)ktab)[3 * k + 2];
  }
  for (int oy0 = 0; oy0 < OH; oy0 += op.band_rows) {
    const int rows_out = min(op.band_rows, OH - oy0);
    __syncthreads();                                             // previous band fully consumed (and ktab written)
#pragma unroll
    for (int q = 0; q < kBandRegs; q++) {
      const int i = (q * kFrameThreads + (int)threadIdx.x) * 4;
      if (i < band_in * rowf) *(lds_v4*)(band + i) = breg[q];
    }
    __syncthreads();
    if (oy0 + op.band_rows < OH) band_load(oy0 + op.band_rows);
    const int tiles = rows_out * mtx * nt;
    for (int wi = wave_id(); wi < tiles; wi += nw) {
      const int tn = wi / (rows_out * mtx), rem = wi - tn * rows_out * mtx;
      const int ry = rem / mtx, tm = rem - ry * mtx;
      const int oy = oy0 + ry, ox = min(tm * 16 + li, OW - 1), n0 = tn << 4;
      const int base = (ry * sh * W + (ox * sw - pl)) * Cin;      // band offset of tap (0,0); may point left of the row start
      const int iyb = oy * sh - pt, ixb = ox * sw - pl;
      f4acc acc = {0.f, 0.f, 0.f, 0.f};
      if (nsteps <= kKReg) {
        // all operand loads of the tile are independent: issue them together, then run the MFMA chain
        float av[kKReg], bv4[kKReg];
#pragma unroll
        for (int s4 = 0; s4 < kKReg; s4++) {
          av[s4] = 0.f; bv4[s4] = 0.f;
          if (s4 < nsteps) {
            const int iy = iyb + (kfyx[s4] >> 8), ix = ixb + (kfyx[s4] & 255);
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            av[s4] = ok ? band[base + koff[s4]] : 0.f;
            bv4[s4] = wl[min(4 * s4 + g, K - 1) * cout_pad + n0 + li];
          }
        }
#pragma unroll
        for (int s4 = 0; s4 < kKReg; s4++)
          if (s4 < nsteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv4[s4], acc, 0, 0, 0);
      } else
      for (int s4 = 0; s4 < nsteps; s4++) {
        const int k = 4 * s4 + g;
        const int koff = ((const __attribute__((address_space(3))) int*)ktab)[3 * k];
        const int fy = ((const __attribute__((address_space(3))) int*)ktab)[3 * k + 1];
        const int fx = ((const __attribute__((address_space(3))) int*)ktab)[3 * k + 2];
        const int iy = iyb + fy, ix = ixb + fx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float a = ok ? band[base + koff] : 0.f;
        const float b = wl[min(k, K - 1) * cout_pad + n0 + li];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
      const int co = n0 + li;
      if (co < Cout) {
        const float bv = bl[co];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int oxr = tm * 16 + 4 * g + r;
          if (oxr < OW) st1(y, (oy * OW + oxr) * y.stride + co, fp_act(acc[r] + bv, act));
        }
      }
    }
  }
}

// ---- depthwise: lane = (output pixel, channel quad) --------------------------------------------------------------------
// Weights + bias staged in LDS when they fit; taps are branch-free (clamped address, zeroed weight outside the image:
// 0 * finite == 0, summation order stays fy, fx ascending).
template <int K, bool XL>
__device__ __forceinline__ void dw_body(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int C = op.Cin, C4 = C >> 2;
  const int kh = K ? K : op.kh, kw = K ? K : op.kw, kk = kh * kw;
  const bool staged = op.stage_floats > 0;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  (void)kk;
  const glb_f* wg = (const glb_f*)w;
  const glb_f* bg = (const glb_f*)bias;
  const int total = op.OH * op.OW * C4;
  const int H = op.H, W = op.W, OW = op.OW, sh = op.sh, sw = op.sw, dh = op.dh, dw = op.dw, pt = op.pt, pl = op.pl, act = op.act;
  // lane = (channel quad, pixel row): the quad is fixed per lane and (oy, ox) advance incrementally — no per-item division
  const int rows = kFrameThreads / C4, cq = threadIdx.x % C4, r0 = threadIdx.x / C4, ch = cq * 4;
  const int step_y = rows / OW, step_x = rows - step_y * OW, P = total / C4;
  int oy = r0 / OW, ox = r0 - oy * OW;
  if (c.tl && blockIdx.x == 0 && threadIdx.x == 0) c.tl[265] += 1;      // count of depthwise ops seen by the timeline
  if (r0 < rows)
  for (int p = r0; p < P; p += rows, ox += step_x, oy += step_y) {
    if (ox >= OW) { ox -= OW; oy++; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // column addresses / validity are shared by all rows of the window
    int cxo[K ? K : 8];
    bool vx[K ? K : 8];
#pragma unroll
    for (int fx = 0; fx < kw; fx++) {
      const int ix = ox * sw - pl + fx * dw;
      vx[fx] = ix >= 0 && ix < W;
      cxo[fx] = min(max(ix, 0), W - 1) * x.stride + ch;
    }
#pragma unroll 1
    for (int fy = 0; fy < kh; fy++) {
      const int iy = oy * sh - pt + fy * dh;
      const bool vy = iy >= 0 && iy < H;
      const int rowo = min(max(iy, 0), H - 1) * W * x.stride;
      const int wrow = fy * kw * C + ch;
#pragma unroll
      for (int fx = 0; fx < kw; fx++) {
        float4 xv;
        if constexpr (XL) xv = ld_lds4(x.l + rowo + cxo[fx]); else xv = ld_glb4(x.g + rowo + cxo[fx]);
        float4 wv = ldw4(staged, wl, wg, wrow + fx * C);
        if (!(vy && vx[fx])) wv = make_float4(0.f, 0.f, 0.f, 0.f);
        acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
        acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
      }
    }
    const float4 b = ldw4(staged, bl, bg, ch);
    float4 v = fp_act4(make_float4(acc.x + b.x, acc.y + b.y, acc.z + b.z, acc.w + b.w), act);
    if (res.valid) { float4 r = ld4(res, p * res.stride + ch); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    st4(y, p * y.stride + ch, v);
  }
}
// ---- depthwise, register-strip form: lane = (channel quad, strip of TX consecutive output columns, output row) ----------------
// The per-pixel form above reads K*K inputs AND K*K weights from LDS for every output quad: at 6x10x128 (5x5) that is 1.6 MB of
// ds_read_b128 per op — the op was LDS-bandwidth bound (12.8k of its 13k cycles).  Here a lane loads, per filter row, the
// (TX-1)*S + K input quads its strip touches and the K weight quads ONCE and forms TX outputs from registers (packed f32 FMAs):
// (NIN + K) / TX loads per output row-tap instead of 2K.  Channel quad is the fastest lane index: ds_read_b128 / global loads of
// neighbouring lanes are contiguous.  Per output the FMA order is fy, fx ascending from 0, bias last — bit-identical to dw_body.
template <int K, int S, int TX, int V, bool XL>     // V = channels per lane (4: b128 accesses; 2: half the registers — the 5x5 forms spill at 4)
__device__ __forceinline__ void dw_strip(cop_t& op, const FrameCtx& c) {
  constexpr int NIN = (TX - 1) * S + K;
  typedef float vec_t __attribute__((ext_vector_type(V)));
  typedef __attribute__((address_space(3))) vec_t lds_vec;
  typedef __attribute__((address_space(1))) vec_t glb_vec;
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c), res = make_ref(op.res, c);
  const int C = op.Cin, CV = C / V;
  const bool staged = op.stage_floats > 0;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  const glb_f* wg = (const glb_f*)(c.weights + op.w_off);
  const glb_f* bg = (const glb_f*)(c.weights + op.b_off);
  const int H = op.H, W = op.W, OH = op.OH, OW = op.OW, pt = op.pt, pl = op.pl, act = op.act;
  const int nstrips = (OW + TX - 1) / TX, total = CV * nstrips * OH;
  if (c.tl && blockIdx.x == 0 && threadIdx.x == 0) c.tl[265] += 1;
  for (int item = threadIdx.x; item < total; item += kFrameThreads) {
    const int t = item / CV, cq = item - t * CV, oy = t / nstrips, sx = t - oy * nstrips;
    const int ch = cq * V, ox0 = sx * TX, ix0 = ox0 * S - pl;
    vec_t acc[TX];
#pragma unroll
    for (int k = 0; k < TX; k++) acc[k] = (vec_t)(0.f);
#pragma unroll 1
    for (int fy = 0; fy < K; fy++) {
      const int iy = oy * S - pt + fy;
      const bool vy = iy >= 0 && iy < H;
      const int rowo = min(max(iy, 0), H - 1) * W * x.stride + ch;
      vec_t xin[NIN], wv[K];
#pragma unroll
      for (int j = 0; j < NIN; j++) {
        const int ix = ix0 + j, off = rowo + min(max(ix, 0), W - 1) * x.stride;
        if constexpr (XL) xin[j] = *(const lds_vec*)(x.l + off); else xin[j] = *(const glb_vec*)(x.g + off);
      }
#pragma unroll
      for (int fx = 0; fx < K; fx++) {
        if (staged) wv[fx] = *(const lds_vec*)(wl + (fy * K + fx) * C + ch); else wv[fx] = *(const glb_vec*)(wg + (fy * K + fx) * C + ch);
        if (!vy) wv[fx] = (vec_t)(0.f);                       // row outside the image: its taps add 0 (as in dw_body)
      }
#pragma unroll
      for (int j = 0; j < NIN; j++) { const int ix = ix0 + j; if (ix < 0 || ix >= W) xin[j] = (vec_t)(0.f); }
#pragma unroll
      for (int k = 0; k < TX; k++) {
#pragma unroll
        for (int fx = 0; fx < K; fx++) acc[k] = __builtin_elementwise_fma(xin[k * S + fx], wv[fx], acc[k]);
      }
    }
    vec_t bq;
    if (staged) bq = *(const lds_vec*)(bl + ch); else bq = *(const glb_vec*)(bg + ch);
#pragma unroll
    for (int k = 0; k < TX; k++) {
      const int ox = ox0 + k;
      if (ox < OW) {
        const int pix = oy * OW + ox;
        vec_t v = acc[k] + bq;
#pragma unroll
        for (int e = 0; e < V; e++) v[e] = fp_act(v[e], act);
        if (res.valid) { vec_t r; if (res.lds) r = *(const lds_vec*)(res.l + pix * res.stride + ch); else r = *(const glb_vec*)(res.g + pix * res.stride + ch); v += r; }
        if (y.lds) *(lds_vec*)(y.l + pix * y.stride + ch) = v; else *(glb_vec*)(y.g + pix * y.stride + ch) = v;
      }
    }
  }
}
template <int K, int S>
__device__ __forceinline__ void dw_strip_pick(cop_t& op, const FrameCtx& c) {
  const bool xl = op.in0.space == kLocLds;
  constexpr int V = K == 5 ? 2 : 4;
  constexpr int TXA = S == 1 ? 5 : 4;           // input quads per filter row: 9 (5x5) / 7 (3x3) at stride 1, 11 / 9 at stride 2
  if (S == 1 && op.OW % 5 != 0 && op.OW % 4 == 0) { if (xl) dw_strip<K, S, 4, V, true>(op, c); else dw_strip<K, S, 4, V, false>(op, c); return; }
  if (xl) dw_strip<K, S, TXA, V, true>(op, c); else dw_strip<K, S, TXA, V, false>(op, c);
}
__device__ __forceinline__ void mo_dw(cop_t& op, const FrameCtx& c) {
  const bool xl = op.in0.space == kLocLds;
  if (op.strip && op.dh == 1 && op.dw == 1 && op.kh == op.kw && op.sh == op.sw && (op.sh == 1 || op.sh == 2)) {
    if (op.kh == 3) { if (op.sh == 1) dw_strip_pick<3, 1>(op, c); else dw_strip_pick<3, 2>(op, c); return; }
    if (op.kh == 5) { if (op.sh == 1) dw_strip_pick<5, 1>(op, c); else dw_strip_pick<5, 2>(op, c); return; }
  }
  if (op.kh == 3) { if (xl) dw_body<3, true>(op, c); else dw_body<3, false>(op, c); }     // planner admits 3x3 and 5x5 only
  else { if (xl) dw_body<5, true>(op, c); else dw_body<5, false>(op, c); }
}

// ---- global average pool of one input into out[coff .. coff+C) -------------------------------------------------------------
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// Wave-owned channel quads: wave w pools quads w, w + 16, ... ENTIRELY — lanes = rows (pixels, or per-tile partial sums), one
// DPP butterfly inside each 16-lane row, the four row totals meet through v_readlane.  No LDS scratch and NO workgroup barrier
// inside the pooling (the previous form met in the scratch twice per 16 quads: 4-6 barriers of ~1.5k cycles per SE op, 55 % of it).
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;   // quad_perm [1,0,3,2] / [2,3,0,1], row_ror:4, row_ror:8
// float4 per lane → wave total of component (lane & 3) in every lane: a 4x4 reduce-scatter inside each quad (3 DPP exchanges
// for all four components: lane q of a quad ends with component q), two rotations by whole quads inside the row of 16, two
// cross-row exchanges — 7 exchanges instead of 4 x 8.
__device__ __forceinline__ float wave_total_scatter(float4 a, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  const float klo = (b0 ? a.y : a.x) + dpp<kDppQuadXor1>(b0 ? a.x : a.y);
  const float khi = (b0 ? a.w : a.z) + dpp<kDppQuadXor1>(b0 ? a.z : a.w);
  float v = (b1 ? khi : klo) + dpp<kDppQuadXor2>(b1 ? klo : khi);        // quad total of component (lane & 3)
  v += dpp<kDppRor4>(v);                                                 // lanes i, i-4, i-8, i-12 of a row hold the same component
  v += dpp<kDppRor8>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
template <bool XL>
__device__ __forceinline__ void gap_wave(const lds_f* xl, const glb_f* xg, int rows, int stride, int C, float denom, const Ref& out, int coff, bool accumulate) {
  const int C4 = C >> 2, lane = threadIdx.x & 63;
  for (int cq = wave_id(); cq < C4; cq += kFrameThreads >> 6) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    auto ldx = [&](int r) { if constexpr (XL) return ld_lds4(xl + r * stride + cq * 4); else return ld_glb4(xg + r * stride + cq * 4); };
    int r = lane;
    for (; r + 64 < rows; r += 128) { const float4 v0 = ldx(r), v1 = ldx(r + 64); a0 = add4(a0, v0); a1 = add4(a1, v1); }
    if (r < rows) a0 = add4(a0, ldx(r));
    a0 = add4(a0, a1);
    float t = wave_total_scatter(a0, lane);
    if (lane < 4) {                                                // lane e holds the total of channel 4 cq + e
      t /= denom;
      const int e = lane;
      if (accumulate) t += ld1(out, coff + cq * 4 + e);            // GAP(a + b) as GAP(a) + GAP(b): the same lane wrote the first part
      st1(out, coff + cq * 4 + e, t);
    }
  }
}

__device__ __forceinline__ void gap_one(const Ref& x, int HW, int C, const Ref& out, int coff, bool accumulate = false) {
  if (x.lds) gap_wave<true>(x.l, x.g, HW, x.stride, C, (float)HW, out, coff, accumulate);
  else gap_wave<false>(x.l, x.g, HW, x.stride, C, (float)HW, out, coff, accumulate);
}

// one pooled part of a (possibly concatenated / summed) global average pool: a tensor, or per-tile partial sums written by
// a segment kernel (segments.hpp) — [n][C] floats in the frame's arena slice, mean = sum / hw.  No barrier inside: the caller
// publishes the means with one barrier after the last part.
__device__ __forceinline__ void gap_part(cop_t& op, int k, const FrameCtx& c, const Ref& out, int coff, bool accumulate) {
  const Ref x = make_ref(op.cat[k], c);
  const int C = op.cat_c[k], np = op.cat_parts[k];
  if (np == 0) { gap_one(x, op.cat_hw[k], C, out, coff, accumulate); return; }
  gap_wave<false>(x.l, x.g, np, C, C, (float)op.cat_hw[k], out, coff, accumulate);
}

__device__ __forceinline__ void mo_gap(cop_t& op, const FrameCtx& c) {
  const Ref out = make_ref(op.out, c);
  const int HW = op.H * op.W;
  if (op.n_cat == 0) { gap_one(make_ref(op.in0, c), HW, op.Cin, out, 0); return; }
  int coff = 0;
  for (int k = 0; k < op.n_cat; k++) {
    gap_part(op, k, c, out, coff, op.gap_sum && k > 0);
    if (!op.gap_sum) coff += op.cat_c[k];
  }
}

// ---- fused squeeze-excite / decoder-gate chain: GAP → FC(+act) [→ FC(+act)] ---------------------------------------------
// One micro-op instead of three: the means stay in LDS, each FC output is a dot product split over L consecutive lanes
// (float4 reads of the [co][ci] weight rows, shuffle reduction) — no scratch round trip, two barriers in total.
// FC layer over L-lane groups.  `fc_preload` fetches a lane's slice of its (first) output row into registers so that
// the global latency overlaps whatever runs before the matching `fc_lanes` call (the pooling phase / the previous FC).
constexpr int kFcPre = 4;   // float4s per lane held in registers (covers Cin <= 128 with 8 lanes per output)
struct FcPre { float4 w[kFcPre]; float b; };
__device__ __forceinline__ int fc_group(int Cin) { int L = 8; while (L > 1 && (Cin % (4 * L)) != 0) L >>= 1; return L; }
__device__ __forceinline__ FcPre fc_preload(int Cin, const glb_f* w2, const glb_f* bias, int Cout) {
  FcPre p;
  const int L = fc_group(Cin), lg = L == 8 ? 3 : (L == 4 ? 2 : (L == 2 ? 1 : 0)), klen = Cin >> lg, sub = threadIdx.x & (L - 1), co = threadIdx.x >> lg;
#pragma unroll
  for (int q = 0; q < kFcPre; q++) p.w[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  p.b = 0.f;
  if (co < Cout) {
    const glb_f* wr = w2 + (size_t)co * Cin + sub * klen;
#pragma unroll
    for (int q = 0; q < kFcPre; q++) if (4 * q < klen) p.w[q] = ld_glb4(wr + 4 * q);
    p.b = bias[co];
  }
  return p;
}
__device__ __forceinline__ void fc_lanes(const Ref& x, int Cin, const glb_f* w2, const glb_f* bias, int Cout, int act, const Ref& y, const FcPre& pre) {
  const int L = fc_group(Cin), lg = L == 8 ? 3 : (L == 4 ? 2 : (L == 2 ? 1 : 0)), klen = Cin >> lg, sub = threadIdx.x & (L - 1), per = kFrameThreads >> lg;
  const bool pre_ok = klen <= 4 * kFcPre;
  for (int co = threadIdx.x >> lg, it = 0; co < Cout; co += per, it++) {
    const glb_f* wr = w2 + (size_t)co * Cin + sub * klen;
    float acc = 0.f;
    if (it == 0 && pre_ok) {
#pragma unroll
      for (int q = 0; q < kFcPre; q++) {
        if (4 * q < klen) {
          const float4 xv = ld4(x, sub * klen + 4 * q), wv = pre.w[q];
          acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
      }
    } else {
      for (int k = 0; k < klen; k += 4) {
        const float4 xv = ld4(x, sub * klen + k), wv = ld_glb4(wr + k);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
      }
    }
    if (L >= 2) acc += dpp<kDppQuadXor1>(acc);      // the L (<= 8, power of two) lanes of an output are consecutive and aligned
    if (L >= 4) acc += dpp<kDppQuadXor2>(acc);
    if (L >= 8) acc += dpp<0x141>(acc);             // row_half_mirror: lane i + lane 7-i = both quad totals of the aligned group of 8
    if (sub == 0) st1(y, co, fp_act(acc + ((it == 0 && pre_ok) ? pre.b : bias[co]), act));
  }
}
// the same layer with its [bias | [co][ci] weights] block staged in LDS by the main loop's DMA (MicroOp::fc_stage)
__device__ __forceinline__ void fc_lanes_lds(const Ref& x, int Cin, const lds_f* w2, const lds_f* bias, int Cout, int act, const Ref& y) {
  const int L = fc_group(Cin), lg = L == 8 ? 3 : (L == 4 ? 2 : (L == 2 ? 1 : 0)), klen = Cin >> lg, sub = threadIdx.x & (L - 1), per = kFrameThreads >> lg;
  for (int co = threadIdx.x >> lg; co < Cout; co += per) {
    const lds_f* wr = w2 + co * Cin + sub * klen;
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < klen; k += 4) {
      const float4 xv = ld4(x, sub * klen + k), wv = ld_lds4(wr + k);
      a0 = fmaf(xv.x, wv.x, a0); a1 = fmaf(xv.y, wv.y, a1); a0 = fmaf(xv.z, wv.z, a0); a1 = fmaf(xv.w, wv.w, a1);
    }
    float acc = a0 + a1;
    if (L >= 2) acc += dpp<kDppQuadXor1>(acc);
    if (L >= 4) acc += dpp<kDppQuadXor2>(acc);
    if (L >= 8) acc += dpp<0x141>(acc);
    if (sub == 0) st1(y, co, fp_act(acc + bias[co], act));
  }
}
__device__ __forceinline__ void mo_se(cop_t& op, const FrameCtx& c) {
  const bool dbg = c.tl && blockIdx.x == 0 && threadIdx.x == 0;     // timeline runs: cycles per phase, summed over all SE ops, in tl[260..264]
  unsigned long long t0 = dbg ? __builtin_readcyclecounter() : 0;
  auto stamp = [&](int slot) { if (dbg) { const unsigned long long t1 = __builtin_readcyclecounter(); c.tl[slot] += t1 - t0; t0 = t1; } };
  const Ref mean = make_ref(op.in1, c), hid = make_ref(op.in2, c), out = make_ref(op.out, c);
  const int HW = op.H * op.W;
  const glb_f* wts = (const glb_f*)c.weights;
  // both FCs' weight slices are requested up front: their HBM/L2 latency hides behind the pooling reductions
  const bool st1 = op.fc_stage[0] > 0, st2 = op.fc_stage[1] > 0;     // layer's [bias | weights] already in LDS (DMA issued an op ago)
  const lds_f* l1 = lds_base() + op.fc_lds[0];
  const lds_f* l2 = lds_base() + op.fc_lds[1];
  FcPre p1, p2;
  if (!st1) p1 = fc_preload(op.Cin, wts + op.w2_off, wts + op.b_off, op.C1);
  if (op.n_fc == 2 && !st2) p2 = fc_preload(op.C1, wts + op.w3_off, wts + op.b3_off, op.C2);
  stamp(260);
  if (op.n_cat == 0) gap_one(make_ref(op.in0, c), HW, op.Cin, mean, 0);
  else {
    int coff = 0;
    for (int k = 0; k < op.n_cat; k++) {
      gap_part(op, k, c, mean, coff, op.gap_sum && k > 0);
      if (!op.gap_sum) coff += op.cat_c[k];
    }
  }
  __syncthreads();                 // means complete (the pooling itself is barrier-free)
  stamp(261);
  const Ref& y1 = op.n_fc == 1 ? out : hid;
  if (st1) fc_lanes_lds(mean, op.Cin, l1 + (int)(op.w2_off - op.b_off), l1, op.C1, op.act, y1);
  else fc_lanes(mean, op.Cin, wts + op.w2_off, wts + op.b_off, op.C1, op.act, y1, p1);
  stamp(262);
  if (op.n_fc == 1) return;
  __syncthreads();
  stamp(263);
  if (st2) fc_lanes_lds(hid, op.C1, l2 + (int)(op.w3_off - op.b3_off), l2, op.C2, op.act2, out);
  else fc_lanes(hid, op.C1, wts + op.w3_off, wts + op.b3_off, op.C2, op.act2, out, p2);
  stamp(264);
}

// ---- elementwise -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float elt1(float a, float b, float cc, int e) {
  switch (e) {
    case kEltAdd: return a + b;
    case kEltMul: return a * b;
    case kEltMulAdd: return __fadd_rn(__fmul_rn(a, b), cc);
    default: return a;
  }
}
__device__ __forceinline__ void mo_elt(cop_t& op, const FrameCtx& c) {
  const Ref a = make_ref(op.in0, c), b = make_ref(op.in1, c), d = make_ref(op.in2, c), y = make_ref(op.out, c);
  const int C4 = op.Cin >> 2, total = op.H * op.W * C4, elt = op.elt, act = op.act, bc = op.bcast1;
  const int rows = kFrameThreads / C4, ch = (threadIdx.x % C4) * 4, r0 = threadIdx.x / C4, P = total / C4;
  if (r0 < rows)
  for (int p = r0; p < P; p += rows) {
    const float4 av = ld4(a, p * a.stride + ch);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), dv = bv;
    if (elt != kEltUnary) bv = ld4(b, bc ? ch : p * b.stride + ch);
    if (elt == kEltMulAdd) dv = ld4(d, p * d.stride + ch);
    float4 v = make_float4(elt1(av.x, bv.x, dv.x, elt), elt1(av.y, bv.y, dv.y, elt), elt1(av.z, bv.z, dv.z, elt), elt1(av.w, bv.w, dv.w, elt));
    st4(y, p * y.stride + ch, fp_act4(v, act));
  }
}

// ---- bilinear resize (TFLite reference association) -----------------------------------------------------------------------------
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
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c);
  const int xs = x.stride, ys = y.stride, H = op.H, W = op.W, OW = op.OW;
  float hs = (float)H / (float)op.OH, ws = (float)W / (float)OW;
  if (op.align_corners && op.OH > 1) hs = (float)(H - 1) / (float)(op.OH - 1);
  if (op.align_corners && OW > 1) ws = (float)(W - 1) / (float)(OW - 1);
  const bool vec = (op.Cin & 3) == 0, hp = op.half_pixel;
  const int CV = vec ? op.Cin >> 2 : op.Cin, total = op.OH * OW * CV;
  const int rows = kFrameThreads / CV, cv = threadIdx.x % CV, r0 = threadIdx.x / CV, P = total / CV;
  const int step_y = rows / OW, step_x = rows - step_y * OW;
  int oy = r0 / OW, ox = r0 - oy * OW;
  if (r0 < rows)
  for (int p = r0; p < P; p += rows, ox += step_x, oy += step_y) {
    if (ox >= OW) { ox -= OW; oy++; }
    float dy, dx; int y0, y1, x0, x1;
    fp_interp(oy, hs, hp, H, &dy, &y0, &y1);
    fp_interp(ox, ws, hp, W, &dx, &x0, &x1);
    if (vec) {
      const int ch = cv * 4;
      const float4 a = ld4(x, (y0 * W + x0) * xs + ch), b = ld4(x, (y1 * W + x0) * xs + ch);
      const float4 cc = ld4(x, (y0 * W + x1) * xs + ch), d = ld4(x, (y1 * W + x1) * xs + ch);
      st4(y, p * ys + ch, make_float4(fp_bilerp(a.x, b.x, cc.x, d.x, dy, dx), fp_bilerp(a.y, b.y, cc.y, d.y, dy, dx),
                                      fp_bilerp(a.z, b.z, cc.z, d.z, dy, dx), fp_bilerp(a.w, b.w, cc.w, d.w, dy, dx)));
    } else {
      st1(y, p * ys + cv, fp_bilerp(ld1(x, (y0 * W + x0) * xs + cv), ld1(x, (y1 * W + x0) * xs + cv), ld1(x, (y0 * W + x1) * xs + cv),
                                    ld1(x, (y1 * W + x1) * xs + cv), dy, dx));
    }
  }
}

__device__ __forceinline__ void mo_concat(cop_t& op, const FrameCtx& c) {
  const Ref y = make_ref(op.out, c);
  const int P = op.OH * op.OW;
  int coff = 0;
  for (int k = 0; k < op.n_cat; k++) {
    const Ref x = make_ref(op.cat[k], c);
    const int C4 = op.cat_c[k] >> 2, total = P * C4;
    const int rows = kFrameThreads / C4, cq = threadIdx.x % C4, r0 = threadIdx.x / C4;
    (void)total;
    if (r0 < rows)
      for (int p = r0; p < P; p += rows) st4(y, p * y.stride + coff + cq * 4, ld4(x, p * x.stride + cq * 4));
    coff += op.cat_c[k];
  }
}

// ---- Convolution2DTransposeBias, kernel == stride ------------------------------------------------------------------------------------
__device__ __forceinline__ void mo_tconv(cop_t& op, const FrameCtx& c) {
  const Ref x = make_ref(op.in0, c), y = make_ref(op.out, c);
  const float* w = c.weights + op.w_off;
  const float* bias = c.weights + op.b_off;
  const int C4 = op.Cin >> 2, P = op.OH * op.OW, Cout = op.Cout, Cin = op.Cin, kh = op.kh, kw = op.kw, OW = op.OW, W = op.W;
  const bool staged = op.stage_floats > 0;
  const lds_f* wl = lds_base() + op.w_lds;
  const lds_f* bl = wl + (int)(op.b_off - op.w_off);
  const glb_f* wg = (const glb_f*)w;
  for (int p = threadIdx.x; p < P; p += kFrameThreads) {
    const int oy = p / OW, ox = p - oy * OW;
    const int iy = oy / kh, fy = oy % kh, ix = ox / kw, fx = ox % kw;
    const int xo = (iy * W + ix) * x.stride;
    for (int oc = 0; oc < Cout; oc++) {
      const int wo = ((fy * kw + fx) * Cout + oc) * Cin;
      float acc = staged ? bl[oc] : as_const(bias)[oc];
      for (int q = 0; q < C4; q++) {
        const float4 xv = ld4(x, xo + q * 4);
        const float4 wv = ldw4(staged, wl, wg, wo + q * 4);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
      }
      st1(y, p * y.stride + oc, fp_act(acc, op.act));
    }
  }
}

// One wave instruction moves 64 lanes x 16 B = 256 floats: LDS address = M0 (wave-uniform base) + lane * 16.
// the [bias | weights] blocks of a squeeze-excite op's FC layers (MicroOp::fc_stage)
__device__ __forceinline__ void stage_fc_async(cop_t& op, const glb_f* gw) {
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  const int lane4 = (int)(threadIdx.x & 63) * 4;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int sf = op.fc_stage[k];
    if (sf == 0) continue;
    const glb_f* src = gw + (k == 0 ? op.b_off : op.b3_off);
    lds_f* dst = lds_base() + op.fc_lds[k];
    for (int c0 = wave_id() * 256; c0 < sf; c0 += (kFrameThreads >> 6) * 256)
      if (c0 + lane4 < sf) __builtin_amdgcn_global_load_lds((glb_vp)(src + c0 + lane4), (lds_vp)(dst + c0), 16, 0, 0);
  }
}

__device__ __forceinline__ void stage_weights_async(cop_t& op, const glb_f* gw) {
  const int sf = op.stage_floats;            // multiple of 4; source and slot are 16-byte aligned
  if (sf == 0) { stage_fc_async(op, gw); return; }
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  const int lane4 = (int)(threadIdx.x & 63) * 4;
  const glb_f* src = gw + op.w_off;
  lds_f* dst = lds_base() + op.w_lds;
  for (int c0 = wave_id() * 256; c0 < sf; c0 += (kFrameThreads >> 6) * 256)
    if (c0 + lane4 < sf) __builtin_amdgcn_global_load_lds((glb_vp)(src + c0 + lane4), (lds_vp)(dst + c0), 16, 0, 0);
}
__global__ __launch_bounds__(kFrameThreads) void frame_program_k(const MicroOp* __restrict__ ops, int n_ops, float* arena, long per_frame_floats,
                                                                float* net_in, float* net_out, const float* __restrict__ weights,
                                                                unsigned long long* timeline, int repeat) {
  FrameCtx c{arena + (size_t)blockIdx.x * (size_t)per_frame_floats, net_in, net_out, weights, (int)blockIdx.x, timeline};
  // Weight staging: an asynchronous global→LDS DMA (global_load_lds_dwordx4: no VGPRs, tracked by vmcnt) of op i+1's
  // block is issued at the top of op i into the slot the planner reserved for it; the barrier that ends op i (vmcnt(0)
  // first) publishes it.  One barrier per op, nothing carried in registers across ops.
  const glb_f* gw = (const glb_f*)weights;
  stage_weights_async(((cop_t*)ops)[0], gw);
  const bool fine = timeline && blockIdx.x == 0 && (threadIdx.x & 63) == 0;   // per-op fine stamps (shader clock) of lane 0 of every wave of workgroup 0
  for (int rep = 0; rep < repeat; rep++)      // repeat > 1 only in timing experiments (warm caches on the later passes)
  for (int i = 0; i < n_ops; i++) {
    unsigned long long t_a = 0, t_b = 0, t_c = 0;
    if (fine) t_a = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                           // previous op complete, this op's weights in LDS
    if (fine) t_b = __builtin_readcyclecounter();
    if (timeline && blockIdx.x == 0 && threadIdx.x == 0) timeline[i] = wall_clock64();
    cop_t& op = ((cop_t*)ops)[i];
    if (i + 1 < n_ops || rep + 1 < repeat) stage_weights_async(((cop_t*)ops)[i + 1 < n_ops ? i + 1 : 0], gw);
    // Descriptor prefetch: every op's 408-byte descriptor is read exactly once per workgroup, so each op opened with exposed
    // scalar-cache misses.  The LAST wave touches every cache line of the descriptor two ops ahead: it alone stalls on the
    // misses, the other fifteen find the lines in the scalar cache when they get there (measured: 258 -> 232 us per launch).
    if (wave_id() == (kFrameThreads >> 6) - 1 && i + 2 < n_ops) {
      typedef __attribute__((address_space(4))) const int cint_t;
      cint_t* nxt = (cint_t*)(ops + i + 2);
      int touch = 0;
#pragma unroll
      for (int l = 0; l < (int)sizeof(MicroOp) / 64 + 1; l++) touch += nxt[min(l * 16, (int)sizeof(MicroOp) / 4 - 1)];
      if (touch == 0x7f123456) lds_base()[0] = 0.f;        // never true: keeps the loads alive
    }
    if (fine) t_c = __builtin_readcyclecounter();
    switch ((StepKind)op.kind) {
      case StepKind::PwConv:
        if (op.gemv) mo_gemv(op, c);
        else mo_pw(op, c);
        break;
      case StepKind::Conv: if (op.mfma) mo_conv_mfma(op, c); else mo_conv(op, c); break;
      case StepKind::DwConv: mo_dw(op, c); break;
      case StepKind::Gap: mo_gap(op, c); break;
      case StepKind::Eltwise: mo_elt(op, c); break;
      case StepKind::Resize: mo_resize(op, c); break;
      case StepKind::Concat: mo_concat(op, c); break;
      case StepKind::TConv: mo_tconv(op, c); break;
      default: if (op.kind == kMicroSe) mo_se(op, c); else if (op.kind == kMicroTail) mo_tail(op, c); break;
    }
    if (fine && i < 64) {                      // [1024 + (op * 16 + wave) * 4]: wait+barrier, weight-DMA issue, body (cycles)
      const unsigned long long t_d = __builtin_readcyclecounter();
      unsigned long long* f4 = timeline + 1024 + (size_t)(i * 16 + (threadIdx.x >> 6)) * 4;
      f4[0] = t_b - t_a; f4[1] = t_c - t_b; f4[2] = t_d - t_c; f4[3] = t_a;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (timeline && blockIdx.x == 0 && threadIdx.x == 0) timeline[n_ops] = wall_clock64();
}

}  // namespace

hipError_t frame_program_prepare(int lds_floats) {
  // The attribute belongs to the process-global kernel, not to a context: always raise it to the full 160 KiB so that a
  // context with a small program can never lower the limit under a live context with a larger one.
  if (lds_floats > kLdsTotalFloats) return hipErrorInvalidValue;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(frame_program_k), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotalFloats * (int)sizeof(float));
}

hipError_t launch_frame_program(const MicroOp* d_ops, int n_ops, int lds_floats, float* arena, long per_frame_floats, float* net_in, float* net_out,
                                const float* weights, int n, hipStream_t s, unsigned long long* timeline) {
  static const int repeat = BSX_DBG_ENV("BSX_PROGRAM_REPEAT") ? atoi(BSX_DBG_ENV("BSX_PROGRAM_REPEAT")) : 1;
  frame_program_k<<<n, kFrameThreads, (size_t)lds_floats * sizeof(float), s>>>(d_ops, n_ops, arena, per_frame_floats, net_in, net_out, weights, timeline,
                                                                               repeat > 0 ? repeat : 1);
  return hipGetLastError();
}

}  // namespace bsx
