// frame_program.hpp — the "one workgroup per camera frame" network program.
//
// MI355X-first execution of Interpreter::Invoke() (/root/reference/lib/libbackscrub.cc:307):
// instead of one launch per operator over the whole batch, ONE launch runs the whole fused
// step list; workgroup f (1024 lanes = 16 waves = one CU) walks the list for frame f with
// every tensor that fits kept in that CU's 160 KB LDS ([pixel][C+pad] rows, pad chosen so
// 16-byte row reads are bank-conflict free) and only the few large full-resolution tensors
// spilled to that frame's slice of the HBM arena.  Streams are independent, so 256 frames
// occupy the 256 CUs with no inter-workgroup communication at all; squeeze-excite and
// decoder-gate global pools are plain workgroup reductions.
//
// This header is shared by the host planner (plan.cpp) and the device code
// (kernels_frame.hip): plain PODs only.
#pragma once
#include <cstdint>

namespace bsx {

// kLocGlobal = the frame's PRIVATE slice of the HBM arena (frame-major: arena + frame*per_frame + off) — workgroups
// run ahead of each other, so frames must never share arena bytes; kLocInput/kLocOutput = the batch-major network
// input / output buffers shared with the image kernels.
enum LocSpace : int { kLocNone = 0, kLocLds = 1, kLocGlobal = 2, kLocInput = 3, kLocOutput = 4 };

struct Loc {
  int space = kLocNone;
  int off = 0;      // LDS: float offset into the dynamic LDS block; global: float offset inside the frame's arena slice (Plan::tensor_off)
  int stride = 0;   // floats between consecutive pixels
  int elems = 0;    // per-frame element count (global addressing)
};

struct MicroOp {
  int kind = 0;     // StepKind
  int H = 1, W = 1, Cin = 1, OH = 1, OW = 1, Cout = 1;
  int kh = 1, kw = 1, sh = 1, sw = 1, dh = 1, dw = 1, pt = 0, pl = 0;
  int act = 0, elt = 0, bcast1 = 0, align_corners = 0, half_pixel = 0;
  int cout_pad = 0, cout_tile = 16;
  int stage_floats = 0;  // >0: weights[w_off .. w_off+stage_floats) (weights, then bias at b_off-w_off) are copied to LDS at float offset
  int w_lds = 0;         //     w_lds by an asynchronous global→LDS DMA issued while the PREVIOUS op runs (the planner gives the slot a
                         //     lifetime of [previous op, this op] in the same first-fit allocation as the tensors)
  int ws_off = 0, band_rows = 0;   // dense conv on the matrix cores: LDS workspace (float offset) holding a band of input rows; output rows per band
  int mfma = 0;     // 1: pointwise conv runs on v_mfma_f32_16x16x4_f32 with the weight block staged in LDS
  int strip = 0;    // depthwise: 1 = register-strip form (dw_strip), 0 = per-pixel form (BSX_NO_DW_STRIP=1)
  int gemv = 0;     // 1: ≤4 output pixels → wave-per-output-channel dot products with [co][ci] weights
  int n_cat = 0;
  int gap_sum = 0;  // pooling ops: the cat[] parts' means are ADDED into the same Cin channels (GAP(a + b)) instead of concatenated
  long long w_off = 0, b_off = 0, w2_off = 0;
  // fused squeeze-excite / gate chain (kind == kMicroSe): GAP(in0 | cat[]) → FC1 (w2_off, b_off, act, Cout=C1) → FC2 (w3_off, b3_off, act2, C2)
  long long w3_off = 0, b3_off = 0;
  int C1 = 0, C2 = 0, act2 = 0, n_fc = 0;
  int fc_stage[2] = {0, 0};   // > 0: [bias | pad | [co][ci] weights] of FC k (fc_stage[k] floats from b_off / b3_off) are DMA'd to LDS offset fc_lds[k]
  int fc_lds[2] = {0, 0};     //      while the previous micro-op runs, like stage_floats / w_lds (single FC steps use entry 0)
  // fused decoder tail (kind == kMicroTail): z = act(pw(x*s + a)) → t = z + act2(dw3x3(z)) → out = act3(tconv2x2(t)); z lives in an LDS row band
  //   pw: w_off/b_off (Cin → cout_pad), dw: w3_off/b3_off, tconv: w4_off/b4_off (Cout = C2); H,W = tail resolution; ws_off/band_rows = z band
  long long w4_off = 0, b4_off = 0;
  int act3 = 0;
  unsigned magic_w = 0;   // ceil(2^32 / W): floor(n / W) == __umulhi(n, magic_w) for n < 65536 (integer division is ~40 instructions)
  Loc in0, in1, in2, res, scale, out;
  Loc cat[4];
  int cat_c[4] = {0, 0, 0, 0};
  int cat_hw[4] = {0, 0, 0, 0};      // pixels each pooled part is averaged over
  int cat_parts[4] = {0, 0, 0, 0};   // > 0: the part arrives as that many per-tile partial sums [n][C] (written by a segment kernel) instead of a tensor
};

constexpr int kMicroTail = 101;              // MicroOp::kind of the fused pw → dw(+residual) → transpose-conv tail
constexpr int kMicroSe = 100;                // MicroOp::kind of the fused GAP→FC→FC chain
constexpr int kFrameThreads = 1024;          // 16 waves: 4 per SIMD
constexpr int kLdsTotalFloats = 160 * 256;   // 160 KiB
constexpr int kLdsZeroFloats = 4;            // the LAST 16 bytes of the block stay 0.0f for the whole kernel (the specialised program zeroes them once): out-of-image taps of the
constexpr int kLdsZeroOff = kLdsTotalFloats - kLdsZeroFloats;   // depthwise bodies read them through an address select instead of zeroing data registers (round 5)
constexpr int kLdsScratchFloats = 2048;      // reduction scratch at the start of the LDS block (pooling partials, gemv partials, the tail's small weight set)
constexpr int kLdsMaxStageFloats = 4224;     // largest weight block staged in LDS (128x32 weights + 128 bias)

}  // namespace bsx
