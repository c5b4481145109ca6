// segments.hpp — the spatially-parallel "segment" kernels around the per-frame program.
//
// Interpreter::Invoke() (/root/reference/lib/libbackscrub.cc:307) for the Meet / MLKit family is cut where the
// graph has GLOBAL dependencies (squeeze-excite and decoder-gate average pools).  Between two such cuts every
// operator is spatially local, so the high-resolution ends of the network — where a frame offers thousands of
// independent pixels — run as ordinary tiled kernels (many 256-lane workgroups per frame, halo recompute through
// LDS, several workgroups per CU hiding each other's latency), and only the low-resolution middle, whose tensors
// fit one CU's LDS, stays a one-workgroup-per-frame program (frame_program.hpp):
//
//   head   stem conv3x3/s2 → 1x1 → depthwise3x3/s2                       (+ pooled partial sums of A and b0)
//   k2     SE gate(b0) → 1x1 (skip B) → 1x1 expand → depthwise3x3/s2     (+ partial sums of B)
//   middle per-frame LDS program: the rest of the encoder, head, decoder levels 3 and 2 (gate of level 2 included)
//   k3     B·g + up(lo2) → 1x1 → dw3x3 + residual → 1x1 (lo)            (+ partial sums of lo)
//   tail   gate(A, lo) → A·g + up(lo) → 1x1 → dw3x3 + residual → Convolution2DTransposeBias 2x2 [→ decode + IIR]
//
// The pooled means the gates need are produced as per-tile partial sums by the kernel that writes the pooled tensor
// and reduced by the consumer; GAP(resize2x(x)) == GAP(x) for the half-pixel 2x up-sampling these graphs use (every
// source pixel carries the same total interpolation weight — verified on the host, otherwise the plan stays unsegmented),
// so the up-sampled tensors are never materialised: each consumer interpolates the low-resolution tensor on the fly.
//
// Plain PODs shared by the planner (plan.cpp) and the device code (kernels_seg.hip); passed to the kernels BY VALUE
// (kernarg → SGPRs: no descriptor fetches on the critical path).
#pragma once
#ifndef BSX_SEG_RTC               // (the text of this header is also part of the graph-specialised segment kernels' translation unit: gen_seg.cpp defines BSX_SEG_RTC)
#include <algorithm>
#include <cmath>
#include <cstdint>
#endif

namespace bsx {

struct SegConvW { long long w_off = 0, b_off = 0; int Cin = 0, Cout = 0, cout_pad = 0, act = 0; };   // dense / 1x1: [k][cout_pad], bias [cout_pad]
struct SegDwW { long long w_off = 0, b_off = 0; int C = 0, act = 0; };                               // depthwise 3x3: [fy][fx][C], bias [C]
struct SegFc { long long w_off = 0, b_off = 0; int Cin = 0, Cout = 0, act = 0; };                    // [co][ci] rows
struct SegPart { long long off = 0; int n = 0, C = 0; float hw = 1.f; };                             // partial sums [n][C] in the frame's arena slice; mean = sum / hw
struct SegGate {                                                                                      // mean parts (concatenated or summed) → FC → [FC]
  int n_parts = 0, sum_parts = 0, n_fc = 0;
  int timing_skip = 0;                // BSX_SEG_GATE_SKIP=1 (timing experiment, results invalid): the gate prologue is replaced by a constant vector
  SegPart part[2];
  SegFc fc[2];
};

constexpr int kSegThreads = 256;
constexpr int kSegMaxC = 16;        // channel count of every tensor at the segment boundaries (A, b0, B, lo2, lo)

struct SegHead {
  int H0 = 0, W0 = 0, H1 = 0, W1 = 0, H2 = 0, W2 = 0;     // network input, A (stem output), b0 (depthwise output)
  int stem_pt = 0, stem_pl = 0, dw_pt = 0, dw_pl = 0;
  SegConvW stem, pw;
  SegDwW dw;
  long long a_off = 0, b0_off = 0, part_a_off = 0, part_b0_off = 0;
  int TR = 0, TC = 0, tiles_y = 0, tiles_x = 0;           // tile of b0 pixels per workgroup
  int lds_floats = 0;
  int dbg_skip = 0;                                       // BSX_SEG_SKIP (timing experiments, results invalid): bit mask of phases this kernel skips
  int rw = 0;                                             // LDS row width of the A region: 16 * ceil((2TC+1) / 16)
  unsigned m_ct = 0;                                      // ceil(65536 / (rw / 16))
};

struct SegK2 {
  int H2 = 0, W2 = 0, H3 = 0, W3 = 0;                     // b0 / B resolution, c0 resolution
  int dw_pt = 0, dw_pl = 0;
  SegGate gate;                                           // s1 = gate(GAP(b0))
  SegConvW pw_a, pw_b;                                    // B = pw_a(b0 * s1); x = act(pw_b(B))
  SegDwW dw;                                              // c0 = act(dw3x3/s2(x))
  long long b0_off = 0, B_off = 0, c0_off = 0, part_B_off = 0;
  int TR = 0, TC = 0, tiles_y = 0, tiles_x = 0;           // tile of c0 pixels
  int lds_floats = 0;
  int dbg_skip = 0;                                       // BSX_SEG_SKIP (timing experiments, results invalid): bit mask of phases this kernel skips
  int rw = 0;                                             // LDS row width of the B region: 16 * ceil((2TC+1) / 16)
  unsigned m_ct = 0;
};

struct SegK3 {
  int H2 = 0, W2 = 0, HL = 0, WL = 0;                     // output resolution, resolution of lo2
  int half_pixel = 0, align_corners = 0;
  SegConvW pw1, pw2;                                      // z = act(pw1(skip*g + up(lo2))); lo = pw2(z + act(dw(z)))
  SegDwW dw;
  long long skip_off = 0, lo2_off = 0, g_off = 0, lo_off = 0, part_lo_off = 0;
  int TR = 0, TC = 0, tiles_y = 0, tiles_x = 0;
  int lds_floats = 0;
  int lo_floats = 0;                                      // LDS floats of the staged low-resolution window: the largest any tile of THIS geometry needs (seg_lo_window_floats)
  float hs = 0.f, ws = 0.f;                               // the two up-sampling scales ((float)in / out, or (in - 1) / (out - 1) with align_corners): launch constants, so the planner divides
  int dbg_skip = 0;                                       // BSX_SEG_SKIP (timing experiments, results invalid): bit mask of phases this kernel skips
};

struct SegTail {
  int H1 = 0, W1 = 0, HL = 0, WL = 0, H0 = 0, W0 = 0;     // A resolution, lo resolution, network output resolution
  int half_pixel = 0, align_corners = 0;
  SegGate gate;                                           // g = gate(GAP(A) | GAP(lo))
  SegConvW pw;                                            // z = act(pw(A*g + up(lo)))
  SegDwW dw;                                              // t = z + act(dw(z))
  long long tc_w_off = 0, tc_b_off = 0;                   // Convolution2DTransposeBias 2x2: [fy][fx][oc][ic], bias [oc]
  int Co = 0, act3 = 0, model_type = 0;
  long long skip_off = 0, lo_off = 0;
  long long pre_gate_off = -1;                            // >= 0: the gate vector (16 floats per frame) was computed ONCE per frame by seg_gate_k and lives here in the arena
  int TR = 0, TC = 0, tiles_y = 0, tiles_x = 0;
  int lds_floats = 0;
  int lo_floats = 0;                                      // as SegK3::lo_floats
  float hs = 0.f, ws = 0.f;                               // as SegK3::hs / ws
  int dbg_skip = 0;                                       // BSX_SEG_SKIP (timing experiments, results invalid): bit mask of phases this kernel skips
};

// LDS floats each kernel needs for the tile sizes in its descriptor (the planner picks the tiles against these; the kernels
// carve the same regions)
constexpr int kSegLoStride = 16;                 // floats per pixel of the staged low-resolution window (dense; bank conflicts are removed by a swizzle: kernels_seg.hip swz_l)
constexpr int kSegLoTileFloats = 12 * 16 * kSegLoStride;   // staged window of the low-resolution tensor a k3 / tail tile interpolates from: <= 12 rows x 16 columns
constexpr int kSegScratchFloats = 640;   // gate vector / means / hidden / partial-sum meeting points
constexpr int kSegGateStageFloats = 512 + 2 * (32 * 32 + 32);   // gate prologue staging (aliases the first tile region)
#ifndef BSX_SEG_RTC               // host-side sizing (the planner); the device code gets the results through the descriptors
inline int seg_row_width(int cols) { return (cols + 15) / 16 * 16; }
inline int seg_head_lds_floats(const SegHead& d) {
  const int AR = 2 * d.TR + 1, AC = 2 * d.TC + 1, IR = 2 * AR + 1, IC = 2 * AC + 1;
  const int win = (IR * IC * 3 + 3) & ~3;                                  // input window (>= 512 floats: it doubles as the partial-sum meeting points at the end)
  return (win > 512 ? win : 512) + AR * AC * 16;                           // + x = act(pw(stem)) (the stem output itself stays in registers: seg_head_k)
}
inline int seg_k2_lds_floats(const SegK2& d) { const int v = 2 * (2 * d.TR + 1) * seg_row_width(2 * d.TC + 1) * 16; return kSegScratchFloats + (v > kSegGateStageFloats ? v : kSegGateStageFloats); }
// The staged window of the low-resolution tensor a k3 / tail tile interpolates from ([LR][LC][20] floats) was reserved at its worst case (12 x 16 pixels = 15 KB) for
// every geometry; a 2x up-sampling tile of 16 x 14 pixels reads 10 x 9.  The planner now walks the tiles of the actual geometry with the kernels' own index
// arithmetic (up_axis, TFLite's clamping) and reserves the largest window + one row and one column of margin: k3 43.5 -> 33 KB (4 workgroups per CU instead
// of 3), tail 35.5 -> 28 KB (5 instead of 4 where its registers allow).
inline void seg_up_axis(int o, float scale, bool half_pixel, int in_size, int* lo, int* hi) {
  const float v = half_pixel ? ((float)o + 0.5f) * scale + -0.5f : (float)o * scale;       // compiled with -ffp-contract=off: the two roundings of the device code
  const float fl = std::floor(v);
  *lo = std::max((int)fl, 0);
  *hi = std::min((int)std::ceil(v), in_size - 1);
}
inline float seg_up_scale(int in, int out, bool align) { return (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out; }
inline int seg_lo_window_floats(int H, int W, int HL, int WL, bool half_pixel, bool align, int TR, int TC, int tiles_y, int tiles_x) {
  const float hs = seg_up_scale(HL, H, align), ws = seg_up_scale(WL, W, align);
  int lr = 1, lc = 1, a, b, c, e;
  for (int ty = 0; ty < tiles_y; ty++) {
    const int r0 = ty * TR;
    seg_up_axis(std::max(r0 - 1, 0), hs, half_pixel, HL, &a, &b);
    seg_up_axis(std::min(r0 + TR, H - 1), hs, half_pixel, HL, &c, &e);
    lr = std::max(lr, e - a + 1);
  }
  for (int tx = 0; tx < tiles_x; tx++) {
    const int c0 = tx * TC;
    seg_up_axis(std::max(c0 - 1, 0), ws, half_pixel, WL, &a, &b);
    seg_up_axis(std::min(c0 + TC, W - 1), ws, half_pixel, WL, &c, &e);
    lc = std::max(lc, e - a + 1);
  }
  const int need = (lr + 1) * (lc + 1) * kSegLoStride;
  return need < kSegLoTileFloats ? need : kSegLoTileFloats;
}
inline int seg_k3_lds_floats(const SegK3& d) { return kSegScratchFloats + (d.TR + 2) * 256 + (d.lo_floats > 0 ? d.lo_floats : kSegLoTileFloats); }   // z tile + the window (t stays in registers)
inline int seg_tail_lds_floats(const SegTail& d) { const int v = (d.TR + 2) * 256; return kSegScratchFloats + (v > kSegGateStageFloats ? v : kSegGateStageFloats) + (d.lo_floats > 0 ? d.lo_floats : kSegLoTileFloats); }
#endif

struct SegPlan {
  bool on = false;
  SegHead head;
  SegK2 k2;
  SegK3 k3;
  SegTail tail;
};

}  // namespace bsx
