// plan.hpp — fused GPU execution plan for one segmentation network.
//
// The reference runs the graph op by op through tflite::Interpreter::Invoke()
// (/root/reference/lib/libbackscrub.cc:307).  Here the graph is compiled once, at
// bsx_new(), into a short list of batched kernel launches ("steps"): activations and
// residual adds are folded into the producing convolution, the squeeze-excite MUL is
// folded into the consuming 1x1 convolution, gate*skip+up becomes one pass, and all
// activation tensors live in one arena with liveness-based reuse so a batch of streams
// stays resident in the Infinity Cache between layers.
#pragma once
#include "debug_switches.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <string>
#include <vector>

#include "frame_program.hpp"
#include "segments.hpp"
#include "tflite_model.hpp"

namespace bsx {

enum class StepKind : int { Conv = 0, PwConv, DwConv, Gap, Eltwise, Resize, Concat, TConv };
enum EltOp : int { kEltAdd = 0, kEltMul = 1, kEltUnary = 2, kEltMulAdd = 3 };

struct Step {
  StepKind kind = StepKind::Conv;
  std::string label;
  int in0 = -1, in1 = -1, in2 = -1;   // graph tensor ids
  int out = -1;
  int residual = -1;                  // tensor added after the activation (conv-type steps)
  int in_scale = -1;                  // [N,1,1,Cin] tensor multiplied into the input (PwConv only)
  int out_bias = -1;                  // [N,1,1,Cout] per-frame vector added to the accumulator before the activation (PwConv only:
                                      // a spatially constant concat branch folded into the consuming 1x1 convolution)
  int H = 1, W = 1, Cin = 1, OH = 1, OW = 1, Cout = 1;
  int kh = 1, kw = 1, sh = 1, sw = 1, dh = 1, dw = 1, pad_t = 0, pad_l = 0;
  int act = kActNone;
  int elt = kEltAdd;
  bool bcast1 = false;                // in1 is [N,1,1,C]
  bool align_corners = false, half_pixel = false;
  int cout_pad = 0;                   // padded Cout of the packed weights
  int cout_tile = 16;                 // output channels per thread in the conv kernels
  size_t w_off = 0, b_off = 0;        // float offsets into the weight arena
  size_t w2_off = 0;                  // [co][ci] copy of the weights for single-pixel (GEMV) steps, 0 if absent
  size_t w16_off = 0;                 // Plan::weights16 offset (halves) of the split-f16 copy [hi | lo][cout_pad][k16_pad], k contiguous
  int k16_pad = 0;                    // Cin rounded up to 32 (0: the step has no f16 copy)
  std::vector<int> concat_in;         // Concat: all inputs
  std::vector<int> concat_c;          // Concat: channels of each input
  bool gap_sum = false;               // Gap over concat_in as a SUM of the parts' means (GAP(a + b) rewritten), not their concatenation
  double macs = 0;                    // per frame
  int last_node = -1;                 // file operator index of the last fused op
  int fuse_dw = -1;                   // per-launch path, PwConv: index of the depthwise step this expand convolution is fused with (ir_expand_dw_k)
  bool in_from_fused_dw = false;      // per-launch path, PwConv: in0 is the output of a fused expand+depthwise pair and has no other reader (may be stored as f16)
  bool fused_away = false;            // per-launch path: the step runs inside an earlier one
  bool fuse_head0 = false;            // per-launch path, stem Conv: runs together with the depthwise and the 1x1 after it (dl_head0_k)
  // per-launch path, three chained 1x1 convolutions a → b → c as one kernel (kernels_nn.hip: pw_chain3_k), launched at b's place (b's per-frame bias exists by then):
  int chain_first = -1, chain_last = -1;   // on b: the indices of a and c
  int chain_mid = -1;                 // on a and c: the index of b (the step is skipped while the chain runs — chain3_on())
  size_t chain_w16_off = 0;           // on b: Plan::weights16 offset (halves) of the chain's weight stream
};
// The weight stream of a chain, in the order pw_chain3_k's waves consume it: 1 KB operand tiles [64 lanes][8 halves], lane (li, g) = output channel li of the
// tile, K values 8g .. 8g+7 of the slab (stage 1: natural order; stages 2 and 3: position i < 4 → slab channel 4g + i, i >= 4 → 16 + 4g + (i - 4)).
constexpr int kChainS0 = 5, kChainP1 = 8, kChainP2 = 8;          // the instantiated shape: 160 → 256 → 256 → (<= 32)
inline size_t chain3_stream_halves() { return (size_t)(kChainS0 * 4 * kChainP1 + kChainP2 * (4 * kChainP1 + 4)) * 512; }

// Geometry of the fused expand + depthwise kernel (kernels_nn.hip: ir_expand_dw_k), shared by the planner (is the pair fusable?) and the
// launcher: CH expanded channels per workgroup, BH depthwise output rows per row band; the band's expanded rows live in LDS.
struct IrGeom { int CH = 0, BH = 0, nbands = 0, rows = 0; };
// LDS of one workgroup: the band of the expanded chunk + depthwise weights, bias and a quad of zeros + (32-channel chunks only) the eight
// per-wave 2 KB buffers the input tiles are re-ordered through (coalesced loads → MFMA fragments)
inline long ir_stage_floats(int CH) { return CH == 32 ? 8 * 512 : 0; }
inline long ir_lds_bytes(int rows, int W, int CH) { return ((long)rows * W * CH + 10 * CH + 4 + ir_stage_floats(CH)) * 4; }
inline IrGeom ir_geometry(int H, int W, int Cexp, int OH, int S, int d) {
  if (const char* e = BSX_DBG_ENV("BSX_IR_GEOM")) {                     // timing experiments: "W:CH,BH" overrides the choice for layers W pixels wide
    int w = 0, ch = 0, bh = 0;
    if (sscanf(e, "%d:%d,%d", &w, &ch, &bh) == 3 && w == W && ch > 0 && Cexp % ch == 0 && bh > 0) {
      IrGeom g; g.CH = ch; g.BH = bh < OH ? bh : OH; g.nbands = (OH + g.BH - 1) / g.BH;
      const int r = S * (g.BH - 1) + 2 * d + 1; g.rows = r < H ? r : H;
      if (ir_lds_bytes(g.rows, W, ch) <= 160 * 1024) return g;
    }
  }
  auto rows_for = [&](int bh) { const int r = S * (bh - 1) + 2 * d + 1; return r < H ? r : H; };
  // 1. the whole frame in one band (no halo rows, every input row read once per chunk): one workgroup per CU
  for (int CH : {32, 24, 16}) {
    if (Cexp % CH) continue;
    if (ir_lds_bytes(rows_for(OH), W, CH) <= 160 * 1024) { IrGeom g; g.CH = CH; g.BH = OH; g.nbands = 1; g.rows = rows_for(OH); return g; }
  }
  // 2. row bands: small enough for TWO workgroups per CU (<= 80 KB each, everything included) — one workgroup's MFMA/global phase then overlaps the other's LDS/VALU phase
  //    (measured on the 129x129 and 65x65 layers: 1.16 -> 0.91 ms, 0.41 -> 0.33 ms against the largest band that fits one workgroup per CU).
  //    Score = useful rows per band row x MFMA column-tile occupancy.
  IrGeom best;
  double best_score = 0;
  for (int CH : {32, 24, 16}) {
    if (Cexp % CH) continue;
    int BH = OH;
    while (BH > 1 && ir_lds_bytes(rows_for(BH), W, CH) > 80 * 1024) BH--;
    if (ir_lds_bytes(rows_for(BH), W, CH) > 80 * 1024 || BH < 2) continue;
    const int nb = (OH + BH - 1) / BH;
    BH = (OH + nb - 1) / nb;                             // even bands
    const double score = (double)(S * BH) / rows_for(BH) * CH / ((CH + 15) / 16 * 16);
    if (score > best_score) { best_score = score; best.CH = CH; best.BH = BH; best.nbands = nb; best.rows = rows_for(BH); }
  }
  return best;
}

// rows of the stem's output one workgroup of dl_head0_k owns (its LDS holds the input rows, the stem band and the depthwise band)
inline long head0_lds_floats(int W0, int W1, int bh) {            // the depthwise band aliases the input rows (dead after the stem)
  const long in_t = (((2l * (bh + 2) + 1) * (W0 + 2) * 3 + 3) & ~3l), dband = (long)bh * W1 * 16;
  return (in_t > dband ? in_t : dband) + (long)(bh + 2) * W1 * 16 + 1024;
}
inline int head0_band_rows(int W0, int W1) {
  if (const char* e = BSX_DBG_ENV("BSX_H0_BH")) { const int bh = atoi(e); if (bh >= 1 && bh <= 8 && head0_lds_floats(W0, W1, bh) * 4 <= 156 * 1024) return bh; }   // timing experiments
  for (int bh = 8; bh >= 2; bh--)                                // two workgroups per CU (their phases overlap) if a band of >= 2 rows allows it
    if (head0_lds_floats(W0, W1, bh) * 4 <= 80 * 1024) return bh;
  for (int bh = 8; bh >= 2; bh--)
    if (head0_lds_floats(W0, W1, bh) * 4 <= 156 * 1024) return bh;
  return 0;
}

struct Plan {
  std::vector<Step> steps;
  std::vector<float> weights;         // packed weight arena (host copy)
  std::vector<uint16_t> weights16;    // IEEE half bit patterns: hi/lo split of the large pointwise-conv weights (split-f16 MFMA GEMM)
  std::vector<long> tensor_off;       // per graph tensor: float offset per stream-slot unit, -1 if not materialised
  size_t arena_floats_per_stream = 0; // arena size = this * n_streams
  int input = -1, output = -1;
  double macs_per_frame = 0;
  // whole-network per-frame program (empty when some step has no micro-op form)
  std::vector<MicroOp> program;
  std::vector<std::string> program_labels;
  // geometry of the workgroup that runs the program for one frame (round 6): lanes per workgroup and the LDS block it may plan into.  1024 lanes + the whole 160 KiB
  // = one frame per CU (rounds 1-5); 512 lanes + 80 KiB = two frames per CU at the same 16 waves.  Chosen per plan by mid_geometry_for() below.
  int mid_lanes = kFrameThreads;
  int lds_total_floats = kLdsTotalFloats;
  int lds_zero_off() const { return lds_total_floats - kLdsZeroFloats; }      // the zero cell is the last 16 bytes of the block the program may use
  int program_lds_floats = 0;         // dynamic LDS the program needs (scratch included)
  int program_scratch_floats = 0;     // reduction scratch at the bottom of the LDS block: kLdsScratchFloats, or 64 when no micro-op of the lowering uses it
  int program_lds_tensors = 0, program_global_tensors = 0;
  std::vector<long> program_ext_offs; // arena offsets of the tensors that cross the program's boundary (read or written by a segment kernel): never elided by the generator
  // every LDS reservation of the program: [off, off+len) floats, alive for steps [from, until] (tensors, weight slots,
  // band workspaces) — checked for overlap by verify_program_lds() at the end of the lowering and by the tests
  struct LdsBlock { int off, len, from, until; std::string what; };
  std::vector<LdsBlock> program_blocks;
  std::string program_check;            // "ok" or the first violation found
  // spatially-parallel segment kernels around the program (segments.hpp); when seg.on, `program` is only the MIDDLE of the network
  SegPlan seg;
  unsigned program_policy = 0;          // the placement policy build_frame_program kept (bit 0 long-lived tensors to the arena, bit 1 expanded tensors elided, bit 2 small tensors top-down)
  long program_arena_bytes = 0;         // arena (HBM / L2) bytes one frame's program touches: every use of an operand that is not LDS-resident
  std::string seg_text;                 // one line per segment kernel (tiles, LDS) for bsx_plan_describe
  std::string describe() const;
};

// Lanes / LDS budget of the per-frame workgroup a plan is built for: 1024 lanes and the whole 160 KiB block.  The debug build reads BSX_MID_LANES = 512 | 1024 and
// BSX_MID_LDS_KB = 64..160 (once per process): round 6's same-box A/B of 512 lanes / 80 KiB — two frames per CU — lost on every configuration
// (profiles/r06b_mid_geometry.txt, docs/design/10-round6.md), so the geometry stays a parameter of the plan and the generator, not a user mode.
struct MidGeometry { int lanes, lds_floats; };
MidGeometry mid_geometry_default();

// Build the plan.  Returns false with `err` for unsupported graph features.
// `reuse_arena=false` gives every tensor its own slot (layer-by-layer debugging).
// "ok", or the first violation among the program's LDS reservations / operands (see Plan::program_blocks)
std::string verify_program_lds(const Plan& plan);

// `segments`: allow the head | k2 | middle | k3 | tail segmentation (segments.hpp) where the graph has that shape; the per-launch
// path executes the plain step list and needs segments = false.
bool build_plan(const Graph& g, Plan* plan, std::string* err, bool reuse_arena = true, bool segments = true);

}  // namespace bsx
