// plan.cpp — graph → fused step list + weight packing + activation arena layout.
#include "debug_switches.hpp"
#include "plan.hpp"
#include "gen_seg.hpp"
#include "gen_mid.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace bsx {
namespace {

// TFLite SAME/VALID geometry (ComputeOutSize / ComputePaddingHeightWidth): SAME → ceil(in/s),
// total pad = max(0,(out-1)s + (k-1)d + 1 - in), leading pad = total/2 (extra goes bottom/right).
void conv_geometry(int in, int k, int s, int d, bool same, int* out, int* pad) {
  int eff = (k - 1) * d + 1;
  *out = same ? (in + s - 1) / s : (in + s - eff) / s;
  int total = (*out - 1) * s + eff - in;
  *pad = total > 0 ? total / 2 : 0;
}

// float → IEEE half, round to nearest even (host side, weights only)
uint16_t f32_to_f16_rn(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));       // inf / nan
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);                                          // clamps to the largest finite half
  if (x < 0x33000001u) return (uint16_t)sign;                                                        // below half of the smallest subnormal
  if (x < 0x38800000u) {                                                                             // subnormal half
    const int e = (int)(x >> 23);
    const uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;                    // 14..24
    uint32_t h = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((x - 0x38000000u) >> 13);
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
  return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 1023;
  uint32_t bits;
  if (e == 31) bits = sign | 0x7f800000u | (m << 13);
  else if (e) bits = sign | ((e + 112) << 23) | (m << 13);
  else if (!m) bits = sign;
  else { int sh = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; sh++; } bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13); }
  float f; memcpy(&f, &bits, 4); return f;
}

bool is_unary(OpType t) { return t == OpType::Relu || t == OpType::Relu6 || t == OpType::HardSwish || t == OpType::Logistic; }
int unary_act(OpType t) {
  switch (t) {
    case OpType::Relu: return kActRelu;
    case OpType::Relu6: return kActRelu6;
    case OpType::HardSwish: return kActHswish;
    case OpType::Logistic: return kActSigmoid;
    default: return kActNone;
  }
}
const char* act_name(int a) {
  switch (a) { case kActRelu: return "relu"; case kActRelu6: return "relu6"; case kActHswish: return "hswish"; case kActSigmoid: return "sigmoid"; default: return "-"; }
}
bool same_dims(const TensorInfo& a, const TensorInfo& b) { for (int i = 0; i < 4; i++) if (a.dims[i] != b.dims[i]) return false; return true; }
bool is_chan_vec(const TensorInfo& v, const TensorInfo& full) { return v.dims[1] == 1 && v.dims[2] == 1 && v.dims[3] == full.dims[3] && (full.dims[1] * full.dims[2] > 1); }

int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

std::string Plan::describe() const {
  static const char* kn[] = {"conv", "pwconv", "dwconv", "gap", "eltwise", "resize", "concat", "tconv"};
  std::string s;
  char line[512];
  int i = 0;
  for (const Step& st : steps) {
    snprintf(line, sizeof line, "%3d %-7s %-26s in %dx%dx%d -> out %dx%dx%d k%dx%d s%d d%d act=%s res=%d scale=%d macs=%.0f t%d->t%d\n", i++,
             kn[(int)st.kind], st.label.c_str(), st.H, st.W, st.Cin, st.OH, st.OW, st.Cout, st.kh, st.kw, st.sh, st.dh, act_name(st.act),
             st.residual, st.in_scale, st.macs, st.in0, st.out);
    s += line;
    if (st.fuse_head0) { s += "      ^ fused with steps 1 and 2 (stem + depthwise + 1x1 in one tiled kernel)\n"; }
    if (st.chain_first >= 0) {
      snprintf(line, sizeof line, "      ^ chained with steps %d and %d at 8192 pixels and more (three 1x1 convolutions in one kernel: the tensors between them stay in registers)\n", st.chain_first, st.chain_last);
      s += line;
    }
    if (st.fuse_dw >= 0) {
      const Step& dd = steps[st.fuse_dw];
      const IrGeom ig = ir_geometry(st.OH, st.OW, st.Cout, dd.OH, dd.sh, dd.dh);
      snprintf(line, sizeof line, "      ^ fused with step %d (expand + depthwise in one kernel: %d channels x %d rows per workgroup, %d band(s))\n", st.fuse_dw, ig.CH, ig.BH, ig.nbands);
      s.insert(s.size() - 1, "");
      s += line;
    }
  }
  return s;
}

// Lower the step list into the per-frame program: one MicroOp per step, every tensor that fits placed in
// LDS (liveness-based first fit inside the 160 KiB block), the rest in the frame's slice of the HBM arena.
// `steps`: the step list to lower (the whole network, or the middle of a segmented plan).  `ext`: tensors that cross the
// program's boundary (produced or consumed by a segment kernel) — they live at their arena offset, never in LDS.
// `part_n[t]` > 0 marks tensor t of a pooling step as "already pooled per tile": [part_n][C] partial sums at tensor_off[t].
static void lower_frame_program(const Graph& g, Plan* plan, const std::vector<Step>& steps, const std::vector<int>& ext, const std::map<int, int>& part_n,
                                const std::map<int, int>& part_hw, const int scratch, const unsigned policy = 0) {
  const bool long_to_hbm = (policy & 1u) != 0, elide_expand = (policy & 2u) != 0, small_top = (policy & 4u) != 0;
  const int kLdsTotal = plan->lds_total_floats, kLdsZero = plan->lds_zero_off();      // the block THIS plan's workgroup owns (Plan::mid_lanes / lds_total_floats)
  plan->program.clear();
  plan->program_scratch_floats = scratch;
  plan->program_labels.clear();
  plan->program_blocks.clear();
  plan->program_check.clear();
  plan->program_ext_offs.clear();
  for (int t : ext) if (t >= 0 && plan->tensor_off[t] >= 0) plan->program_ext_offs.push_back(plan->tensor_off[t]);
  const int NS = (int)steps.size();
  const int NT = (int)g.tensors.size();
  std::vector<int> last(NT, -1);
  for (int s = 0; s < NS; s++) {
    const Step& st = steps[s];
    // steps without a micro-op form → no program (the per-launch path is used instead)
    if ((st.kind == StepKind::PwConv || st.kind == StepKind::Eltwise || st.kind == StepKind::DwConv || st.kind == StepKind::Gap ||
         st.kind == StepKind::TConv) && (st.Cin % 4)) return;
    if (st.out_bias >= 0) return;                          // per-frame output bias exists in the per-launch kernels only
    for (int t : {st.in0, st.in1, st.in2, st.residual, st.in_scale, st.out}) if (t >= 0) last[t] = s;
    for (int t : st.concat_in) last[t] = s;
  }
  last[g.output] = NS + 1;
  for (int t : ext) if (t >= 0) last[t] = NS + 1;          // consumed after the program ends
  // Policy `elide_expand` (see place()): steps whose output is never materialised because the depthwise of the next step computes it chunk by chunk.  Decided up front,
  // from sizes alone, because the fused form runs the 1x1 INSIDE the next step: its staged weights and its LDS operands must stay alive one step longer.
  auto padded = [&](int t) { const TensorInfo& ti = g.tensors[t]; const int C = ti.dims[3], P = ti.dims[1] * ti.dims[2]; return (P * (C + (((C / 4) % 2 == 0) ? 4 : 8)) + 3) / 4 * 4; };
  std::vector<char> elide(NS, 0);
  if (elide_expand)
    for (int s = 0; s + 1 < NS; s++) {
      const Step& a = steps[s];
      const Step& d = steps[s + 1];
      if (!(a.kind == StepKind::PwConv && a.residual < 0 && a.out >= 0 && last[a.out] == s + 1 && d.kind == StepKind::DwConv && d.in0 == a.out && d.residual < 0 && d.dh == 1 && d.dw == 1)) continue;
      if (a.out == g.output || std::find(ext.begin(), ext.end(), a.out) != ext.end() || g.tensors[a.out].dims[3] % 8 || a.cout_pad < (g.tensors[a.out].dims[3] + 15) / 16 * 16) continue;
      const int P = g.tensors[a.out].dims[1] * g.tensors[a.out].dims[2], need = padded(a.out), need_o = padded(d.out), ws = (P * 20 + 3) / 4 * 4 /* the 16-channel chunk */, room = kLdsTotal - scratch - 2 * kLdsMaxStageFloats;
      if (need + need_o > room && (need_o > room ? ws : need_o + ws) <= room) {     // (a depthwise output too large for LDS lives in the arena either way: then only the chunk workspace must fit)
        elide[s] = 1;
        for (int t : {a.in0, a.in2, a.in_scale}) if (t >= 0) last[t] = std::max(last[t], s + 1);
      }
    }
  const bool no_lds = BSX_DBG_ENV("BSX_PROGRAM_NO_LDS") != nullptr;  // debugging: every tensor in the HBM arena
  std::vector<Loc> loc(NT);
  struct Blk { int off, len, until; };
  std::vector<Blk> live;
  // ---- reserved zone for the long-lived tensors (see place()): stacked from the top of the block, at most a quarter of it
  std::map<int, int> reserved_at;
  int reserved = 0;
  std::vector<int> firstdef(NT, -1);
  for (int s = 0; s < NS; s++) if (steps[s].out >= 0 && firstdef[steps[s].out] < 0) firstdef[steps[s].out] = s;
  if (!BSX_DBG_ENV("BSX_PLAN_NO_TOPDOWN")) {
    for (int t = 0; t < NT; t++) {
      if (firstdef[t] < 0 || last[t] - firstdef[t] <= 8 || t == g.input || t == g.output) continue;
      if (std::find(ext.begin(), ext.end(), t) != ext.end()) continue;
      const TensorInfo& ti = g.tensors[t];
      const int C = ti.dims[3], P = ti.dims[1] * ti.dims[2];
      if (C % 4 || P <= 1) continue;
      const int pad = ((C / 4) % 2 == 0) ? 4 : 8, need = (P * (C + pad) + 3) / 4 * 4;
      if (reserved + need > kLdsTotal / 4) continue;
      reserved += need;
      reserved_at[t] = kLdsZero - reserved;
    }
  }
  const int cap = kLdsZero - reserved;          // everything that is not reserved allocates in [scratch, cap); [kLdsZero, kLdsTotal) is the zero cell
  int high = scratch;
  auto place = [&](int t, int s) {
    if (t < 0 || loc[t].space != kLocNone) return;
    const TensorInfo& ti = g.tensors[t];
    int C = ti.dims[3], P = ti.dims[1] * ti.dims[2];
    Loc l;
    l.elems = (int)ti.elems();
    bool lds_ok = (C % 4 == 0) && t != g.input && t != g.output && !no_lds;
    int pad = ((C / 4) % 2 == 0) ? 4 : 8;          // (C+pad)/4 odd → 16-byte row reads hit distinct bank quads
    int stride = P > 1 ? C + pad : C;
    int need = (P * stride + 3) / 4 * 4;
    // a tensor that would leave no room for the weight slots of the ops around it turns those ops into their slow unstaged
    // forms (MLKit's 16x16x128 tensors are 132 KB): such a tensor goes to HBM instead
    if (need > kLdsTotal - scratch - 2 * kLdsMaxStageFloats && !BSX_DBG_ENV("BSX_PLAN_NO_SLOT_RESERVE")) lds_ok = false;
    // Policy `long_to_hbm`: a LONG-LIVED tensor (a skip connection: alive across more than 8 steps) that is too large for the reserved zone would sit in the
    // general area for its whole life and push every large short-lived tensor of the levels below it into the arena (segm_full: the 72 KB level-3 skip keeps the
    // 76 KB depthwise outputs of all three level-4 blocks in HBM).  It is written once and read twice: it goes to the arena instead, and the short-lived tensors,
    // which are written and read back-to-back by latency-bound ops, get the LDS.  build_frame_program keeps whichever policy moves fewer arena bytes.
    if (long_to_hbm && lds_ok && reserved_at.find(t) == reserved_at.end() && firstdef[t] >= 0 && last[t] - firstdef[t] > 8 && P > 1 &&
        std::find(ext.begin(), ext.end(), t) == ext.end()) lds_ok = false;
    // Policy `elide_expand`: the output of a 1x1 that only the depthwise of the NEXT step reads need not exist at all — with its "home" in the arena the generator
    // computes it chunk by chunk straight into the depthwise's LDS workspace (gen_mid.cpp: mid_pw_feeds_dw).  Taken where the expanded tensor and the depthwise's
    // output cannot both be LDS-resident: first-fit would give the LDS to the expanded tensor (it comes first) and send the depthwise output — read twice more, by
    // the squeeze-excite pool and the project 1x1 — to the arena (segm_full's three 9x16x128 blocks).
    if (lds_ok && s >= 0 && s < NS && elide[s] && steps[s].out == t) lds_ok = false;
    if (lds_ok) {
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < s; }), live.end());
      std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      int pos = scratch;
      // Long-lived tensors (skip connections: alive across more than 8 steps) own a RESERVED zone at the top of the block for the whole program
      // (`reserved`, computed below; everything else allocates below it): a skip tensor dropped by first-fit into the middle of the block splits
      // the free space for its whole lifetime (segm_lite: the level-3 skip `C` sat at 62 KB and the 96 KB expanded tensor two steps later went to
      // HBM although 110 KB were free).
      auto rz = reserved_at.find(t);
      if (rz != reserved_at.end()) pos = rz->second;
      else if (small_top && need <= kLdsTotal / 6) {
        // Policy `small_top`: small tensors (block inputs / residuals: <= 1/6 of the block) allocate from the TOP like the weight slots, so that they do not end up
        // in the middle of the block — above whatever large tensor was alive when they were placed — and split the space the next large tensor needs
        // (segm_full: the 20 KB block input at 57 KB kept the 76 KB depthwise output out of a block with 80 KB free)
        pos = cap + 1;
        int hi = cap;
        for (int k = (int)live.size() - 1; k >= -1; k--) {
          const int lo = k >= 0 ? live[k].off + live[k].len : scratch;
          if (hi - lo >= need) { pos = hi - need; break; }
          if (k >= 0) hi = std::min(hi, live[k].off);
        }
      } else
      for (const Blk& b : live) { if (pos + need <= b.off) break; pos = std::max(pos, b.off + b.len); }
      if (rz != reserved_at.end() || pos + need <= cap) {
        l.space = kLocLds; l.off = pos; l.stride = stride;
        if (rz == reserved_at.end()) live.push_back({pos, need, last[t]});
        plan->program_blocks.push_back({pos, need, s, last[t], "tensor " + std::to_string(t)});
        high = std::max(high, pos + need);
        plan->program_lds_tensors++;
        loc[t] = l;
        return;
      }
    }
    l.space = t == g.input ? kLocInput : (t == g.output ? kLocOutput : kLocGlobal);
    l.off = (int)plan->tensor_off[t]; l.stride = C;
    plan->program_global_tensors++;
    loc[t] = l;
  };
  plan->program_lds_tensors = plan->program_global_tensors = 0;
  place(g.input, -1);
  for (int t : ext) {
    if (t < 0 || loc[t].space != kLocNone) continue;
    const TensorInfo& ti = g.tensors[t];
    Loc l;
    l.elems = (int)ti.elems(); l.space = kLocGlobal; l.off = (int)plan->tensor_off[t]; l.stride = ti.dims[3];
    loc[t] = l;
    plan->program_global_tensors++;
  }
  // Weight staging slots.  stage[s] floats of step s are DMA'd to LDS while the PREVIOUS micro-op runs, so the slot must be
  // free from the first step of that previous micro-op (q) to s.  q is conservative: a GAP → FC.. chain may fuse into one op.
  auto gemv_form = [&](const Step& st) { return st.kind == StepKind::PwConv && st.OH * st.OW <= 4 && st.OH * st.OW * st.Cout * 16 <= kLdsScratchFloats; };   // (a lowering with reduced scratch never contains one: see build_frame_program)
  std::vector<int> stage(NS, 0), slot(NS, 0);
  std::vector<std::vector<int>> slots_from(NS);
  for (int s = 0; s < NS; s++) {
    const Step& st = steps[s];
    long nb = st.kind == StepKind::DwConv ? st.Cout : (st.kind == StepKind::TConv ? st.Cout : st.cout_pad);
    long range = ((long)st.b_off - (long)st.w_off) + ((nb + 3) / 4) * 4;
    // FC layers of a GAP → FC [→ FC] chain (fused into one squeeze-excite micro-op below): [bias | [co][ci] weights] is one contiguous
    // range of the weight arena; it is staged while the op BEFORE the pool runs, so that the FCs read LDS instead of waiting for L2/HBM
    if (st.kind == StepKind::PwConv && st.OH * st.OW == 1 && st.w2_off > st.b_off && !BSX_DBG_ENV("BSX_NO_FC_STAGE")) {
      auto fc1px = [&](int k) { const Step& f = steps[k]; return f.kind == StepKind::PwConv && f.OH * f.OW == 1; };
      int q0 = s;
      while (q0 > 0 && fc1px(q0 - 1)) q0--;
      const long frange = ((long)st.w2_off - (long)st.b_off) + (long)st.Cout * st.Cin;
      if (q0 > 0 && s - q0 <= 1 && steps[q0 - 1].kind == StepKind::Gap && frange <= kLdsMaxStageFloats) {
        stage[s] = (int)((frange + 3) / 4 * 4);
        slots_from[std::max(q0 - 2, 0)].push_back(s);
      }
      continue;
    }
    bool uses = (st.kind == StepKind::PwConv && !gemv_form(st)) || st.kind == StepKind::Conv || st.kind == StepKind::DwConv || st.kind == StepKind::TConv;
    if (st.kind == StepKind::PwConv && st.cout_pad % 16 != 0) uses = false;     // only the matrix-core form stages
    if (!(uses && st.b_off > st.w_off && range <= kLdsMaxStageFloats)) continue;
    stage[s] = (int)range;
    int q = s > 0 ? s - 1 : 0;
    auto fc_like = [&](int k) { const Step& f = steps[k]; return f.kind == StepKind::PwConv && f.OH * f.OW == 1; };
    if (s > 0 && fc_like(q)) {
      while (q > 0 && fc_like(q - 1)) q--;
      if (q > 0 && steps[q - 1].kind == StepKind::Gap) q--;
    }
    if (s > 2 && steps[s - 1].kind == StepKind::TConv && steps[s - 2].kind == StepKind::DwConv) q = std::min(q, s - 3);
    slots_from[q].push_back(s);
  }
  std::vector<MicroOp> prog;
  std::vector<std::string> labels;
  std::vector<int> tail_ws(NS, -1), tail_rows(NS, 0);
  auto tail_pattern = [&](int s) {
    if (s + 2 >= NS || BSX_DBG_ENV("BSX_PROGRAM_NO_TAIL")) return false;
    const Step& a = steps[s];
    const Step& b = steps[s + 1];
    const Step& c2 = steps[s + 2];
    return a.kind == StepKind::PwConv && a.residual < 0 && a.Cin == 16 && a.cout_pad == 16 && a.Cout == 16 &&
           a.OH * a.OW > 1024 && last[a.out] == s + 1 &&
           b.kind == StepKind::DwConv && b.in0 == a.out && b.residual == a.out && b.kh == 3 && b.kw == 3 && b.sh == 1 && b.sw == 1 && b.dh == 1 &&
           b.dw == 1 && b.pad_t == 1 && b.pad_l == 1 && last[b.out] == s + 2 &&
           c2.kind == StepKind::TConv && c2.in0 == b.out && c2.kh == 2 && c2.kw == 2 && c2.Cout <= 4;
  };
  for (int s = 0; s < NS; s++) {
    const Step& st = steps[s];
    for (int s2 : slots_from[s]) {
      const int need = (stage[s2] + 3) / 4 * 4;
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < s; }), live.end());
      std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      // weight slots live for two steps and are small: placed from the TOP of the block (like the skip tensors) they leave the bottom
      // contiguous for the large activation tensors (a 9.6 KB slot at 54 KB kept segm_lite's 96 KB expanded tensor out of LDS)
      int pos = -1, hi = cap;
      if (!BSX_DBG_ENV("BSX_PLAN_NO_TOPDOWN")) {
        for (int k = (int)live.size() - 1; k >= -1; k--) {
          const int lo = k >= 0 ? live[k].off + live[k].len : scratch;
          if (hi - lo >= need) { pos = hi - need; break; }
          if (k >= 0) hi = std::min(hi, live[k].off);
        }
        if (pos < 0) pos = cap;
      } else {
        pos = scratch;
        for (const Blk& b : live) { if (pos + need <= b.off) break; pos = std::max(pos, b.off + b.len); }
      }
      if (pos + need <= cap) {
        const int until = elide[s2] ? s2 + 1 : s2;          // an elided 1x1 runs inside the depthwise of the next step
        slot[s2] = pos; live.push_back({pos, need, until}); high = std::max(high, pos + need);
        plan->program_blocks.push_back({pos, need, s, until, "weights of step " + std::to_string(s2)});
      }
      else stage[s2] = 0;                       // no room: the op falls back to its unstaged form
    }
    if (tail_pattern(s)) {
      // z row band: (R+2) rows x W pixels x (C+4) floats, alive for the three fused steps
      // the largest band that fits the LDS that is free right now: every band recomputes its two halo rows, so taller bands
      // mean less redundant work and fewer barriers (the kernel keeps a band's 16-pixel tiles in flight 4 per wave: <= 64 tiles)
      auto need_for = [&](int r) { return (r + 2) * st.W * (st.Cout + 4); };
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < s; }), live.end());
      std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      const int rmax = BSX_DBG_ENV("BSX_TAIL_ROWS") ? atoi(BSX_DBG_ENV("BSX_TAIL_ROWS")) : 16;
      for (int R = rmax; R >= 1 && tail_ws[s] < 0; R = R > 4 ? R - 2 : R - 1) {
        const int need = need_for(R);
        if ((R + 2) * st.W > 64 * 16 * 4) continue;           // phase A: <= 4 tiles per wave and band
        int pos = scratch;
        for (const Blk& b : live) { if (pos + need <= b.off) break; pos = std::max(pos, b.off + b.len); }
        if (pos + need <= cap) {
          tail_ws[s] = pos; tail_rows[s] = R; live.push_back({pos, need, s + 2}); high = std::max(high, pos + need);
          plan->program_blocks.push_back({pos, need, s, s + 2, "tail band of step " + std::to_string(s)});
        }
      }
    }
    place(st.out, s);
    MicroOp m;
    m.kind = (int)st.kind;
    m.H = st.H; m.W = st.W; m.Cin = st.Cin; m.OH = st.OH; m.OW = st.OW; m.Cout = st.Cout;
    m.kh = st.kh; m.kw = st.kw; m.sh = st.sh; m.sw = st.sw; m.dh = st.dh; m.dw = st.dw; m.pt = st.pad_t; m.pl = st.pad_l;
    m.act = st.act; m.elt = st.elt; m.bcast1 = st.bcast1; m.align_corners = st.align_corners; m.half_pixel = st.half_pixel;
    m.cout_pad = st.cout_pad; m.cout_tile = st.cout_tile;
    m.w_off = (long long)st.w_off; m.b_off = (long long)st.b_off; m.w2_off = (long long)st.w2_off;
    m.gemv = gemv_form(st) ? 1 : 0;
    m.strip = (st.kind == StepKind::DwConv && !BSX_DBG_ENV("BSX_NO_DW_STRIP")) ? 1 : 0;
    auto L = [&](int t) { return t >= 0 ? loc[t] : Loc(); };
    m.in0 = L(st.in0); m.in1 = L(st.in1); m.in2 = L(st.in2); m.res = L(st.residual); m.scale = L(st.in_scale); m.out = L(st.out);
    if (st.concat_in.size() > 4) return;
    m.n_cat = (int)st.concat_in.size();
    m.gap_sum = st.gap_sum ? 1 : 0;
    for (int k = 0; k < m.n_cat; k++) {
      const int ct = st.concat_in[k];
      m.cat[k] = L(ct); m.cat_c[k] = st.concat_c[k];
      m.cat_hw[k] = g.tensors[ct].dims[1] * g.tensors[ct].dims[2];
      auto pn = part_n.find(ct);
      if (pn != part_n.end()) { m.cat_parts[k] = pn->second; m.cat_hw[k] = part_hw.at(ct); }
    }
    if (st.kind == StepKind::DwConv && !((st.kh == 3 && st.kw == 3) || (st.kh == 5 && st.kw == 5))) return;   // program has 3x3 / 5x5 bodies only
    // weights + bias are contiguous in the arena ([w][pad to 4][b]); the whole range is staged in the slot planned above
    m.stage_floats = stage[s]; m.w_lds = slot[s];
    if (st.kind == StepKind::PwConv && st.OH * st.OW == 1 && st.w2_off > st.b_off) {     // FC: the staged range is [bias | w2], not [w | bias]
      m.fc_stage[0] = stage[s]; m.fc_lds[0] = slot[s]; m.stage_floats = 0; m.w_lds = 0;
    }
    if (st.kind == StepKind::PwConv && !m.gemv) {
      // MFMA form whenever the weight block + bias got an LDS slot and Cout tiles by 16
      m.cout_tile = 16;
      m.mfma = (st.cout_pad % 16 == 0 && m.stage_floats > 0) ? 1 : 0;
      if (!m.mfma) m.stage_floats = 0;   // the SGPR-fed VALU body reads its weights from memory
    }
    if (st.kind == StepKind::Conv && m.stage_floats > 0 && st.cout_pad % 16 == 0 && m.in0.space != kLocLds && st.residual < 0 &&
        st.dh == 1 && st.dw == 1 && st.kh * st.kw * st.Cin <= 256) {
      // matrix-core stem: find an LDS workspace for a band of input rows (+ the per-k table) among the free blocks at this step
      const int rowf = st.W * st.Cin;
      int band = 16;
      auto need_for = [&](int b) { return ((b - 1) * st.sh + st.kh) * rowf + 3 * (((st.kh * st.kw * st.Cin + 3) / 4) * 4) + 8; };
      auto band_floats = [&](int b) { return ((b - 1) * st.sh + st.kh) * rowf; };
      while (band > 1 && band_floats(band) > 8192) band /= 2;   // the band is double-buffered through 2 float4 registers per lane
      const int need = (need_for(band) + 3) / 4 * 4;
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < s; }), live.end());
      std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      int pos = scratch;
      for (const Blk& b : live) { if (pos + need <= b.off) break; pos = std::max(pos, b.off + b.len); }
      if (pos + need <= cap && band_floats(band) <= 8192) {
        m.mfma = 1; m.ws_off = pos; m.band_rows = band;
        live.push_back({pos, need, s});          // occupied for this step only
        plan->program_blocks.push_back({pos, need, s, s, "stem band of step " + std::to_string(s)});
        high = std::max(high, pos + need);
      }
    }
    if (st.kind == StepKind::DwConv && m.strip && m.in0.space == kLocGlobal && st.dh == 1 && st.dw == 1 && st.residual < 0 && !BSX_DBG_ENV("BSX_PLAN_NO_DW_STAGE")) {
      // Depthwise on a tensor that lives in the arena: every input element is needed by K output rows and ~2 strips, i.e. it is read ~10 times — from an L2 that
      // 32 frames' tensors share.  Where LDS has room the specialised kernel walks the channels in chunks of CK: the chunk of the WHOLE input ([H*W][CK + 4]) is
      // brought into an LDS workspace once, coalesced, and the taps run from there (gen_mid.cpp).  ws_off / band_rows (= CK) carry the reservation; the
      // interpreter ignores them and keeps reading the arena.
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.until < s; }), live.end());
      std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      for (int CK : {32, 16}) {                      // (8-channel chunks leave most lanes without an item and cost two barriers each: the arena form is faster)
        if (m.band_rows || (st.Cin % CK) % 8) continue;        // whole chunks, or a ragged last one of whole 8-channel pieces (88 = 5 x 16 + 8: round 4)
        const int need = (st.H * st.W * (CK + 4) + 3) / 4 * 4;
        int pos = scratch;
        for (const Blk& b : live) { if (pos + need <= b.off) break; pos = std::max(pos, b.off + b.len); }
        if (pos + need <= cap) {
          m.ws_off = pos; m.band_rows = CK;
          live.push_back({pos, need, s});          // occupied for this step only
          plan->program_blocks.push_back({pos, need, s, s, "depthwise input chunk of step " + std::to_string(s)});
          high = std::max(high, pos + need);
        }
      }
    }
    if (m.scale.space != kLocNone && m.scale.space != kLocLds) return;   // the pw micro-op reads SE scales with ds_read only
    if (BSX_DBG_ENV("BSX_PROGRAM_NOP")) m.kind = 99;
    if (const char* only = BSX_DBG_ENV("BSX_PROGRAM_ONLY")) { if (atoi(only) != s) m.kind = 99; }   // timing experiments: one live op
    prog.push_back(m);
    {
      char buf[200];
      snprintf(buf, sizeof buf, "%-12s %dx%dx%d->%dx%dx%d%s%s in:%s out:%s", st.label.c_str(), st.H, st.W, st.Cin, st.OH, st.OW, st.Cout,
               m.mfma ? " mfma" : "", m.gemv ? " gemv" : "", m.in0.space == kLocLds ? "lds" : "hbm", m.out.space == kLocLds ? "lds" : "hbm");
      labels.push_back(buf);
    }
  }
  // peephole: GAP → FC(act) [→ FC(act)] on single-pixel vectors → one fused micro-op (means/hidden stay in their LDS slots)
  if (!BSX_DBG_ENV("BSX_PROGRAM_NO_SE")) {
    std::vector<MicroOp> fusedp;
    std::vector<std::string> flabels;
    for (size_t i = 0; i < prog.size(); i++) {
      const MicroOp& g0 = prog[i];
      auto is_fc = [&](size_t k, const Loc& in) {
        if (k >= prog.size()) return false;
        const MicroOp& f = prog[k];
        const Step& st = steps[k];
        return f.kind == (int)StepKind::PwConv && f.OH * f.OW == 1 && st.w2_off != 0 && f.res.space == kLocNone && f.scale.space == kLocNone &&
               f.in0.space == kLocLds && f.in0.off == in.off && (f.out.space == kLocLds || f.out.space == kLocGlobal) && f.Cin % 4 == 0;
      };
      auto single_use = [&](size_t k) { const Step& st = steps[k]; return last[st.out] == (int)k + 1; };
      if (tail_ws[i] >= 0 && prog[i].mfma && prog[i].out.space == kLocGlobal && prog[i].scale.space != kLocGlobal) {
        // pw(+muladd) → dw 3x3 (+act, + z) → tconv 2x2: z never leaves LDS, t never leaves registers
        MicroOp m = prog[i];
        const MicroOp& d = prog[i + 1];
        const MicroOp& t = prog[i + 2];
        m.kind = kMicroTail; m.mfma = 0; m.stage_floats = 0;
        m.w3_off = d.w_off; m.b3_off = d.b_off; m.act2 = d.act;
        m.w4_off = t.w_off; m.b4_off = t.b_off; m.act3 = t.act; m.C2 = t.Cout;
        m.OH = t.OH; m.OW = t.OW; m.out = t.out;
        m.ws_off = tail_ws[i]; m.band_rows = tail_rows[i];
        m.magic_w = (unsigned)((0x100000000ull + (unsigned long long)m.W - 1) / (unsigned long long)m.W);
        fusedp.push_back(m);
        flabels.push_back("tail[pw+dw+tconv] " + labels[i]);
        i += 2;
        continue;
      }
      if (g0.kind == (int)StepKind::Gap && g0.out.space == kLocLds && is_fc(i + 1, g0.out) && single_use(i)) {
        MicroOp m = g0;
        const MicroOp& f1 = prog[i + 1];
        m.kind = kMicroSe; m.in1 = g0.out; m.in2 = f1.out; m.n_fc = 1;
        m.w2_off = (long long)steps[i + 1].w2_off; m.b_off = f1.b_off; m.act = f1.act; m.C1 = f1.Cout; m.out = f1.out;
        m.fc_stage[0] = f1.fc_stage[0]; m.fc_lds[0] = f1.fc_lds[0]; m.fc_stage[1] = 0;
        size_t used = 2;
        if (is_fc(i + 2, f1.out) && single_use(i + 1)) {
          const MicroOp& f2 = prog[i + 2];
          m.n_fc = 2; m.w3_off = (long long)steps[i + 2].w2_off; m.b3_off = f2.b_off; m.act2 = f2.act; m.C2 = f2.Cout; m.out = f2.out;
          m.fc_stage[1] = f2.fc_stage[0]; m.fc_lds[1] = f2.fc_lds[0];
          used = 3;
        }
        fusedp.push_back(m);
        flabels.push_back("se[" + std::to_string(m.n_fc) + "fc" + (m.fc_stage[0] ? "+w1" : "") + (m.fc_stage[1] ? "+w2" : "") + "] " + labels[i]);
        i += used - 1;
      } else {
        fusedp.push_back(g0);
        flabels.push_back(labels[i]);
      }
    }
    prog.swap(fusedp);
    labels.swap(flabels);
  }
  plan->program = std::move(prog);
  plan->program_labels = std::move(labels);
  plan->program_lds_floats = high;
  if (BSX_DBG_ENV("BSX_PLAN_DEBUG"))
    for (const auto& b : plan->program_blocks) fprintf(stderr, "lds block [%6d, %6d) steps [%2d, %2d] %s\n", b.off, b.off + b.len, b.from, b.until, b.what.c_str());
  plan->program_check = verify_program_lds(*plan);
  if (plan->program_check != "ok") plan->program.clear();      // never run a program whose LDS reservations collide
}

// The reduction scratch at the bottom of the LDS block (frame_program.hpp: kLdsScratchFloats, 8 KB) is used by the single-pixel GEMV form and by the
// fused decoder tail only.  A lowering that contains neither (every FC folded into a squeeze-excite op: the middle of a segmented plan) is
// repeated with 256 bytes of scratch — segm_lite's 6x10x96 tensor missed LDS by 1.2 KB.
// bytes of arena (HBM / L2) operands one frame's program touches: every use of a tensor that is not in LDS, read or written
static long program_arena_bytes(const Plan& plan) {
  long b = 0;
  const int n = (int)plan.program.size();
  for (int i = 0; i < n; i++) {
    const MicroOp& m = plan.program[i];
    const bool out_elided = mid_pw_feeds_dw(plan, i), in_elided = i > 0 && mid_pw_feeds_dw(plan, i - 1);     // the tensor between a fused 1x1 → depthwise pair never exists
    // a fused 1x1 runs once per channel chunk of the depthwise behind it: its arena operands are read that many times
    const long reps = out_elided ? (plan.program[i + 1].Cin + plan.program[i + 1].band_rows - 1) / plan.program[i + 1].band_rows : 1;
    for (const Loc* l : {&m.in0, &m.in1, &m.in2, &m.res, &m.out}) {
      if (l->space != kLocGlobal || (l == &m.out && out_elided) || (l == &m.in0 && in_elided)) continue;
      b += 4L * l->elems * (l == &m.out ? 1 : reps);
    }
    for (int c = 0; c < m.n_cat; c++) if (m.cat[c].space == kLocGlobal && m.cat_parts[c] == 0) b += 4L * m.cat[c].elems;
  }
  return b;
}
static void build_frame_program(const Graph& g, Plan* plan, const std::vector<Step>& steps, const std::vector<int>& ext = {},
                                const std::map<int, int>& part_n = {}, const std::map<int, int>& part_hw = {}) {
  auto lower = [&](unsigned policy) {
    lower_frame_program(g, plan, steps, ext, part_n, part_hw, kLdsScratchFloats, policy);
    if (plan->program.empty() || BSX_DBG_ENV("BSX_PLAN_FULL_SCRATCH")) return;
    for (const MicroOp& m : plan->program)
      if ((m.kind == (int)StepKind::PwConv && m.gemv) || m.kind == kMicroTail || m.kind == (int)StepKind::Conv) return;
    lower_frame_program(g, plan, steps, ext, part_n, part_hw, 64, policy);
    if (plan->program.empty()) lower_frame_program(g, plan, steps, ext, part_n, part_hw, kLdsScratchFloats, policy);
  };
  // The placement is greedy (program order, first fit); which tensors SHOULD lose the LDS is a policy question with three independent answers (lower_frame_program:
  // bit 0 long-lived tensors to the arena, bit 1 expanded tensors elided, bit 2 small tensors from the top).  All eight combinations are lowered — microseconds on the
  // host — and the one whose program touches the fewest arena bytes per frame is kept; ties keep the lowest policy number (0 = the round-3 planner).
  // BSX_PLAN_POLICY=<0..7> forces one (A/B timing).
  if (const char* force = BSX_DBG_ENV("BSX_PLAN_POLICY")) { plan->program_policy = (unsigned)atoi(force) & 7u; lower(plan->program_policy); plan->program_arena_bytes = program_arena_bytes(*plan); return; }
  unsigned best = 0;
  long best_b = -1;
  for (unsigned pol = 0; pol < 8; pol++) {
    lower(pol);
    if (plan->program.empty()) continue;
    const long b = program_arena_bytes(*plan);
    if (best_b < 0 || b < best_b) { best_b = b; best = pol; }
  }
  lower(best);
  plan->program_policy = best;
  plan->program_arena_bytes = program_arena_bytes(*plan);
}

// Independent check of the lowering: no two LDS reservations that are alive at the same step may share a float, every block
// stays inside [scratch, Plan::lds_zero_off()) — below the zero cell —, and every LDS operand of every micro-op lies inside the block area.
std::string verify_program_lds(const Plan& plan) {
  const auto& b = plan.program_blocks;
  for (size_t i = 0; i < b.size(); i++) {
    if (b[i].off < plan.program_scratch_floats || b[i].off + b[i].len > plan.lds_zero_off() || b[i].len <= 0 || b[i].from > b[i].until)      // (the zero cell above is nobody's: ADVICE r5)
      return "block out of range: " + b[i].what;
    for (size_t j = i + 1; j < b.size(); j++) {
      const bool time = b[i].from <= b[j].until && b[j].from <= b[i].until;
      const bool mem = b[i].off < b[j].off + b[j].len && b[j].off < b[i].off + b[i].len;
      if (time && mem) return "overlap: " + b[i].what + " [" + std::to_string(b[i].from) + "," + std::to_string(b[i].until) + "] and " + b[j].what + " [" +
                              std::to_string(b[j].from) + "," + std::to_string(b[j].until) + "]";
    }
  }
  for (const MicroOp& m : plan.program) {
    for (const Loc* l : {&m.in0, &m.in1, &m.in2, &m.res, &m.scale, &m.out})
      if (l->space == kLocLds && (l->off < plan.program_scratch_floats || l->off >= plan.program_lds_floats)) return "operand outside the LDS block area";
    if (m.stage_floats > 0 && (m.w_lds < plan.program_scratch_floats || m.w_lds + m.stage_floats > plan.program_lds_floats)) return "weight slot outside the LDS block area";
  }
  return "ok";
}


// ---- segmentation of the Meet / MLKit family (segments.hpp) ------------------------------------------------------------------
namespace {

// total interpolation weight every source row/column receives from RESIZE_BILINEAR in -> out (same clamping as the kernels);
// GAP(resize(x)) == GAP(x) iff the weights are uniform
bool resize_weights_uniform(int in, int out, bool align, bool half_pixel) {
  std::vector<double> wsum(in, 0.0);
  const float scale = (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out;
  for (int o = 0; o < out; o++) {
    const float v = half_pixel ? ((float)o + 0.5f) * scale - 0.5f : (float)o * scale;
    const float fl = std::floor(v);
    const int lo = std::max((int)fl, 0), hi = std::min((int)std::ceil(v), in - 1);
    const float frac = v - (float)lo;
    wsum[lo] += 1.0 - frac; wsum[hi] += frac;
  }
  const double want = (double)out / (double)in;
  for (double x : wsum) if (std::fabs(x - want) > 1e-5) return false;
  return true;
}

SegConvW conv_w(const Step& st) { SegConvW c; c.w_off = (long long)st.w_off; c.b_off = (long long)st.b_off; c.Cin = st.kh * st.kw * st.Cin; c.Cout = st.Cout; c.cout_pad = st.cout_pad; c.act = st.act; return c; }
SegDwW dw_w(const Step& st) { SegDwW c; c.w_off = (long long)st.w_off; c.b_off = (long long)st.b_off; c.C = st.Cout; c.act = st.act; return c; }
SegFc fc_w(const Step& st) { SegFc c; c.w_off = (long long)st.w2_off; c.b_off = (long long)st.b_off; c.Cin = st.Cin; c.Cout = st.Cout; c.act = st.act; return c; }

bool is_pw16(const Step& st) { return st.kind == StepKind::PwConv && st.Cin == 16 && st.Cout == 16 && st.cout_pad == 16 && st.OH * st.OW > 4; }
bool is_fc_step(const Step& st) { return st.kind == StepKind::PwConv && st.OH * st.OW == 1 && st.w2_off != 0 && st.residual < 0 && st.in_scale < 0 && st.in2 < 0 && st.Cin <= 32 && st.Cin % 8 == 0 && st.Cout <= 32; }
bool is_dw3(const Step& st, int stride) {
  return st.kind == StepKind::DwConv && st.kh == 3 && st.kw == 3 && st.sh == stride && st.sw == stride && st.dh == 1 && st.dw == 1;
}
int tiles_for(int extent, int target) { return (extent + target - 1) / target; }

}  // namespace

// Recognise  head | k2 | middle | k3 | tail  in the fused step list and lower it: four segment descriptors + the per-frame
// program for the middle.  Returns false (plan untouched apart from scratch) when the graph does not have this shape.
static bool seg_fail(int where) { if (BSX_DBG_ENV("BSX_SEG_DEBUG")) fprintf(stderr, "segmentation: check %d failed\n", where); return false; }
static bool build_segments(Graph& g, Plan* plan) {
  const std::vector<Step>& S = plan->steps;
  const int NS = (int)S.size();
  if (NS < 9 + 15) return seg_fail(1);
  // operand slots that read tensor t (a pooling step lists its parts in concat_in; its in0 merely repeats the first part)
  auto uses_of = [&](int t) {
    int n = 0;
    for (const Step& q : S) {
      for (int u : {q.concat_in.empty() ? q.in0 : -1, q.in1, q.in2, q.residual, q.in_scale}) n += (u == t);
      for (int u : q.concat_in) n += (u == t);
    }
    return n + (t == g.output);
  };
  auto dims = [&](int t, int k) { return g.tensors[t].dims[k]; };

  // ---- head: stem → 1x1 → dw/s2
  const Step &stem = S[0], &hpw = S[1], &hdw = S[2];
  if (!(stem.kind == StepKind::Conv && stem.in0 == g.input && stem.kh == 3 && stem.kw == 3 && stem.sh == 2 && stem.sw == 2 && stem.dh == 1 && stem.dw == 1 &&
        stem.Cin == 3 && stem.Cout == 16 && stem.cout_pad == 16 && stem.residual < 0)) return seg_fail(2);
  const int A = stem.out;
  if (!(is_pw16(hpw) && hpw.in0 == A && hpw.residual < 0 && hpw.in_scale < 0 && hpw.in2 < 0 && uses_of(hpw.out) == 1)) return seg_fail(3);
  if (!(is_dw3(hdw, 2) && hdw.in0 == hpw.out && hdw.Cout == 16 && hdw.residual < 0)) return seg_fail(4);
  const int b0 = hdw.out;
  // ---- k2: GAP(b0) → FC → FC → 1x1 (scaled) → 1x1 expand → dw/s2
  const Step &g1 = S[3], &f1a = S[4], &f1b = S[5], &pwa = S[6], &pwb = S[7], &kdw = S[8];
  if (!(g1.kind == StepKind::Gap && g1.concat_in.empty() && g1.in0 == b0 && is_fc_step(f1a) && f1a.in0 == g1.out && is_fc_step(f1b) && f1b.in0 == f1a.out &&
        f1a.Cin == 16 && f1b.Cout == 16 && uses_of(b0) == 2)) return seg_fail(5);
  if (!(is_pw16(pwa) && pwa.in0 == b0 && pwa.in_scale == f1b.out && pwa.residual < 0 && pwa.in2 < 0 && uses_of(f1b.out) == 1)) return seg_fail(6);
  const int B = pwa.out;
  if (!(pwb.kind == StepKind::PwConv && pwb.in0 == B && pwb.Cin == 16 && pwb.cout_pad % 16 == 0 && pwb.residual < 0 && pwb.in_scale < 0 && pwb.in2 < 0 &&
        uses_of(pwb.out) == 1 && pwb.Cout % 4 == 0)) return seg_fail(7);
  if (!(is_dw3(kdw, 2) && kdw.in0 == pwb.out && kdw.residual < 0)) return seg_fail(8);
  const int c0 = kdw.out;
  // ---- tail: resize → GAP(A, up) → FC → FC → 1x1 (A*g + up) → dw + residual → tconv
  const Step &tr = S[NS - 7], &tg = S[NS - 6], &tf1 = S[NS - 5], &tf2 = S[NS - 4], &tpw = S[NS - 3], &tdw = S[NS - 2], &ttc = S[NS - 1];
  if (!(tr.kind == StepKind::Resize && tr.Cin == 16 && tr.OH == dims(A, 1) && tr.OW == dims(A, 2))) return seg_fail(9);
  const int lo = tr.in0, up = tr.out;
  auto pools = [](const Step& q, int a, int b) { return q.kind == StepKind::Gap && q.concat_in.size() == 2 && ((q.concat_in[0] == a && q.concat_in[1] == b) || (q.gap_sum && q.concat_in[0] == b && q.concat_in[1] == a)); };
  if (!(pools(tg, A, up) && is_fc_step(tf1) && tf1.in0 == tg.out &&
        is_fc_step(tf2) && tf2.in0 == tf1.out && tf2.Cout == 16)) return seg_fail(10);
  if (!(is_pw16(tpw) && tpw.in0 == A && tpw.in_scale == tf2.out && tpw.in2 == up && tpw.residual < 0)) return seg_fail(11);
  if (!(is_dw3(tdw, 1) && tdw.in0 == tpw.out && tdw.residual == tpw.out && tdw.Cout == 16 && tdw.pad_t == 1 && tdw.pad_l == 1)) return seg_fail(12);
  if (!(ttc.kind == StepKind::TConv && ttc.in0 == tdw.out && ttc.kh == 2 && ttc.kw == 2 && ttc.Cin == 16 && (ttc.Cout == 1 || ttc.Cout == 2) && ttc.out == g.output)) return seg_fail(13);
  if (uses_of(A) != 3 || uses_of(up) != 2 || uses_of(tpw.out) != 2 || uses_of(tdw.out) != 1 || uses_of(tf2.out) != 1) return seg_fail(14);
  if (!resize_weights_uniform(tr.H, tr.OH, tr.align_corners, tr.half_pixel) || !resize_weights_uniform(tr.W, tr.OW, tr.align_corners, tr.half_pixel)) return seg_fail(15);
  // ---- k3: resize → GAP(B, up2) → FC → FC → 1x1 (B*g + up2) → dw + residual → 1x1 → lo
  const Step &kr = S[NS - 14], &kg = S[NS - 13], &kf1 = S[NS - 12], &kf2 = S[NS - 11], &kp1 = S[NS - 10], &kd = S[NS - 9], &kp2 = S[NS - 8];
  if (!(kr.kind == StepKind::Resize && kr.Cin == 16 && kr.OH == dims(B, 1) && kr.OW == dims(B, 2))) return seg_fail(16);
  const int lo2 = kr.in0, up2 = kr.out;
  if (!(pools(kg, B, up2) && is_fc_step(kf1) && kf1.in0 == kg.out &&
        is_fc_step(kf2) && kf2.in0 == kf1.out && kf2.Cout == 16)) return seg_fail(17);
  if (!(is_pw16(kp1) && kp1.in0 == B && kp1.in_scale == kf2.out && kp1.in2 == up2 && kp1.residual < 0)) return seg_fail(18);
  if (!(is_dw3(kd, 1) && kd.in0 == kp1.out && kd.residual == kp1.out && kd.Cout == 16 && kd.pad_t == 1 && kd.pad_l == 1)) return seg_fail(19);
  if (!(is_pw16(kp2) && kp2.in0 == kd.out && kp2.out == lo && kp2.residual < 0 && kp2.in_scale < 0 && kp2.in2 < 0)) return seg_fail(20);
  if (uses_of(B) != 3 || uses_of(up2) != 2 || uses_of(kp1.out) != 2 || uses_of(kd.out) != 1 || uses_of(kf2.out) != 1 || uses_of(lo) != 1 || uses_of(lo2) != 1) return seg_fail(21);
  if (!resize_weights_uniform(kr.H, kr.OH, kr.align_corners, kr.half_pixel) || !resize_weights_uniform(kr.W, kr.OW, kr.align_corners, kr.half_pixel)) return seg_fail(22);
  if (uses_of(c0) != 1) return seg_fail(23);
  // activations the kernels implement: a clamp (none / relu / relu6) everywhere, hard-swish also on the stem; the transpose convolution: none or the logistic
  auto clampish = [](int a) { return a == kActNone || a == kActRelu || a == kActRelu6; };
  if (!(clampish(stem.act) || stem.act == kActHswish) || !(ttc.act == kActNone || ttc.act == kActSigmoid) || (ttc.act == kActSigmoid && ttc.Cout != 1)) return seg_fail(34);
  for (const Step* q : {&hpw, &hdw, &pwa, &pwb, &kdw, &kp1, &kd, &kp2, &tpw, &tdw}) if (!clampish(q->act)) return seg_fail(35);

  // ---- dedicated, never-reused arena space for everything that crosses a kernel boundary (a segment kernel reads and writes
  //      different tiles of its tensors concurrently, so liveness-based sharing inside one kernel would be a race)
  size_t top = plan->arena_floats_per_stream;
  auto reserve = [&](size_t floats) { size_t at = top; top += (floats + 63) / 64 * 64; return (long long)at; };
  auto synth = [&](const char* name, int n, int C) {
    TensorInfo t; t.dims[0] = 1; t.dims[1] = n; t.dims[2] = 1; t.dims[3] = C; t.shape = {1, n, 1, C}; t.name = name;
    g.tensors.push_back(t);
    plan->tensor_off.push_back(-1);
    return (int)g.tensors.size() - 1;
  };
  SegPlan sp;
  // tile geometry (BSX_SEG_TILES="hTR,hTC,k2TR,k2TC,k3TR,k3TC,tTR,tTC" overrides the targets)
  // kernel limits: head TR <= 4, TC <= 15 (one depthwise row per wave, 2TC+1 <= 32 columns of 6 floats per input row lane chunk);
  // k2 TC <= 15; k3 / tail TC <= 14 (TC + 2 <= 16: one MFMA tile per halo-region row)
  int tgt[8] = {4, 14, 4, 7, 16, 14, 18, 14};       // (tail rows 16 -> 18, round 4: segm_full's 72 rows split into 4 x 18 instead of 5 x 15 — no padded rows, one halo pair fewer:
                                                   //  tail 466 -> 414 us at 1024 HD streams, profiles/r04q; 48 and 128 rows still split into 16s)
  if (const char* e = BSX_DBG_ENV("BSX_SEG_TILES")) sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &tgt[0], &tgt[1], &tgt[2], &tgt[3], &tgt[4], &tgt[5], &tgt[6], &tgt[7]);
  auto split = [&](int extent, int target, int* tile, int* n) { *n = tiles_for(extent, std::max(1, target)); *tile = (extent + *n - 1) / *n; };

  SegHead& h = sp.head;
  h.H0 = stem.H; h.W0 = stem.W; h.H1 = stem.OH; h.W1 = stem.OW; h.H2 = hdw.OH; h.W2 = hdw.OW;
  h.stem_pt = stem.pad_t; h.stem_pl = stem.pad_l; h.dw_pt = hdw.pad_t; h.dw_pl = hdw.pad_l;
  h.stem = conv_w(stem); h.pw = conv_w(hpw); h.dw = dw_w(hdw);
  split(h.H2, tgt[0], &h.TR, &h.tiles_y); split(h.W2, tgt[1], &h.TC, &h.tiles_x);
  h.lds_floats = seg_head_lds_floats(h);
  h.rw = seg_row_width(2 * h.TC + 1); h.m_ct = (65536u + (unsigned)(h.rw / 16) - 1) / (unsigned)(h.rw / 16);
  if (h.TR > 4 || h.TC > 15 || (4 * h.TC + 3) * 3 > 192) return seg_fail(30);
  SegK2& k2 = sp.k2;
  k2.H2 = hdw.OH; k2.W2 = hdw.OW; k2.H3 = kdw.OH; k2.W3 = kdw.OW; k2.dw_pt = kdw.pad_t; k2.dw_pl = kdw.pad_l;
  k2.pw_a = conv_w(pwa); k2.pw_b = conv_w(pwb); k2.dw = dw_w(kdw);
  split(k2.H3, tgt[2], &k2.TR, &k2.tiles_y); split(k2.W3, tgt[3], &k2.TC, &k2.tiles_x);
  k2.lds_floats = seg_k2_lds_floats(k2);
  k2.rw = seg_row_width(2 * k2.TC + 1); k2.m_ct = (65536u + (unsigned)(k2.rw / 16) - 1) / (unsigned)(k2.rw / 16);
  if (k2.TC > 15 || ((2 * k2.TR + 1) * (k2.rw / 16) + 3) / 4 > 6) return seg_fail(31);
  SegK3& k3 = sp.k3;
  k3.H2 = kp1.OH; k3.W2 = kp1.OW; k3.HL = kr.H; k3.WL = kr.W; k3.half_pixel = kr.half_pixel; k3.align_corners = kr.align_corners;
  k3.pw1 = conv_w(kp1); k3.pw2 = conv_w(kp2); k3.dw = dw_w(kd);
  split(k3.H2, tgt[4], &k3.TR, &k3.tiles_y); split(k3.W2, tgt[5], &k3.TC, &k3.tiles_x);
  k3.hs = seg_up_scale(k3.HL, k3.H2, k3.align_corners != 0); k3.ws = seg_up_scale(k3.WL, k3.W2, k3.align_corners != 0);
  if (!BSX_DBG_ENV("BSX_SEG_LO_WORST")) k3.lo_floats = seg_lo_window_floats(k3.H2, k3.W2, k3.HL, k3.WL, k3.half_pixel != 0, k3.align_corners != 0, k3.TR, k3.TC, k3.tiles_y, k3.tiles_x);
  k3.lds_floats = seg_k3_lds_floats(k3);
  if (k3.TC > 14 || k3.TR > 18) return seg_fail(32);
  SegTail& tl = sp.tail;
  tl.H1 = tpw.OH; tl.W1 = tpw.OW; tl.HL = tr.H; tl.WL = tr.W; tl.H0 = ttc.OH; tl.W0 = ttc.OW; tl.half_pixel = tr.half_pixel; tl.align_corners = tr.align_corners;
  tl.pw = conv_w(tpw); tl.dw = dw_w(tdw); tl.tc_w_off = (long long)ttc.w_off; tl.tc_b_off = (long long)ttc.b_off; tl.Co = ttc.Cout; tl.act3 = ttc.act;
  split(tl.H1, tgt[6], &tl.TR, &tl.tiles_y); split(tl.W1, tgt[7], &tl.TC, &tl.tiles_x);
  tl.hs = seg_up_scale(tl.HL, tl.H1, tl.align_corners != 0); tl.ws = seg_up_scale(tl.WL, tl.W1, tl.align_corners != 0);
  if (!BSX_DBG_ENV("BSX_SEG_LO_WORST")) tl.lo_floats = seg_lo_window_floats(tl.H1, tl.W1, tl.HL, tl.WL, tl.half_pixel != 0, tl.align_corners != 0, tl.TR, tl.TC, tl.tiles_y, tl.tiles_x);
  tl.lds_floats = seg_tail_lds_floats(tl);
  if (tl.TC > 14 || tl.TR > 18) return seg_fail(33);
  const int lds_cap = 160 * 256;
  if (h.lds_floats > lds_cap || k2.lds_floats > lds_cap || k3.lds_floats > lds_cap || tl.lds_floats > lds_cap) return seg_fail(24);
  if (ttc.OH != 2 * tpw.OH || ttc.OW != 2 * tpw.OW || h.H2 != k3.H2 || h.W2 != k3.W2) return seg_fail(25);
  if (h.tiles_y * h.tiles_x > 1024 || k2.tiles_y * k2.tiles_x > 1024 || k3.tiles_y * k3.tiles_x > 1024) return seg_fail(26);

  const int pA = synth("partials(A)", h.tiles_y * h.tiles_x, 16), pb0 = synth("partials(b0)", h.tiles_y * h.tiles_x, 16);
  const int pB = synth("partials(B)", k2.tiles_y * k2.tiles_x, 16), plo = synth("partials(lo)", k3.tiles_y * k3.tiles_x, 16);
  // The tail's gate (two pooled means → FC → FC: identical for every tile of a frame) is computed once per frame by a one-workgroup-per-frame launch between k3
  // and the tail instead of by each of the tail's workgroups: the prologue — partial sums of two tensors, two weight blocks, three barriers — measured a quarter of
  // the tail kernel (lite 67.7 -> 50.8 us, MLKit/HD 260 -> 192 us with the prologue skipped).  BSX_SEG_NO_GATE_KERNEL=1 keeps it inside the tail.
  const int pgt = BSX_DBG_ENV("BSX_SEG_NO_GATE_KERNEL") ? -1 : synth("gate(tail)", 1, 16);
  if (pgt >= 0) plan->tensor_off[pgt] = reserve(16);
  for (int t : {A, b0, B, c0, lo2, lo, kf2.out, pA, pb0, pB, plo}) plan->tensor_off[t] = reserve(g.tensors[t].elems());
  // tensors that exist only inside a segment kernel are never materialised
  for (int t : {hpw.out, g1.out, f1a.out, f1b.out, pwb.out, up2, kg.out, kf1.out, kp1.out, kd.out, up, tg.out, tf1.out, tf2.out, tpw.out, tdw.out})
    if (t != g.output) plan->tensor_off[t] = -1;
  plan->arena_floats_per_stream = top;

  h.a_off = plan->tensor_off[A]; h.b0_off = plan->tensor_off[b0]; h.part_a_off = plan->tensor_off[pA]; h.part_b0_off = plan->tensor_off[pb0];
  k2.b0_off = plan->tensor_off[b0]; k2.B_off = plan->tensor_off[B]; k2.c0_off = plan->tensor_off[c0]; k2.part_B_off = plan->tensor_off[pB];
  k2.gate.n_parts = 1; k2.gate.n_fc = 2; k2.gate.sum_parts = 0;
  k2.gate.part[0].off = plan->tensor_off[pb0]; k2.gate.part[0].n = h.tiles_y * h.tiles_x; k2.gate.part[0].C = 16; k2.gate.part[0].hw = (float)(h.H2 * h.W2);
  k2.gate.fc[0] = fc_w(f1a); k2.gate.fc[1] = fc_w(f1b);
  k3.skip_off = plan->tensor_off[B]; k3.lo2_off = plan->tensor_off[lo2]; k3.g_off = plan->tensor_off[kf2.out]; k3.lo_off = plan->tensor_off[lo];
  k3.part_lo_off = plan->tensor_off[plo];
  tl.skip_off = plan->tensor_off[A]; tl.lo_off = plan->tensor_off[lo];
  tl.gate.n_parts = 2; tl.gate.n_fc = 2; tl.gate.sum_parts = tg.gap_sum ? 1 : 0;
  tl.gate.part[0].off = plan->tensor_off[pA]; tl.gate.part[0].n = h.tiles_y * h.tiles_x; tl.gate.part[0].C = 16; tl.gate.part[0].hw = (float)(h.H1 * h.W1);
  tl.gate.part[1].off = plan->tensor_off[plo]; tl.gate.part[1].n = k3.tiles_y * k3.tiles_x; tl.gate.part[1].C = 16; tl.gate.part[1].hw = (float)(k3.H2 * k3.W2);
  tl.gate.fc[0] = fc_w(tf1); tl.gate.fc[1] = fc_w(tf2);
  if (BSX_DBG_ENV("BSX_SEG_GATE_SKIP")) k2.gate.timing_skip = tl.gate.timing_skip = 1;
  if (const char* e = BSX_DBG_ENV("BSX_SEG_SKIP")) sscanf(e, "%d,%d,%d,%d", &h.dbg_skip, &k2.dbg_skip, &k3.dbg_skip, &tl.dbg_skip);      // "head,k2,k3,tail" phase masks
  tl.pre_gate_off = pgt >= 0 ? plan->tensor_off[pgt] : -1;
  if (tl.gate.fc[0].Cin != (tl.gate.sum_parts ? 16 : 32) || k2.gate.fc[0].Cin != 16) return seg_fail(27);

  // ---- the middle: steps 9 .. NS-15, then the level-2 gate with its pooled inputs replaced: GAP(B) arrives as partial sums
  //      from k2, GAP(up2) == GAP(lo2) (uniform 2x interpolation weights)
  std::vector<Step> mid(S.begin() + 9, S.begin() + (NS - 14));
  Step gate = kg;
  gate.concat_in = {B, lo2};
  gate.in0 = B;
  gate.label += "*";
  mid.push_back(gate); mid.push_back(kf1); mid.push_back(kf2);
  std::map<int, int> part_n, part_hw;
  part_n[B] = k2.tiles_y * k2.tiles_x; part_hw[B] = k2.H2 * k2.W2;
  // the program sees the partial sums of B under B's tensor id: point that id at the partials for the duration of the lowering
  const long B_real = plan->tensor_off[B];
  plan->tensor_off[B] = plan->tensor_off[pB];
  build_frame_program(g, plan, mid, {c0, B, lo2, kf2.out}, part_n, part_hw);
  plan->tensor_off[B] = B_real;
  if (plan->program.empty()) return seg_fail(28);
  sp.on = true;
  plan->seg = sp;
#ifdef BSX_DEBUG_SWITCHES
  // BSX_SEG_DUMP=<file> (debug build): the four segment descriptors of this plan as C++ constants — the input of tools/seg_probe.sh, the ahead-of-time experiment that
  // preceded the hipRTC-specialised segment kernels (gen_seg.cpp emits the same text for them)
  if (const char* path = BSX_DBG_ENV("BSX_SEG_DUMP")) {
    if (FILE* f = fopen(path, "w")) { fputs(seg_constants_text(sp, "kProbe").c_str(), f); fclose(f); }
  }
#endif
  char line[256];
  plan->seg_text.clear();
  auto add = [&](const char* name, int TR, int TC, int ty, int tx, int lds) {
    snprintf(line, sizeof line, "segment %-5s tile %dx%d, %dx%d tiles per frame, LDS %.1f KiB\n", name, TR, TC, ty, tx, lds / 256.0);
    plan->seg_text += line;
  };
  add("head", h.TR, h.TC, h.tiles_y, h.tiles_x, h.lds_floats); add("k2", k2.TR, k2.TC, k2.tiles_y, k2.tiles_x, k2.lds_floats);
  add("k3", k3.TR, k3.TC, k3.tiles_y, k3.tiles_x, k3.lds_floats); add("tail", tl.TR, tl.TC, tl.tiles_y, tl.tiles_x, tl.lds_floats);
  return true;
}


MidGeometry mid_geometry_default() {
  static const MidGeometry geo = [] {
    MidGeometry m{kFrameThreads, kLdsTotalFloats};
    if (const char* e = BSX_DBG_ENV("BSX_MID_LANES")) { const int v = atoi(e); if (v == 512 || v == 1024) m.lanes = v; }
    if (const char* e = BSX_DBG_ENV("BSX_MID_LDS_KB")) { const int v = atoi(e); if (v >= 64 && v <= 160) m.lds_floats = v * 256; }
    return m;
  }();
  return geo;
}

bool build_plan(const Graph& g_in, Plan* plan, std::string* err, bool reuse_arena, bool segments) {
  auto fail = [&](const std::string& m) { if (err) *err = m; return false; };
  { const MidGeometry geo = mid_geometry_default(); plan->mid_lanes = geo.lanes; plan->lds_total_floats = geo.lds_floats; }
  Graph g = g_in;                       // local copy: the rewrite passes below may append synthetic tensors
  int NT = (int)g.tensors.size();
  const int NN = (int)g.nodes.size();

  // consumers of each tensor
  std::vector<std::vector<int>> users(NT);
  for (int i = 0; i < NN; i++) for (int t : g.nodes[i].inputs) if (t >= 0 && !g.tensors[t].is_const) users[t].push_back(i);
  auto single_user = [&](int t) -> int { return (t != g.output && users[t].size() == 1) ? users[t][0] : -1; };

  std::vector<char> fused(NN, 0);
  std::vector<Step> steps;
  std::vector<float>& W = plan->weights;
  W.clear();
  auto align_w = [&]() { while (W.size() % 4) W.push_back(0.f); };

  for (int i = 0; i < NN; i++) {
    if (fused[i]) continue;
    const Node& n = g.nodes[i];
    const TensorInfo& out_t = g.tensors[n.output];
    Step st;
    st.last_node = n.index;
    auto T = [&](int k) -> const TensorInfo& { return g.tensors[n.inputs[k]]; };

    auto chain_epilogue = [&](int cur) -> int {
      // fold [unary act] then [residual ADD] that solely consume the running output
      for (;;) {
        int u = single_user(cur);
        if (u < 0 || fused[u]) break;
        const Node& y = g.nodes[u];
        if (is_unary(y.type) && st.act == kActNone && st.residual < 0) {
          st.act = unary_act(y.type); fused[u] = 1; cur = y.output; st.last_node = y.index; continue;
        }
        if (y.type == OpType::Add && y.act == kActNone && st.residual < 0 && y.inputs.size() == 2) {
          int other = y.inputs[0] == cur ? y.inputs[1] : y.inputs[0];
          if (other >= 0 && other != cur && !g.tensors[other].is_const && same_dims(g.tensors[other], g.tensors[cur])) {
            st.residual = other; fused[u] = 1; cur = y.output; st.last_node = y.index; continue;
          }
        }
        break;
      }
      return cur;
    };

    switch (n.type) {
      case OpType::Conv:
      case OpType::FullyConnected: {
        if (n.inputs.size() < 2) return fail("conv without weights");
        const TensorInfo& x = T(0);
        const TensorInfo& w = T(1);
        if (!w.is_const) return fail("non-constant conv weights unsupported");
        const TensorInfo* b = (n.inputs.size() > 2 && n.inputs[2] >= 0) ? &T(2) : nullptr;
        if (b && !b->is_const) return fail("non-constant bias unsupported");
        bool fc = n.type == OpType::FullyConnected;
        st.in0 = n.inputs[0];
        st.H = x.dims[1]; st.W = x.dims[2]; st.Cin = x.dims[3];
        if (fc) {
          st.Cout = w.dims[2]; st.kh = st.kw = 1;
          if (w.dims[3] != st.Cin) return fail("FULLY_CONNECTED depth mismatch");
          st.OH = st.H; st.OW = st.W;
        } else {
          st.Cout = w.dims[0]; st.kh = w.dims[1]; st.kw = w.dims[2];
          if (w.dims[3] != st.Cin) return fail("CONV_2D depth mismatch");
          st.sh = n.stride_h; st.sw = n.stride_w; st.dh = n.dil_h; st.dw = n.dil_w;
          conv_geometry(st.H, st.kh, st.sh, st.dh, n.same_padding, &st.OH, &st.pad_t);
          conv_geometry(st.W, st.kw, st.sw, st.dw, n.same_padding, &st.OW, &st.pad_l);
        }
        if (st.OH != out_t.dims[1] || st.OW != out_t.dims[2] || st.Cout != out_t.dims[3]) return fail("conv output shape mismatch at op #" + std::to_string(n.index));
        bool pw = st.kh == 1 && st.kw == 1 && st.sh == 1 && st.sw == 1;
        st.kind = pw ? StepKind::PwConv : StepKind::Conv;
        st.act = n.act;
        // squeeze-excite fold: input = MUL(x, s[N,1,1,C]) consumed only here → scale on load
        if (pw) {
          int prod = -1;
          for (int j = 0; j < i; j++) if (!fused[j] && g.nodes[j].output == st.in0) prod = j;
          if (prod >= 0 && g.nodes[prod].type == OpType::Mul && g.nodes[prod].act == kActNone && single_user(st.in0) == i) {
            const Node& m = g.nodes[prod];
            int a = m.inputs[0], s2 = m.inputs[1];
            if (a >= 0 && s2 >= 0 && !g.tensors[a].is_const && !g.tensors[s2].is_const) {
              if (is_chan_vec(g.tensors[a], g.tensors[s2])) std::swap(a, s2);
              if (is_chan_vec(g.tensors[s2], g.tensors[a])) {
                // the MUL step was already emitted (it precedes us); remove it
                for (size_t k = 0; k < steps.size(); k++)
                  if (steps[k].kind == StepKind::Eltwise && steps[k].out == st.in0) { steps.erase(steps.begin() + k); break; }
                st.in0 = a; st.in_scale = s2;
              }
            }
          }
        }
        // pack weights [kh][kw][ci][co_pad]
        int ct = st.Cout <= 16 ? 16 : 32;
        if (!pw) ct = 16;
        st.cout_tile = ct;
        st.cout_pad = round_up(st.Cout, ct);
        align_w();
        st.w_off = W.size();
        W.resize(W.size() + (size_t)st.kh * st.kw * st.Cin * st.cout_pad, 0.f);
        for (int o = 0; o < st.Cout; o++) for (int fy = 0; fy < st.kh; fy++) for (int fx = 0; fx < st.kw; fx++) for (int c = 0; c < st.Cin; c++) {
          size_t src = fc ? ((size_t)o * st.Cin + c) : ((((size_t)o * st.kh + fy) * st.kw + fx) * st.Cin + c);
          W[st.w_off + (((size_t)fy * st.kw + fx) * st.Cin + c) * st.cout_pad + o] = w.f32[src];
        }
        align_w();
        st.b_off = W.size();
        W.resize(W.size() + st.cout_pad, 0.f);
        if (b) for (int o = 0; o < st.Cout; o++) W[st.b_off + o] = b->f32[o];
        if (pw && st.OH * st.OW <= 4 && st.Cin % 4 == 0) {  // GEMV form: [co][ci]
          align_w();
          st.w2_off = W.size();
          W.resize(W.size() + (size_t)st.Cout * st.Cin);
          for (int o = 0; o < st.Cout; o++) for (int c = 0; c < st.Cin; c++)
            W[st.w2_off + (size_t)o * st.Cin + c] = w.f32[fc ? ((size_t)o * st.Cin + c) : ((size_t)o * st.Cin + c)];
        }
        st.macs = (double)st.OH * st.OW * st.Cout * st.kh * st.kw * st.Cin;
        st.out = chain_epilogue(n.output);
        st.label = (fc ? "fc#" : "conv#") + std::to_string(n.index);
        break;
      }
      case OpType::DwConv: {
        const TensorInfo& x = T(0);
        const TensorInfo& w = T(1);
        if (!w.is_const) return fail("non-constant depthwise weights unsupported");
        if (n.depth_mult != 1) return fail("depth_multiplier != 1 unsupported");
        const TensorInfo* b = (n.inputs.size() > 2 && n.inputs[2] >= 0) ? &T(2) : nullptr;
        st.kind = StepKind::DwConv;
        st.in0 = n.inputs[0];
        st.H = x.dims[1]; st.W = x.dims[2]; st.Cin = st.Cout = x.dims[3];
        st.kh = w.dims[1]; st.kw = w.dims[2];
        if (w.dims[3] != st.Cin) return fail("depthwise channel mismatch");
        if (st.Cin % 4) return fail("depthwise channels must be a multiple of 4");
        st.sh = n.stride_h; st.sw = n.stride_w; st.dh = n.dil_h; st.dw = n.dil_w;
        conv_geometry(st.H, st.kh, st.sh, st.dh, n.same_padding, &st.OH, &st.pad_t);
        conv_geometry(st.W, st.kw, st.sw, st.dw, n.same_padding, &st.OW, &st.pad_l);
        if (st.OH != out_t.dims[1] || st.OW != out_t.dims[2]) return fail("depthwise output shape mismatch at op #" + std::to_string(n.index));
        st.act = n.act;
        st.cout_pad = st.Cout;
        align_w();
        st.w_off = W.size();
        W.insert(W.end(), w.f32.begin(), w.f32.end());  // already [kh][kw][C]
        align_w();
        st.b_off = W.size();
        W.resize(W.size() + st.Cout, 0.f);
        if (b) for (int o = 0; o < st.Cout; o++) W[st.b_off + o] = b->f32[o];
        st.macs = (double)st.OH * st.OW * st.Cout * st.kh * st.kw;
        st.out = chain_epilogue(n.output);
        st.label = "dw#" + std::to_string(n.index);
        break;
      }
      case OpType::TransposeConvBias: {
        const TensorInfo& x = T(0);
        const TensorInfo& w = T(1);
        const TensorInfo& b = T(2);
        if (!w.is_const || !b.is_const) return fail("non-constant transpose-conv weights unsupported");
        st.kind = StepKind::TConv;
        st.in0 = n.inputs[0];
        st.H = x.dims[1]; st.W = x.dims[2]; st.Cin = x.dims[3];
        st.Cout = w.dims[0]; st.kh = w.dims[1]; st.kw = w.dims[2];
        st.sh = n.tconv_stride_h; st.sw = n.tconv_stride_w;
        // lib/transpose_conv_bias.cc:171-181: SAME pad = max(0, k-(in-1)%s-1); out = s*(in-1)+k-pad
        int pad_h = n.tconv_padding_same ? std::max(0, st.kh - (st.H - 1) % st.sh - 1) : 0;
        int pad_w = n.tconv_padding_same ? std::max(0, st.kw - (st.W - 1) % st.sw - 1) : 0;
        st.OH = st.sh * (st.H - 1) + st.kh - pad_h; st.OW = st.sw * (st.W - 1) + st.kw - pad_w;
        st.pad_t = pad_h / 2; st.pad_l = pad_w / 2;
        if (st.kh != st.sh || st.kw != st.sw || st.pad_t || st.pad_l)
          return fail("Convolution2DTransposeBias: only kernel==stride, zero-pad geometry is implemented");
        if (st.OH != out_t.dims[1] || st.OW != out_t.dims[2] || st.Cout != out_t.dims[3]) return fail("tconv output shape mismatch");
        if (st.Cin % 4) return fail("tconv input channels must be a multiple of 4");
        align_w();
        st.w_off = W.size();  // [fy][fx][oc][ic]
        W.resize(W.size() + (size_t)st.kh * st.kw * st.Cout * st.Cin);
        for (int o = 0; o < st.Cout; o++) for (int fy = 0; fy < st.kh; fy++) for (int fx = 0; fx < st.kw; fx++) for (int c = 0; c < st.Cin; c++)
          W[st.w_off + (((size_t)fy * st.kw + fx) * st.Cout + o) * st.Cin + c] = w.f32[(((size_t)o * st.kh + fy) * st.kw + fx) * st.Cin + c];
        align_w();
        st.b_off = W.size();
        W.insert(W.end(), b.f32.begin(), b.f32.begin() + st.Cout);
        st.macs = (double)st.OH * st.OW * st.Cout * st.Cin;
        st.out = chain_epilogue(n.output);
        if (st.residual >= 0) return fail("residual after tconv unsupported");
        st.label = "tconv#" + std::to_string(n.index);
        break;
      }
      case OpType::AvgPool: {
        const TensorInfo& x = T(0);
        if (n.filter_h != x.dims[1] || n.filter_w != x.dims[2] || out_t.dims[1] != 1 || out_t.dims[2] != 1)
          return fail("only global AVERAGE_POOL_2D is implemented (op #" + std::to_string(n.index) + ")");
        if (n.act != kActNone) return fail("fused activation on AVERAGE_POOL_2D unsupported");
        st.kind = StepKind::Gap;
        st.in0 = n.inputs[0]; st.out = n.output;
        st.H = x.dims[1]; st.W = x.dims[2]; st.Cin = st.Cout = x.dims[3];
        st.label = "gap#" + std::to_string(n.index);
        if (st.Cin % 4) return fail("global average pool channels must be a multiple of 4");
        // GAP(concat(a, b)) == concat(GAP(a), GAP(b)): pool the parts directly, never build the concat
        if (single_user(st.in0) == i) {
          for (size_t k = 0; k < steps.size(); k++)
            if (steps[k].kind == StepKind::Concat && steps[k].out == st.in0) {
              st.concat_in = steps[k].concat_in; st.concat_c = steps[k].concat_c; st.in0 = st.concat_in[0];
              steps.erase(steps.begin() + k);
              st.label = "gapcat#" + std::to_string(n.index);
              break;
            }
          // GAP(a + b) == GAP(a) + GAP(b): when the sum feeds nothing but the pool it is never materialised
          // (MLKit's decoder gates pool skip+up; the full-resolution sum cost a 3 MB/frame HBM round trip)
          if (st.concat_in.empty())
            for (size_t k = 0; k < steps.size(); k++) {
              const Step& e = steps[k];
              if (e.kind == StepKind::Eltwise && e.out == st.in0 && e.elt == kEltAdd && !e.bcast1 && e.act == kActNone && e.in1 >= 0 &&
                  !BSX_DBG_ENV("BSX_NO_GAP_SUM")) {
                st.concat_in = {e.in0, e.in1}; st.concat_c = {st.Cin, st.Cin}; st.gap_sum = true; st.in0 = e.in0;
                steps.erase(steps.begin() + k);
                st.label = "gapsum#" + std::to_string(n.index);
                break;
              }
            }
        }
        break;
      }
      case OpType::Relu: case OpType::Relu6: case OpType::HardSwish: case OpType::Logistic: {
        const TensorInfo& x = T(0);
        st.kind = StepKind::Eltwise; st.elt = kEltUnary; st.act = unary_act(n.type);
        st.in0 = n.inputs[0]; st.out = n.output;
        st.H = st.OH = x.dims[1]; st.W = st.OW = x.dims[2]; st.Cin = st.Cout = x.dims[3];
        st.label = "act#" + std::to_string(n.index);
        break;
      }
      case OpType::Add: case OpType::Mul: {
        if (n.inputs.size() != 2) return fail("binary op arity");
        int a = n.inputs[0], b = n.inputs[1];
        if (g.tensors[a].is_const || g.tensors[b].is_const) return fail("constant operand in ADD/MUL unsupported");
        if (is_chan_vec(g.tensors[a], g.tensors[b])) std::swap(a, b);
        const TensorInfo& x = g.tensors[a];
        st.kind = StepKind::Eltwise; st.elt = n.type == OpType::Add ? kEltAdd : kEltMul;
        st.in0 = a; st.in1 = b; st.out = n.output; st.act = n.act;
        st.bcast1 = is_chan_vec(g.tensors[b], x);
        if (!st.bcast1 && !same_dims(g.tensors[b], x)) return fail("unsupported broadcast in ADD/MUL at op #" + std::to_string(n.index));
        st.H = st.OH = x.dims[1]; st.W = st.OW = x.dims[2]; st.Cin = st.Cout = x.dims[3];
        st.label = std::string(n.type == OpType::Add ? "add#" : "mul#") + std::to_string(n.index);
        // gate*skip + up  →  one pass
        if (n.type == OpType::Mul && st.bcast1 && n.act == kActNone) {
          int u = single_user(n.output);
          if (u >= 0 && !fused[u] && g.nodes[u].type == OpType::Add && g.nodes[u].act == kActNone) {
            const Node& y = g.nodes[u];
            int other = y.inputs[0] == n.output ? y.inputs[1] : y.inputs[0];
            if (other >= 0 && !g.tensors[other].is_const && same_dims(g.tensors[other], x)) {
              st.elt = kEltMulAdd; st.in2 = other; st.out = y.output; fused[u] = 1; st.last_node = y.index;
              st.label = "muladd#" + std::to_string(n.index);
            }
          }
        }
        break;
      }
      case OpType::ResizeBilinear: {
        const TensorInfo& x = T(0);
        st.kind = StepKind::Resize;
        st.in0 = n.inputs[0]; st.out = n.output;
        st.H = x.dims[1]; st.W = x.dims[2]; st.Cin = st.Cout = x.dims[3];
        st.OH = out_t.dims[1]; st.OW = out_t.dims[2];
        if (n.inputs.size() > 1 && n.inputs[1] >= 0) {
          const TensorInfo& sz = T(1);
          if (sz.i32.size() >= 2 && (sz.i32[0] != st.OH || sz.i32[1] != st.OW)) return fail("RESIZE_BILINEAR size disagrees with output shape");
        }
        st.align_corners = n.align_corners; st.half_pixel = n.half_pixel;
        st.label = "resize#" + std::to_string(n.index);
        break;
      }
      case OpType::Concat: {
        int ax = n.axis < 0 ? n.axis + 4 : n.axis + (4 - (int)out_t.shape.size());
        if (ax != 3) return fail("only channel-axis CONCATENATION is implemented");
        st.kind = StepKind::Concat;
        st.out = n.output;
        st.H = st.OH = out_t.dims[1]; st.W = st.OW = out_t.dims[2]; st.Cout = out_t.dims[3];
        for (int t : n.inputs) { if (g.tensors[t].is_const) return fail("constant CONCATENATION input unsupported"); if (g.tensors[t].dims[3] % 4) return fail("concat channels must be multiples of 4"); st.concat_in.push_back(t); st.concat_c.push_back(g.tensors[t].dims[3]); }
        st.in0 = st.concat_in[0];
        st.label = "concat#" + std::to_string(n.index);
        break;
      }
      default:
        return fail("operator #" + std::to_string(n.index) + " has no GPU implementation");
    }
    steps.push_back(std::move(st));
  }
  // a fused group executes where its LAST op stood, so every external input already exists
  std::stable_sort(steps.begin(), steps.end(), [](const Step& a, const Step& b) { return a.last_node < b.last_node; });

  // ---- linear-algebra rewrites on the step list ------------------------------------------------------------------
  // operand slots reading tensor t (steps that list their parts in concat_in repeat the first part in in0: counted once)
  auto uses_of = [&](int t) { int n = 0; for (const Step& q : steps) { for (int u : {q.concat_in.empty() ? q.in0 : -1, q.in1, q.in2, q.residual, q.in_scale, q.out_bias}) n += (u == t); for (int u : q.concat_in) n += (u == t); } return n + (t == g.output); };
  const bool no_rewrites = BSX_DBG_ENV("BSX_NO_REWRITES") != nullptr;
  // (a) pw(resize(x)) → resize(pw(x)): a 1x1 convolution without activation commutes with bilinear interpolation (both
  //     are linear and the interpolation weights sum to 1, so the bias passes through); done at the LOW resolution the
  //     convolution costs 1/4 of the MACs and the up-sampled many-channel tensor is never materialised.  The result differs
  //     from the reference order only by f32 rounding (covered by the 1e-4 logit tolerance).
  for (size_t i = 0; i < steps.size() && !no_rewrites; i++) {
    if (steps[i].kind != StepKind::Resize || uses_of(steps[i].out) != 1) continue;
    size_t j = i + 1;
    for (; j < steps.size(); j++) if (steps[j].in0 == steps[i].out) break;
    if (j >= steps.size()) continue;
    Step& R = steps[i];
    Step& Pw = steps[j];
    if (Pw.kind != StepKind::PwConv || Pw.act != kActNone || Pw.residual >= 0 || Pw.in_scale >= 0 || Pw.in2 >= 0) continue;
    if (R.OH * R.OW <= R.H * R.W) continue;   // only worth it when up-sampling
    TensorInfo nt;
    nt.dims[0] = 1; nt.dims[1] = R.H; nt.dims[2] = R.W; nt.dims[3] = Pw.Cout;
    nt.shape = {1, R.H, R.W, Pw.Cout};
    nt.name = "lowres_pw";
    g.tensors.push_back(nt);
    const int tnew = (int)g.tensors.size() - 1;
    Step pw2 = Pw, r2 = R;
    pw2.in0 = R.in0; pw2.H = pw2.OH = R.H; pw2.W = pw2.OW = R.W; pw2.out = tnew;
    pw2.macs = (double)R.H * R.W * Pw.Cout * Pw.Cin;
    pw2.label = Pw.label + "@lo";
    r2.in0 = tnew; r2.Cin = r2.Cout = Pw.Cout; r2.out = Pw.out;
    r2.label = R.label + "'";
    steps.erase(steps.begin() + j);
    steps[i] = pw2;
    steps.insert(steps.begin() + i + 1, r2);
  }
  NT = (int)g.tensors.size();
  // (b) y = a*s + b feeding only a 1x1 convolution: formed on the fly while the convolution loads its input
  //     (same two roundings as the separate MUL and ADD ops, so bit-identical to the unfused form)
  for (size_t i = 0; i < steps.size() && !no_rewrites; i++) {
    if (steps[i].kind != StepKind::Eltwise || steps[i].elt != kEltMulAdd || !steps[i].bcast1 || steps[i].act != kActNone) continue;
    if (uses_of(steps[i].out) != 1) continue;
    size_t j = i + 1;
    for (; j < steps.size(); j++) if (steps[j].in0 == steps[i].out) break;
    if (j >= steps.size()) continue;
    Step& Pw = steps[j];
    if (Pw.kind != StepKind::PwConv || Pw.in_scale >= 0 || Pw.in2 >= 0 || Pw.Cin % 4) continue;
    Pw.in0 = steps[i].in0; Pw.in_scale = steps[i].in1; Pw.in2 = steps[i].in2;   // in2 of a conv step = tensor added to the scaled input
    Pw.label += "+muladd";
    steps.erase(steps.begin() + i);
    i--;
  }
  // (c) y = a * s (s a per-channel vector) feeding only a 1x1 convolution — the Meet head's pw(x) * sigmoid(pw(GAP(x))) in front of the
  //     first decoder convolution, which rewrite (a) has just moved next to it: the same fold as a squeeze-excite MUL
  for (size_t i = 0; i < steps.size() && !no_rewrites; i++) {
    if (steps[i].kind != StepKind::Eltwise || steps[i].elt != kEltMul || !steps[i].bcast1 || steps[i].act != kActNone) continue;
    if (uses_of(steps[i].out) != 1) continue;
    size_t j = i + 1;
    for (; j < steps.size(); j++) if (steps[j].in0 == steps[i].out) break;
    if (j >= steps.size()) continue;
    Step& Pw = steps[j];
    if (Pw.kind != StepKind::PwConv || Pw.in_scale >= 0 || Pw.in2 >= 0 || Pw.Cin % 4 || Pw.OH * Pw.OW <= 4) continue;
    Pw.in0 = steps[i].in0; Pw.in_scale = steps[i].in1;
    Pw.label += "+mul";
    steps.erase(steps.begin() + i);
    i--;
  }
  // (d) a global average pool runs right before its first consumer (steps are in file order; the Meet head has an independent
  //     1x1 convolution between the pool and the FC that reads it): adjacent, the pool and its FC chain fuse into one micro-op
  for (size_t i = 0; i + 2 < steps.size() && !no_rewrites; i++) {
    if (steps[i].kind != StepKind::Gap) continue;
    const int t = steps[i].out;
    size_t j = i + 1;
    auto reads = [&](const Step& q) { for (int u : {q.in0, q.in1, q.in2, q.residual, q.in_scale}) if (u == t) return true; for (int u : q.concat_in) if (u == t) return true; return false; };
    while (j < steps.size() && !reads(steps[j])) j++;
    if (j >= steps.size() || j == i + 1) continue;
    std::rotate(steps.begin() + i, steps.begin() + i + 1, steps.begin() + j);   // the pool moves to position j-1
  }

  // (e) RESIZE_BILINEAR to the same size is the identity (DeepLab's 33 → 33 resize in front of the final 33 → 257 one): its
  //     consumers read its input directly
  for (size_t i = 0; i < steps.size() && !no_rewrites; i++) {
    const Step R = steps[i];
    if (R.kind != StepKind::Resize || R.H != R.OH || R.W != R.OW || R.out == g.output) continue;
    for (Step& q : steps) {
      for (int* u : {&q.in0, &q.in1, &q.in2, &q.residual, &q.in_scale}) if (*u == R.out) *u = R.in0;
      for (int& u : q.concat_in) if (u == R.out) u = R.in0;
    }
    steps.erase(steps.begin() + i);
    i--;
  }
  // (f) concat(broadcast(p), X) feeding only a 1x1 convolution, p a per-frame [1,1,Ca] vector up-sampled to X's size (DeepLab's ASPP
  //     image-pooling branch): the convolution is linear, so  conv(concat(bcast p, X)) = W[0:Ca]·p + W[Ca:]·X + b.  The first term is
  //     one GEMV per frame (a per-frame bias vector); the 512-channel concat and the broadcast tensor are never built and the
  //     convolution's K halves.  Weight rows are [ci][cout_pad], so both halves are views of the packed block: no repacking.
  for (size_t i = 0; i + 1 < steps.size() && !no_rewrites; i++) {
    if (steps[i].kind != StepKind::Concat || steps[i].concat_in.size() != 2 || uses_of(steps[i].out) != 1) continue;
    size_t j = i + 1;
    for (; j < steps.size(); j++) if (steps[j].in0 == steps[i].out) break;
    if (j >= steps.size()) continue;
    if (steps[j].kind != StepKind::PwConv || steps[j].in_scale >= 0 || steps[j].in2 >= 0 || steps[j].residual >= 0 || steps[j].OH * steps[j].OW <= 4) continue;
    const int tb = steps[i].concat_in[0], tx = steps[i].concat_in[1];
    size_t r = steps.size();
    for (size_t k = 0; k < i; k++) if (steps[k].kind == StepKind::Resize && steps[k].out == tb && steps[k].H == 1 && steps[k].W == 1) r = k;
    if (r >= steps.size() || uses_of(tb) != 1) continue;
    const int p = steps[r].in0, Ca = steps[i].concat_c[0], Cb = steps[i].concat_c[1];
    if (Ca % 4 || Cb % 4) continue;
    Step conv = steps[j];
    TensorInfo nt;
    nt.dims[0] = 1; nt.dims[1] = 1; nt.dims[2] = 1; nt.dims[3] = conv.Cout; nt.shape = {1, 1, 1, conv.Cout}; nt.name = "frame_bias";
    g.tensors.push_back(nt);
    const int tbias = (int)g.tensors.size() - 1;
    Step gemv = conv;                                  // W[0:Ca]·p + b, no activation
    gemv.in0 = p; gemv.H = gemv.W = gemv.OH = gemv.OW = 1; gemv.Cin = Ca; gemv.act = kActNone; gemv.out = tbias; gemv.w2_off = 0;
    gemv.macs = (double)Ca * conv.Cout; gemv.label = conv.label + "[pool branch]";
    conv.in0 = tx; conv.Cin = Cb; conv.w_off = conv.w_off + (size_t)Ca * conv.cout_pad; conv.out_bias = tbias;
    align_w();
    conv.b_off = W.size(); W.resize(W.size() + conv.cout_pad, 0.f);      // its bias moved into the per-frame vector
    conv.macs = (double)conv.OH * conv.OW * conv.Cout * Cb; conv.label += "-pool";
    // the resize and the concat disappear; the GEMV takes the concat's place (p exists by then)
    steps[j] = conv;
    steps[i] = gemv;
    steps.erase(steps.begin() + r);
    i--;
  }

  NT = (int)g.tensors.size();
  // ---- split-f16 copies of the large pointwise-conv weights: w = hi + lo (+ 2^-22 relative), [hi | lo][cout_pad][Kp], k contiguous
  plan->weights16.clear();
  for (Step& st : steps) {
    st.w16_off = 0; st.k16_pad = 0;
    if (st.kind != StepKind::PwConv || st.Cin < 8 || st.Cin % 4 != 0 || st.cout_pad % 16 != 0 || st.OH * st.OW <= 4) continue;
    const int Kp = round_up(st.Cin, 32);
    while (plan->weights16.size() % 8) plan->weights16.push_back(0);          // 16-byte aligned rows
    st.w16_off = plan->weights16.size(); st.k16_pad = Kp;
    plan->weights16.resize(plan->weights16.size() + 2 * (size_t)st.cout_pad * Kp, 0);
    uint16_t* hi = plan->weights16.data() + st.w16_off;
    uint16_t* lo = hi + (size_t)st.cout_pad * Kp;
    for (int k = 0; k < st.Cin; k++) for (int o = 0; o < st.Cout; o++) {
      const float wv = W[st.w_off + (size_t)k * st.cout_pad + o];
      const uint16_t h = f32_to_f16_rn(wv);
      hi[(size_t)o * Kp + k] = h;
      lo[(size_t)o * Kp + k] = f32_to_f16_rn(wv - f16_to_f32(h));
    }
  }

  // ---- chains of three 1x1 convolutions (DeepLab's ASPP head) for pw_chain3_k: a (32 S0 → 32 P1 channels) → b (→ 32 P2, may carry the folded pool branch as a
  // per-frame bias) → c (→ at most 32), clamp activations, each intermediate read by the next step only.  The stream is appended to weights16.
  for (size_t ia = 0; ia < steps.size() && !BSX_DBG_ENV("BSX_NO_CHAIN3"); ia++) {
    Step& a = steps[ia];
    auto plain_pw = [](const Step& q) { return q.kind == StepKind::PwConv && q.k16_pad > 0 && q.residual < 0 && q.in_scale < 0 && q.in2 < 0 && q.act < kActHswish; };
    if (!plain_pw(a) || a.out_bias >= 0 || a.Cin != 32 * kChainS0 || a.Cout != 32 * kChainP1 || uses_of(a.out) != 1 || a.chain_mid >= 0) continue;
    size_t ib = ia + 1, ic;
    for (; ib < steps.size(); ib++) if (steps[ib].in0 == a.out) break;
    if (ib >= steps.size()) continue;
    Step& b = steps[ib];
    if (!plain_pw(b) || b.Cout != 32 * kChainP2 || b.OH != a.OH || b.OW != a.OW || uses_of(b.out) != 1) continue;
    for (ic = ib + 1; ic < steps.size(); ic++) if (steps[ic].in0 == b.out) break;
    if (ic >= steps.size()) continue;
    Step& c = steps[ic];
    if (!plain_pw(c) || c.out_bias >= 0 || c.Cout > 32 || c.OH != a.OH || c.OW != a.OW) continue;
    while (plan->weights16.size() % 8) plan->weights16.push_back(0);
    b.chain_w16_off = plan->weights16.size();
    b.chain_first = (int)ia; b.chain_last = (int)ic; a.chain_mid = c.chain_mid = (int)ib;
    auto put_tile = [&](const Step& q, int tile, int slab, bool permuted, bool lo) {
      for (int lane = 0; lane < 64; lane++) for (int i = 0; i < 8; i++) {
        const int li = lane & 15, gq = lane >> 4, o = 16 * tile + li;
        const int k = 32 * slab + (permuted ? (i < 4 ? 4 * gq + i : 16 + 4 * gq + (i - 4)) : 8 * gq + i);
        uint16_t v = 0;
        if (o < q.Cout && k < q.Cin) {
          const float wv = W[q.w_off + (size_t)k * q.cout_pad + o];
          const uint16_t h = f32_to_f16_rn(wv);
          v = lo ? f32_to_f16_rn(wv - f16_to_f32(h)) : h;
        }
        plan->weights16.push_back(v);
      }
    };
    for (int s = 0; s < kChainS0; s++) for (int t = 0; t < 2 * kChainP1; t++) { put_tile(a, t, s, false, false); put_tile(a, t, s, false, true); }
    for (int p = 0; p < kChainP2; p++) {
      for (int s = 0; s < kChainP1; s++) for (int h = 0; h < 2; h++) { put_tile(b, 2 * p + h, s, true, false); put_tile(b, 2 * p + h, s, true, true); }
      for (int o = 0; o < 2; o++) { put_tile(c, o, p, true, false); put_tile(c, o, p, true, true); }
    }
  }

  // ---- activation arena: first-fit over [first def, last use] intervals, in per-stream float units
  plan->tensor_off.assign(NT, -1);
  std::vector<int> first(NT, -1), last(NT, -1);
  auto touch = [&](int t, int s) { if (t < 0) return; if (first[t] < 0) first[t] = s; last[t] = s; };
  const int NS = (int)steps.size();
  touch(g.input, -1);
  for (int s = 0; s < NS; s++) {
    const Step& st = steps[s];
    touch(st.in0, s); touch(st.in1, s); touch(st.in2, s); touch(st.residual, s); touch(st.in_scale, s); touch(st.out_bias, s);
    for (int t : st.concat_in) touch(t, s);
    touch(st.out, s);
  }
  // The per-launch path fuses neighbouring steps into one kernel AFTER this allocation (expand 1x1 + depthwise: ir_expand_dw_k; stem +
  // depthwise + 1x1: dl_head0_k): such a kernel writes the LAST step's output while other workgroups still read the FIRST step's input.
  // An output placed over a tensor whose last reader is the step before it (the expand's input in a block without a residual) is then a
  // race between workgroups — found by the full-batch twin test at 1024 DeepLab streams: 3 % of the streams wrong, every 2-8 stream test
  // green.  A convolution-type step's output therefore becomes live two steps early, i.e. from the first step any fusion can start at.
  for (int s = 0; s < NS; s++)
    if ((steps[s].kind == StepKind::DwConv || steps[s].kind == StepKind::PwConv) && first[steps[s].out] == s) first[steps[s].out] = std::max(0, s - 2);
  // a chain runs at its middle step's place: its input stays live until then, its output is written while other workgroups still read that input
  for (int s = 0; s < NS; s++)
    if (steps[s].chain_first >= 0) {
      const int tx = steps[steps[s].chain_first].in0, to = steps[steps[s].chain_last].out;
      last[tx] = std::max(last[tx], steps[s].chain_last);
      first[to] = std::min(first[to], std::max(0, steps[s].chain_first - 2));
    }
  first[g.input] = -1;
  last[g.output] = NS + 1;  // keep the network output alive for the decode stage
  last[g.input] = std::max(last[g.input], 0);
  struct Block { size_t off, len; int until; };
  std::vector<Block> live;
  size_t high = 0;
  std::vector<int> order;
  for (int t = 0; t < NT; t++) if (first[t] >= -1 && last[t] >= 0 && !g.tensors[t].is_const && (t == g.input || first[t] >= 0)) order.push_back(t);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
  for (int t : order) {
    size_t len = (g.tensors[t].elems() + 63) / 64 * 64;  // 256-byte granules
    int start = first[t];
    if (reuse_arena) live.erase(std::remove_if(live.begin(), live.end(), [&](const Block& b) { return b.until < start; }), live.end());
    std::sort(live.begin(), live.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
    size_t pos = 0;
    for (const Block& b : live) { if (pos + len <= b.off) break; pos = std::max(pos, b.off + b.len); }
    plan->tensor_off[t] = (long)pos;
    live.push_back({pos, len, last[t]});
    high = std::max(high, pos + len);
  }
  plan->arena_floats_per_stream = high;
  plan->input = g.input;
  plan->output = g.output;
  plan->macs_per_frame = 0;
  for (const Step& st : steps) plan->macs_per_frame += st.macs;
  plan->steps = std::move(steps);
  plan->seg = SegPlan();
  bool seg_ok = false;
  if (segments && !BSX_DBG_ENV("BSX_NO_SEGMENTS")) {
    // build_segments re-points tensor offsets, grows the arena and appends synthetic tensors BEFORE its last checks (an unsupported middle):
    // a failed attempt must leave the plan exactly as the unsegmented paths expect it
    const std::vector<long> off0 = plan->tensor_off;
    const size_t arena0 = plan->arena_floats_per_stream, nt0 = g.tensors.size();
    seg_ok = build_segments(g, plan);
    if (!seg_ok) { plan->tensor_off = off0; plan->arena_floats_per_stream = arena0; g.tensors.resize(nt0); }
  }
  if (!seg_ok) {
    plan->seg = SegPlan();
    build_frame_program(g, plan, plan->steps);
  }
  // ---- per-launch path (graphs without a frame program: DeepLab): expand 1x1 → depthwise 3x3 pairs of the inverted-residual blocks run as
  // ONE kernel (kernels_nn.hip: ir_expand_dw_k) — the 6x-expanded tensor, the largest write and read of the block, lives only in LDS.
  if (plan->program.empty() && !BSX_DBG_ENV("BSX_NO_IR_FUSE")) {
    std::vector<Step>& S = plan->steps;
    auto uses = [&](int t) { int n = 0; for (const Step& q : S) { for (int u : {q.concat_in.empty() ? q.in0 : -1, q.in1, q.in2, q.residual, q.in_scale, q.out_bias}) n += (u == t); for (int u : q.concat_in) n += (u == t); } return n + (t == g.output); };
    // stem conv 3x3/s2 (3 → 16) → depthwise 3x3 → 1x1 (16 → ≤16): one tiled kernel, the two 16-channel full-resolution tensors only in LDS
    if (S.size() > 3 && !BSX_DBG_ENV("BSX_NO_HEAD0")) {
      Step& c0 = S[0]; Step& d1 = S[1]; Step& p2 = S[2];
      const bool ok = c0.kind == StepKind::Conv && c0.kh == 3 && c0.kw == 3 && c0.sh == 2 && c0.sw == 2 && c0.dh == 1 && c0.dw == 1 && c0.Cin == 3 && c0.Cout == 16 &&
                      c0.cout_pad == 16 && c0.residual < 0 && c0.act < kActHswish && c0.in0 == g.input && uses(c0.out) == 1 &&
                      d1.kind == StepKind::DwConv && d1.in0 == c0.out && d1.kh == 3 && d1.kw == 3 && d1.sh == 1 && d1.sw == 1 && d1.dh == 1 && d1.dw == 1 &&
                      d1.pad_t == 1 && d1.pad_l == 1 && d1.OH == d1.H && d1.OW == d1.W && d1.Cin == 16 && d1.residual < 0 && d1.act < kActHswish && uses(d1.out) == 1 &&
                      p2.kind == StepKind::PwConv && p2.in0 == d1.out && p2.Cin == 16 && p2.Cout % 4 == 0 && p2.Cout <= 16 && p2.residual < 0 && p2.in_scale < 0 &&
                      p2.in2 < 0 && p2.out_bias < 0 && p2.act < kActHswish && p2.OH == d1.OH && p2.OW == d1.OW && p2.out != g.output &&
                      2 * (c0.OW - 1) - c0.pad_l + 2 <= c0.W && c0.pad_l >= 0 && c0.pad_l <= 1 && (c0.W + 2) * 3 <= 1024 && head0_band_rows(c0.W, c0.OW) >= 2 &&
                      (2 * (head0_band_rows(c0.W, c0.OW) + 2) + 1) * c0.W * 3 <= 4 * 8 * 512;      // the band's input rows: at most 8 quads per lane (dl_head0_k)
      if (ok) { c0.fuse_head0 = true; d1.fused_away = true; p2.fused_away = true; }
    }
    for (size_t i = 0; i + 1 < S.size(); i++) {
      Step& a = S[i];
      Step& d = S[i + 1];
      if (a.kind != StepKind::PwConv || a.k16_pad <= 0 || a.k16_pad > 96 || a.residual >= 0 || a.in_scale >= 0 || a.in2 >= 0 || a.out_bias >= 0 ||
          a.OH * a.OW < 64 || a.act >= kActHswish) continue;
      if (d.kind != StepKind::DwConv || d.in0 != a.out || d.kh != 3 || d.kw != 3 || d.sh != d.sw || d.dh != d.dw || d.residual >= 0 || d.Cin != a.Cout ||
          d.dh < 1 || d.dh > 4 || d.act >= kActHswish || uses(a.out) != 1) continue;
      if (d.sh == 1) { if (d.pad_t != d.dh || d.pad_l != d.dw || d.OH != d.H || d.OW != d.W) continue; }      // SAME, any dilation: sliding-window column walk
      else if (d.sh != 2 || d.dh != 1 || d.pad_t < 0 || d.pad_t > 1 || d.pad_l < 0 || d.pad_l > 1) continue;   // stride 2: plain 3x3
      if (ir_geometry(a.OH, a.OW, a.Cout, d.OH, d.sh, d.dh).CH == 0) continue;
      a.fuse_dw = (int)i + 1;
      d.fused_away = true;
      // the depthwise output's only reader, when it is a GEMM-capable 1x1: lets the reduced-precision storage mode keep that tensor in f16
      if (i + 2 < S.size()) {
        Step& pj = S[i + 2];
        if (pj.kind == StepKind::PwConv && pj.in0 == d.out && uses(d.out) == 1 && pj.k16_pad > 0 && pj.in_scale < 0 && pj.in2 < 0 && (pj.Cin & 3) == 0 &&
            pj.cout_pad % 16 == 0 && pj.OH == d.OH && pj.OW == d.OW)
          pj.in_from_fused_dw = true;
      }
    }
  }
  return true;
}

}  // namespace bsx
