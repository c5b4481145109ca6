// gen_mid.cpp — emits the graph-specialised source of the per-frame network program (see mid_prelude.hip, rtc.hpp).
//
// Input: the planner's micro-op list (Plan::program — geometry, LDS offsets, weight slots, operand address spaces; plan.cpp).
// Output: one HIP translation unit = the device templates of mid_prelude.hip + one traits struct per op + a kernel that calls the
// op bodies in program order with one barrier between them.  Nothing in the kernel is read from a table at run time.
// Graphs with micro-ops the templates do not cover return an empty string: the interpreter (kernels_frame.hip) runs them.
#include "debug_switches.hpp"
#include "gen_mid.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

namespace bsx {
namespace {

const char kPrelude[] =
#include "mid_prelude_str.inc"      // written by build.py into the object directory (-I): csrc/mid_prelude.hip, comments stripped
    ;

struct Out {
  std::string s;
  void f(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {       // never truncates: a cut line could swallow the statement after it
    char buf[2048];
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    const int need = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (need >= 0 && (size_t)need < sizeof buf) s.append(buf, (size_t)need);
    else if (need >= 0) { std::string big((size_t)need + 1, '\0'); vsnprintf(&big[0], big.size(), fmt, ap2); big.resize((size_t)need); s += big; }
    va_end(ap2);
  }
};

// an op label as a one-line comment: no line breaks, nothing that could close or continue a comment
std::string comment_safe(const std::string& in) {
  std::string o;
  for (char ch : in.substr(0, 200)) o += (ch == '\n' || ch == '\r' || ch == '\\' || (unsigned char)ch < 32) ? ' ' : ch;
  return o;
}

int sp_of(const Loc& l) { return l.space == kLocNone ? 0 : (l.space == kLocLds ? 1 : 2); }      // SP_NONE / SP_LDS / SP_GLB
bool plain(const Loc& l) { return l.space == kLocNone || l.space == kLocLds || l.space == kLocGlobal; }   // no network input / output buffers in a middle program


}  // namespace


// depthwise ops the strip body covers, and those among them that run chunk by chunk through an LDS workspace the planner reserved (input in the arena)
static bool dw_geom_ok(const MicroOp& d) { return d.strip && d.dh == 1 && d.dw == 1 && d.kh == d.kw && (d.kh == 3 || d.kh == 5) && d.sh == d.sw && (d.sh == 1 || d.sh == 2) && d.Cin % 4 == 0; }
bool mid_dw_chunked(const MicroOp& d) {
  return d.kind == (int)StepKind::DwConv && dw_geom_ok(d) && d.in0.space == kLocGlobal && d.band_rows > 0 && (d.Cin % d.band_rows) % 8 == 0 && d.res.space == kLocNone &&
         !BSX_DBG_ENV("BSX_RTC_NO_DW_STAGE");
}
// 1x1 → depthwise without the tensor in between (see generate_mid_source): is op j a 1x1 whose arena output is only read by the chunked depthwise j + 1, with every
// LDS region the 1x1 still needs disjoint from what the depthwise places?  Shared with the planner's cost model (plan.cpp: program_arena_bytes) — an elided tensor
// costs no arena bytes.
bool mid_pw_feeds_dw(const Plan& plan, int j) {
  const std::vector<MicroOp>& P = plan.program;
  const int n = (int)P.size();
  auto disjoint = [](long a0, long a1, long b0, long b1) { return a1 <= b0 || b1 <= a0; };
  if (BSX_DBG_ENV("BSX_RTC_NO_PWDW") || j < 0 || j + 1 >= n) return false;
  const MicroOp& a = P[j];
  const MicroOp& d = P[j + 1];
  if (!(a.kind == (int)StepKind::PwConv && a.mfma && !a.gemv && a.stage_floats > 0 && a.out.space == kLocGlobal && a.res.space == kLocNone) || !mid_dw_chunked(d)) return false;
  if (d.in0.off != a.out.off || d.Cin != a.Cout || d.band_rows % 16 || a.cout_pad < a.Cout) return false;
  for (long e : plan.program_ext_offs) if (e == a.out.off) return false;           // a segment kernel reads this tensor after the program: it must exist in the arena
  for (int q = j + 2; q < n; q++) {                               // the expanded tensor has no other reader (until its arena slot is written again: slots are re-used)
    for (const Loc* l : {&P[q].in0, &P[q].in1, &P[q].in2, &P[q].res, &P[q].scale}) if (l->space == kLocGlobal && l->off == a.out.off) return false;
    for (int c = 0; c < P[q].n_cat; c++) if (P[q].cat[c].space == kLocGlobal && P[q].cat[c].off == a.out.off) return false;
    if (P[q].out.space == kLocGlobal && P[q].out.off == a.out.off) break;
  }
  const long pin = (long)a.H * a.W, ws0 = d.ws_off, ws1 = ws0 + (long)d.H * d.W * (d.band_rows + 4);
  std::vector<std::pair<long, long>> keep, placed;
  keep.push_back({a.w_lds, a.w_lds + a.stage_floats});                                   // the 1x1's staged weights
  if (a.in0.space == kLocLds) keep.push_back({a.in0.off, a.in0.off + pin * a.in0.stride});
  if (a.in2.space == kLocLds) keep.push_back({a.in2.off, a.in2.off + pin * a.in2.stride});
  if (a.scale.space == kLocLds) keep.push_back({a.scale.off, a.scale.off + a.Cin});
  placed.push_back({ws0, ws1});
  if (d.stage_floats > 0) placed.push_back({d.w_lds, d.w_lds + d.stage_floats});
  if (d.out.space == kLocLds) placed.push_back({d.out.off, d.out.off + (long)d.OH * d.OW * d.out.stride});
  for (auto& kq : keep) for (auto& pq : placed) if (!disjoint(kq.first, kq.second, pq.first, pq.second)) return false;
  return true;
}

std::string generate_mid_source(const Plan& plan, std::string* why, bool act16, bool opaque_tid) {
  // address space of an ACTIVATION tensor operand: arena tensors are packed halves in the 16-bit storage mode (mid_prelude.hip: SP_GLB16)
  auto asp = [&](const Loc& l) { const int sp = sp_of(l); return (sp == 2 && act16) ? 3 : sp; };
  auto loc = [&](Out& o, const char* p, const Loc& l) { o.f("  static constexpr int %s_SP = %d, %s_OFF = %d, %s_ST = %d;\n", p, asp(l), p, l.off, p, l.stride); };
  auto fail = [&](const std::string& m) { if (why) *why = m; return std::string(); };
  const std::vector<MicroOp>& P = plan.program;
  if (P.empty()) return fail("no program");
  // BSX_RTC_FINE=1 (timing experiments, tools/program_timeline.py --fine): per-wave shader-clock stamps around the barrier and the body of every op
  // (a different source, i.e. a different cache entry, from the product kernel)
  const bool fine = BSX_DBG_ENV("BSX_RTC_FINE") != nullptr;
  Out o;
  o.s.reserve(sizeof kPrelude + 64 * 1024);
  if (BSX_DBG_ENV("BSX_RTC_EXP_MFMA")) o.s += "#define BSXM_EXP_MFMA_QUARTER 1   // timing experiment: WRONG RESULTS (mid_prelude.hip op_pw)\n";
  // the workgroup's geometry (Plan::mid_lanes, Plan::lds_total_floats): the prelude's kThreads / kWaves / kZeroOff come from these two macros
  o.f("#define BSXM_LANES %d\n#define BSXM_ZERO_OFF %d\n", plan.mid_lanes, plan.lds_zero_off());
  if (opaque_tid) o.s += "#define BSXM_OPAQUE_TID 1\n";      // every op re-derives its lane indices (mid_prelude.hip: tid_now) — the form for graphs whose plain kernel spills
  o.s += kPrelude;
  o.f("\nnamespace bsxm {\n");
  std::string body;
  Out k;
  const int n = (int)P.size();
  // weight staging of op i (issued while op i-1 runs; op 0's before the first barrier)
  auto stage_of = [&](int i, Out& dst) {
    const MicroOp& m = P[i];
    if (m.stage_floats > 0) dst.f("  stage<%d>(W + %lld, L + %d);\n", m.stage_floats, m.w_off, m.w_lds);
    else {
      if (m.fc_stage[0] > 0) dst.f("  stage<%d>(W + %lld, L + %d);\n", m.fc_stage[0], m.b_off, m.fc_lds[0]);
      if (m.fc_stage[1] > 0) dst.f("  stage<%d>(W + %lld, L + %d);\n", m.fc_stage[1], m.b3_off, m.fc_lds[1]);
    }
  };
  stage_of(0, k);
  // 1x1 → depthwise without the tensor in between: when op j is a 1x1 convolution whose output lives in the arena and whose only reader is the chunked
  // depthwise j + 1 (plan.cpp reserved an LDS workspace for ITS input chunks), every chunk of the expanded tensor is COMPUTED straight into the workspace instead
  // of being written to the arena by op j and loaded back by op j + 1.  Op j then emits nothing but its barrier.  The regions the 1x1 needs while op j + 1 runs
  // (its input if that is in LDS, its staged weights) were planned to live until op j only: the fusion is taken only where they are disjoint from everything op j + 1
  // places (workspace, depthwise weights, depthwise output), and the DMA of op j + 2's weights — normally issued at the top of op j + 1 — waits for the last chunk.
  auto dw_chunked = [&](const MicroOp& d) { return mid_dw_chunked(d); };
  (void)dw_chunked;
  auto pw_feeds_dw = [&](int j) { return !fine && mid_pw_feeds_dw(plan, j); };
  for (int i = 0; i < n; i++) {
    const MicroOp& m = P[i];
    for (const Loc* l : {&m.in0, &m.in1, &m.in2, &m.res, &m.scale, &m.out}) if (!plain(*l)) return fail("operand in the network input / output buffer");
    if (fine && i > 0 && i <= 64) k.f("  FINE_END(%d);\n", i - 1);
    if (fine) k.f("  f_a = __builtin_readcyclecounter();\n");
    k.f("  op_barrier();\n  if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[%d] = __builtin_amdgcn_s_memrealtime();\n", i);
    k.f("  // ---- P%d %s\n", i, i < (int)plan.program_labels.size() ? comment_safe(plan.program_labels[i]).c_str() : "");      // label AFTER the barrier it belongs to
    if (fine) k.f("  f_b = __builtin_readcyclecounter();\n");
    const bool fed_by_prev = i > 0 && pw_feeds_dw(i - 1);            // this depthwise computes its input chunks itself: op i + 1's weight DMA waits for the last one
    if (i + 1 < n && !fed_by_prev) stage_of(i + 1, k);
    // The FC weights of a squeeze-excite op come from L2 (64 KB per 128 x 128 layer, the same bytes for every workgroup): requested ONE OP EARLY,
    // into registers that stay live across the depthwise / 1x1 op in front of it, their delivery overlaps that op instead of stalling the FCs.
    auto fc_loads = [&](int j) {
      const MicroOp& q = P[j];
      auto fcl = [&](int which, int cin, int cout, int stage, int lds, long long w2, long long b) {
        k.f("  FcRegs<%d, %d> fc%d_%d;\n", cin, cout, j, which);
        if (stage > 0) k.f("  fc_load<%d, %d, SP_LDS, %d, %d>(L, W, fc%d_%d);\n", cin, cout, lds + (int)(w2 - b), lds, j, which);
        else k.f("  fc_load<%d, %d, SP_GLB, %lld, %lld>(L, W, fc%d_%d);\n", cin, cout, w2, b, j, which);
      };
      fcl(1, q.Cin, q.C1, q.fc_stage[0], q.fc_lds[0], q.w2_off, q.b_off);
      if (q.n_fc == 2) fcl(2, q.C1, q.C2, q.fc_stage[1], q.fc_lds[1], q.w3_off, q.b3_off);
    };
    auto early = [&](int j) {        // may the loads of SE op j be issued at the top of op j - 1?  (unstaged weights only: a staged block lands at j's barrier)
      if (j <= 0 || j >= n || P[j].kind != kMicroSe || P[j].fc_stage[0] > 0 || P[j].fc_stage[1] > 0) return false;
      if (BSX_DBG_ENV("BSX_RTC_NO_EARLY_FC")) return false;
      // register budget: up to 34 live FC registers + the op's own.  1x1 ops need ~35; the depthwise bodies 60-78 in their fully unrolled LDS form
      // (op_dw: K (NIN + K) V <= 144) and ~105 in the one-row-ahead form, which would spill
      const MicroOp& pv = P[j - 1];
      if (pv.kind == (int)StepKind::PwConv && pv.mfma && !pv.gemv) return true;
      if (pv.kind != (int)StepKind::DwConv || pv.in0.space != kLocLds) return false;
      const int K = pv.kh, S = pv.sh, V = K == 5 ? 2 : 4, TX = (S == 1 && !(pv.OW % 5 != 0 && pv.OW % 4 == 0)) ? 5 : 4, NIN = (TX - 1) * S + K;
      return K * (NIN + K) * V <= 144;
    };
    if (early(i + 1)) fc_loads(i + 1);
    if (m.kind == (int)StepKind::PwConv && m.mfma && !m.gemv) {
      if (m.Cin % 4 || m.Cout % 4 || m.cout_pad % 16 || m.stage_floats <= 0) return fail("pw: channel counts / unstaged weights");   // op_pw stores 16-byte channel quads
      if (m.scale.space != kLocNone && m.scale.space != kLocLds) return fail("pw: scale vector outside LDS");
      o.f("struct Op%d {\n  static constexpr int P = %d, CIN = %d, COUT = %d, CPAD = %d, ACT = %d;\n", i, m.OH * m.OW, m.Cin, m.Cout, m.cout_pad, m.act);
      loc(o, "X", m.in0); loc(o, "Y", m.out); loc(o, "R", m.res); loc(o, "D", m.in2);
      o.f("  static constexpr int S_OFF = %d, W_LDS = %d, B_LDS = %d;\n", m.scale.space == kLocLds ? m.scale.off : -1, m.w_lds, m.w_lds + (int)(m.b_off - m.w_off));
      o.f("  static constexpr int N0 = 0, NCOLS = %d, YSUB = 0;\n", m.cout_pad);
      o.f("  static constexpr bool NFAST = %s;\n};\n", ((m.out.space == kLocGlobal && !BSX_DBG_ENV("BSX_RTC_NO_NFAST")) || (m.in0.space == kLocGlobal && m.cout_pad > 16 && !BSX_DBG_ENV("BSX_RTC_NO_NFAST_IN"))) ? "true" : "false");      // arena INPUT: the column tiles of one row tile run back to back, its A rows are fetched once
      if (pw_feeds_dw(i)) k.f("  // (computed chunk by chunk inside P%d: the tensor between them is never written)\n", i + 1);
      else k.f("  op_pw<Op%d>(L, A);\n", i);
    } else if (m.kind == (int)StepKind::DwConv) {
      const bool ok = m.strip && m.dh == 1 && m.dw == 1 && m.kh == m.kw && (m.kh == 3 || m.kh == 5) && m.sh == m.sw && (m.sh == 1 || m.sh == 2) && m.Cin % 4 == 0;
      if (!ok) return fail("dw: geometry outside the strip form");
      const int K = m.kh, S = m.sh, V = K == 5 ? 2 : 4;
      int TX = S == 1 ? 5 : 4;
      if (S == 1 && m.OW % 5 != 0 && m.OW % 4 == 0) TX = 4;
      const bool staged = m.stage_floats > 0;
      if (!staged && (m.w_off > 0x7fffffffll || m.b_off > 0x7fffffffll)) return fail("dw: weight offset");
      // one traits struct per channel chunk: the whole layer (CK = C), or — input in the arena and an LDS workspace planned for it (plan.cpp) — chunks of
      // band_rows channels staged through the workspace by load_chunk
      const bool chunked = mid_dw_chunked(m);
      const int CK = chunked ? m.band_rows : m.Cin, nch = (m.Cin + CK - 1) / CK;
      for (int c = 0; c < nch; c++) {
        const int CKc = std::min(CK, m.Cin - c * CK);            // the last chunk may be ragged (88 channels = 5 x 16 + 8); the workspace keeps its CK + 4 row stride
        char name[32];
        if (chunked) snprintf(name, sizeof name, "Op%d_%d", i, c); else snprintf(name, sizeof name, "Op%d", i);
        o.f("struct %s {\n  static constexpr int K = %d, S = %d, H = %d, W = %d, OH = %d, OW = %d, PT = %d, PL = %d, C = %d, CW = %d, YC0 = %d, ACT = %d, V = %d, TX = %d;\n", name, K, S,
            m.H, m.W, m.OH, m.OW, m.pt, m.pl, CKc, m.Cin, c * CK, m.act, V, TX);
        // ZC: out-of-image taps through the zero cell (op_dw) — for every depthwise op whose input is in LDS: a planned tensor, or (round 6) the workspace of the
        // chunk-by-chunk form.  Round 5 had to leave the chunked form out: MLKit's kernel sat at 128 registers and the zero-cell form took its spill from 372 to 524
        // bytes (+7 %, profiles/r05j); with the lane index read through tid_now() (mid_prelude.hip) that kernel needs 92 registers and nothing spills.
        // (debug build: BSX_RTC_NO_ZERO_CELL=1 switches the form off everywhere, BSX_RTC_NO_ZC_CHUNK=1 for the chunked ops only — the A/B switches)
        o.f("  static constexpr bool ZC = %s;\n", ((chunked && BSX_DBG_ENV("BSX_RTC_NO_ZC_CHUNK")) || BSX_DBG_ENV("BSX_RTC_NO_ZERO_CELL")) ? "false" : "true");
        if (chunked) o.f("  static constexpr int X_SP = 1, X_OFF = %d, X_ST = %d;\n", m.ws_off, CK + 4); else loc(o, "X", m.in0);
        loc(o, "Y", m.out); loc(o, "R", m.res);
        if (staged) o.f("  static constexpr int W_SP = SP_LDS, W_OFF = %d, B_OFF = %d;\n};\n", m.w_lds + c * CK, m.w_lds + (int)(m.b_off - m.w_off) + c * CK);
        else o.f("  static constexpr int W_SP = SP_GLB, W_OFF = %lld, B_OFF = %lld;\n};\n", m.w_off + c * CK, m.b_off + c * CK);
        if (chunked) {
          if (c > 0) k.f("  __syncthreads();\n");                 // the previous chunk's taps are done with the workspace
          if (fed_by_prev) {
            const MicroOp& a = P[i - 1];
            o.f("struct Op%d_%d {\n  static constexpr int P = %d, CIN = %d, COUT = %d, CPAD = %d, ACT = %d;\n", i - 1, c, a.OH * a.OW, a.Cin, a.Cout, a.cout_pad, a.act);
            loc(o, "X", a.in0); loc(o, "R", a.res); loc(o, "D", a.in2);
            o.f("  static constexpr int Y_SP = 1, Y_OFF = %d, Y_ST = %d;\n", m.ws_off, CK + 4);
            o.f("  static constexpr int S_OFF = %d, W_LDS = %d, B_LDS = %d;\n", a.scale.space == kLocLds ? a.scale.off : -1, a.w_lds, a.w_lds + (int)(a.b_off - a.w_off));
            o.f("  static constexpr int N0 = %d, NCOLS = %d, YSUB = %d;\n  static constexpr bool NFAST = true;\n};\n", c * CK, (CKc + 15) / 16 * 16, c * CK);
            k.f("  op_pw<Op%d_%d>(L, A);\n  __syncthreads();\n", i - 1, c);
            if (c == nch - 1 && i + 1 < n) stage_of(i + 1, k);    // every wave is past the last 1x1 chunk: its weight slot is free for the next op's DMA only now
          } else
            k.f("  load_chunk<%d, %d, %d, %d, %d, %d, %d, %d>(L, A);\n  __syncthreads();\n", asp(m.in0), m.in0.off, m.H * m.W, m.in0.stride, CKc, c * CK, m.ws_off, CK + 4);
        }
        k.f("  op_dw<%s>(L, A, W);\n", name);
      }
    } else if (m.kind == kMicroSe) {
      if (m.in1.space != kLocLds || (m.n_fc == 2 && m.in2.space != kLocLds) || m.Cin % 4 || m.C1 % 1) return fail("se: mean / hidden vectors outside LDS");
      if (m.w2_off > 0x7fffffffll || m.w3_off > 0x7fffffffll) return fail("se: weight offset");
      if (!early(i)) fc_loads(i);        // staged weights (LDS) or no op in front: requested here, ahead of the pooling
      // pooling parts → mean vector (LDS, at in1)
      const int mean = m.in1.off;
      auto part = [&](const Loc& l, int rows, int C, int hw, int coff, bool accum, bool partials) {
        const int st = partials ? C : l.stride;
        k.f("  gap_part<%d, %d, %d, %d, %d, %d, %d, %s, %d>(L, A);\n", partials ? 2 : asp(l), l.off, st, rows, C, hw, coff, accum ? "true" : "false", mean);
      };
      if (m.n_cat == 0) part(m.in0, m.H * m.W, m.Cin, m.H * m.W, 0, false, false);
      else {
        int coff = 0;
        for (int c = 0; c < m.n_cat; c++) {
          if (!plain(m.cat[c]) || m.cat_c[c] % 4) return fail("se: pooled part");
          const bool partials = m.cat_parts[c] > 0;
          if (partials && m.cat[c].space != kLocGlobal) return fail("se: partial sums outside the arena");
          part(m.cat[c], partials ? m.cat_parts[c] : m.cat_hw[c], m.cat_c[c], m.cat_hw[c], coff, m.gap_sum && c > 0, partials);
          if (!m.gap_sum) coff += m.cat_c[c];
        }
      }
      k.f("  __syncthreads();\n");
      const Loc& y1 = m.n_fc == 1 ? m.out : m.in2;
      k.f("  fc_apply<%d, %d, %d, %d, %d, %d>(L, A, fc%d_1);\n", m.Cin, m.C1, m.act, mean, sp_of(y1), y1.off, i);
      if (m.n_fc == 2) {
        k.f("  __syncthreads();\n");
        k.f("  fc_apply<%d, %d, %d, %d, %d, %d>(L, A, fc%d_2);\n", m.C1, m.C2, m.act2, m.in2.off, sp_of(m.out), m.out.off, i);
      }
    } else if (m.kind == (int)StepKind::Resize) {
      if (m.Cin % 4) return fail("resize: channels");
      o.f("struct Op%d {\n  static constexpr int H = %d, W = %d, OH = %d, OW = %d, C = %d;\n  static constexpr bool HALF_PIXEL = %s, ALIGN = %s;\n", i, m.H, m.W, m.OH, m.OW, m.Cin,
          m.half_pixel ? "true" : "false", m.align_corners ? "true" : "false");
      loc(o, "X", m.in0); loc(o, "Y", m.out);
      o.f("};\n");
      k.f("  op_resize<Op%d>(L, A);\n", i);
    } else {
      return fail("micro-op kind " + std::to_string(m.kind) + " has no specialised body");
    }
  }
  o.f("}  // namespace bsxm\n\nusing namespace bsxm;\n");
  // two workgroups per CU (<= half of the LDS) must also fit the register file twice: 16 waves per CU = 4 per SIMD = 128 registers — said to the compiler, which would
  // otherwise take the 256 a lone 512-lane workgroup could have
  const bool two_per_cu = plan.mid_lanes <= 512 && plan.lds_total_floats <= kLdsTotalFloats / 2;
  o.f("extern \"C\" __global__ void __launch_bounds__(%d)%s bsx_mid(float* __restrict__ arena, long per_frame, const float* __restrict__ weights, unsigned long long* tl) {\n", plan.mid_lanes,
      two_per_cu ? " __attribute__((amdgpu_waves_per_eu(4, 4)))" : "");
  // the plan's whole LDS block: the planner's tensors and slots end below its zero cell — the last 16 bytes, which the depthwise bodies read out-of-image taps from
  if (plan.program_lds_floats > plan.lds_zero_off()) return fail("LDS plan reaches into the zero cell");
  o.f("  static_assert(kZeroOff == %d && kThreads == %d, \"geometry macros\");\n", plan.lds_zero_off(), plan.mid_lanes);
  o.f("  __shared__ __attribute__((aligned(16))) float smem[%d];\n", plan.lds_total_floats);
  o.f("  lds_f* L = (lds_f*)smem;\n  glb_f* A = (glb_f*)(arena + (size_t)blockIdx.x * (size_t)per_frame);\n  const glb_f* W = (const glb_f*)weights;\n");
  o.f("  if (threadIdx.x < %d) L[kZeroOff + threadIdx.x] = 0.f;      // visible to every wave behind the first op's barrier\n", kLdsZeroFloats);
  if (fine) o.f("  unsigned long long f_a = 0, f_b = 0;\n"
                "#define FINE_END(i) do { const unsigned long long f_d = __builtin_readcyclecounter(); if (tl && blockIdx.x == 0 && (threadIdx.x & 63) == 0) { "
                "unsigned long long* f4 = tl + 1024 + ((i) * 16 + (threadIdx.x >> 6)) * 4; f4[0] = f_b - f_a; f4[1] = 0; f4[2] = f_d - f_b; f4[3] = f_a; } } while (0)\n");
  o.s += k.s;
  if (fine && n <= 64) o.f("  FINE_END(%d);\n", n - 1);
  o.f("  op_barrier();\n  if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[%d] = __builtin_amdgcn_s_memrealtime();\n}\n", n);
  return o.s;
}

}  // namespace bsx
