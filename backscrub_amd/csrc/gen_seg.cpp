// gen_seg.cpp — source generator of the graph-specialised SEGMENT kernels (round 6).
//
// The segment kernels (kernels_seg.hip: head | k2 | k3 | tail, segments.hpp) take their geometry, weight offsets and activation kinds from a descriptor passed as a kernel
// argument.  When a context is created the same source text is compiled once more by hipRTC with the descriptors of THE LOADED GRAPH in front as compile-time constants
// (and the template arguments of the one variant this graph / mode needs as macros): extern "C" kernels bsx_seg_head / _k2 / _k3 / _tail.  Same arithmetic in the same
// order — the results are bit-identical to the ahead-of-time kernels (tests/test_gpu_parity.py) — with loop trip counts, index arithmetic and activation branches resolved
// by the compiler.  The ahead-of-time kernels stay: they run when hipRTC is unavailable, and they are the logits-writing variant of the tail the stage tests use.
//
// Reference: Interpreter::Invoke() (/root/reference/lib/libbackscrub.cc:307) builds its execution plan for the loaded graph when the interpreter is created (:205-217);
// this is that step for the high-resolution ends of the Meet / MLKit networks.
#include "gen_seg.hpp"

#include <cstdarg>
#include <cstdio>

namespace bsx {
namespace {

// segments.hpp + mfma_tile.hpp + kernels_seg.hip, comments stripped, flattened by backscrub_amd/build.py behind a small hipRTC preamble; the marker line BSXS_SEG_CONSTANTS
// marks where the constants go (inside namespace bsx::segrtc, after the descriptor structs)
const char kSegSource[] =
#include "seg_rtc_src.inc"
    ;

}  // namespace

std::string seg_constants_text(const SegPlan& sp, const char* pre) {
  std::string out;
  auto P = [&](const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    out += buf;
  };
  auto conv = [&](const char* n, const SegConvW& c) { P("  t.%s.w_off = %lldll; t.%s.b_off = %lldll; t.%s.Cin = %d; t.%s.Cout = %d; t.%s.cout_pad = %d; t.%s.act = %d;\n", n, c.w_off, n, c.b_off, n, c.Cin, n, c.Cout, n, c.cout_pad, n, c.act); };
  auto dw = [&](const char* n, const SegDwW& c) { P("  t.%s.w_off = %lldll; t.%s.b_off = %lldll; t.%s.C = %d; t.%s.act = %d;\n", n, c.w_off, n, c.b_off, n, c.C, n, c.act); };
  auto gate = [&](const char* n, const SegGate& g_) {
    P("  t.%s.n_parts = %d; t.%s.sum_parts = %d; t.%s.n_fc = %d; t.%s.timing_skip = 0;\n", n, g_.n_parts, n, g_.sum_parts, n, g_.n_fc, n);
    for (int i = 0; i < 2; i++) {
      P("  t.%s.part[%d].off = %lldll; t.%s.part[%d].n = %d; t.%s.part[%d].C = %d; t.%s.part[%d].hw = (float)%.9g;\n", n, i, g_.part[i].off, n, i, g_.part[i].n, n, i, g_.part[i].C, n, i, (double)g_.part[i].hw);
      P("  t.%s.fc[%d].w_off = %lldll; t.%s.fc[%d].b_off = %lldll; t.%s.fc[%d].Cin = %d; t.%s.fc[%d].Cout = %d; t.%s.fc[%d].act = %d;\n", n, i, g_.fc[i].w_off, n, i, g_.fc[i].b_off, n, i, g_.fc[i].Cin, n, i,
              g_.fc[i].Cout, n, i, g_.fc[i].act);
    }
  };
  const SegHead& H = sp.head; const SegK2& K2 = sp.k2; const SegK3& K3 = sp.k3; const SegTail& T = sp.tail;
  P("// segment descriptors of the loaded graph as constants (gen_seg.cpp: seg_constants_text)\nconstexpr SegHead %sHEAD = [] { SegHead t{};\n", pre);
  P("  t.H0 = %d; t.W0 = %d; t.H1 = %d; t.W1 = %d; t.H2 = %d; t.W2 = %d; t.stem_pt = %d; t.stem_pl = %d; t.dw_pt = %d; t.dw_pl = %d;\n", H.H0, H.W0, H.H1, H.W1, H.H2, H.W2, H.stem_pt, H.stem_pl, H.dw_pt, H.dw_pl);
  conv("stem", H.stem); conv("pw", H.pw); dw("dw", H.dw);
  P("  t.a_off = %lldll; t.b0_off = %lldll; t.part_a_off = %lldll; t.part_b0_off = %lldll; t.TR = %d; t.TC = %d; t.tiles_y = %d; t.tiles_x = %d; t.lds_floats = %d; t.rw = %d; t.m_ct = %uu;\n  return t; }();\n",
          H.a_off, H.b0_off, H.part_a_off, H.part_b0_off, H.TR, H.TC, H.tiles_y, H.tiles_x, H.lds_floats, H.rw, H.m_ct);
  P("constexpr SegK2 %sK2 = [] { SegK2 t{};\n  t.H2 = %d; t.W2 = %d; t.H3 = %d; t.W3 = %d; t.dw_pt = %d; t.dw_pl = %d;\n", pre, K2.H2, K2.W2, K2.H3, K2.W3, K2.dw_pt, K2.dw_pl);
  gate("gate", K2.gate); conv("pw_a", K2.pw_a); conv("pw_b", K2.pw_b); dw("dw", K2.dw);
  P("  t.b0_off = %lldll; t.B_off = %lldll; t.c0_off = %lldll; t.part_B_off = %lldll; t.TR = %d; t.TC = %d; t.tiles_y = %d; t.tiles_x = %d; t.lds_floats = %d; t.rw = %d; t.m_ct = %uu;\n  return t; }();\n",
          K2.b0_off, K2.B_off, K2.c0_off, K2.part_B_off, K2.TR, K2.TC, K2.tiles_y, K2.tiles_x, K2.lds_floats, K2.rw, K2.m_ct);
  P("constexpr SegK3 %sK3 = [] { SegK3 t{};\n  t.H2 = %d; t.W2 = %d; t.HL = %d; t.WL = %d; t.half_pixel = %d; t.align_corners = %d;\n", pre, K3.H2, K3.W2, K3.HL, K3.WL, K3.half_pixel, K3.align_corners);
  conv("pw1", K3.pw1); conv("pw2", K3.pw2); dw("dw", K3.dw);
  P("  t.skip_off = %lldll; t.lo2_off = %lldll; t.g_off = %lldll; t.lo_off = %lldll; t.part_lo_off = %lldll; t.TR = %d; t.TC = %d; t.tiles_y = %d; t.tiles_x = %d; t.lds_floats = %d; t.lo_floats = %d; t.hs = (float)%.9g; t.ws = (float)%.9g;\n  return t; }();\n",
          K3.skip_off, K3.lo2_off, K3.g_off, K3.lo_off, K3.part_lo_off, K3.TR, K3.TC, K3.tiles_y, K3.tiles_x, K3.lds_floats, K3.lo_floats, (double)K3.hs, (double)K3.ws);
  P("constexpr SegTail %sTAIL = [] { SegTail t{};\n  t.H1 = %d; t.W1 = %d; t.HL = %d; t.WL = %d; t.H0 = %d; t.W0 = %d; t.half_pixel = %d; t.align_corners = %d;\n", pre, T.H1, T.W1, T.HL, T.WL, T.H0, T.W0, T.half_pixel, T.align_corners);
  gate("gate", T.gate); conv("pw", T.pw); dw("dw", T.dw);
  P("  t.tc_w_off = %lldll; t.tc_b_off = %lldll; t.Co = %d; t.act3 = %d; t.model_type = %d; t.skip_off = %lldll; t.lo_off = %lldll; t.pre_gate_off = %lldll; t.TR = %d; t.TC = %d; t.tiles_y = %d; t.tiles_x = %d; "
             "t.lds_floats = %d; t.lo_floats = %d; t.hs = (float)%.9g; t.ws = (float)%.9g;\n  return t; }();\n",
          T.tc_w_off, T.tc_b_off, T.Co, T.act3, T.model_type, T.skip_off, T.lo_off, T.pre_gate_off, T.TR, T.TC, T.tiles_y, T.tiles_x, T.lds_floats, T.lo_floats, (double)T.hs, (double)T.ws);
  return out;
}

std::string generate_seg_source(const Plan& plan, bool h16, bool u8in, std::string* why) {
  auto fail = [&](const char* m) { if (why) *why = m; return std::string(); };
  if (!plan.seg.on) return fail("no segment kernels in this plan");
  const SegPlan& sp = plan.seg;
  const bool sig = sp.tail.act3 == kActSigmoid;
  if (!((sp.tail.Co == 2 && !sig) || sp.tail.Co == 1)) return fail("transpose-convolution output channels");
  std::string src = kSegSource;
  const std::string mark = "BSXS_SEG_CONSTANTS";      // (build.py renames every BSX_ macro of the embedded text to BSXS_)
  const size_t at = src.find(mark);
  if (at == std::string::npos) return fail("embedded source has no constants marker");
  src.replace(at, mark.size(), seg_constants_text(sp, "kSeg"));
  char head[512];
  snprintf(head, sizeof head, "#define BSXS_SEG_RTC 1\n#define BSXS_SEG_HS %d\n#define BSXS_SEG_H16 %d\n#define BSXS_SEG_U8 %d\n#define BSXS_SEG_SIG %d\n#define BSXS_SEG_CO %d\n",
           sp.head.stem.act == kActHswish ? 1 : 0, h16 ? 1 : 0, u8in ? 1 : 0, sig ? 1 : 0, sp.tail.Co);
  return std::string(head) + src;
}

}  // namespace bsx
