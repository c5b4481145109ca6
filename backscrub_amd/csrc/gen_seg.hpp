// gen_seg.hpp — source generator of the graph-specialised segment kernels (gen_seg.cpp).
#pragma once
#include <string>

#include "plan.hpp"

namespace bsx {

// HIP source of extern "C" kernels bsx_seg_head / bsx_seg_k2 / bsx_seg_k3 / bsx_seg_tail (the decode-fused tail) specialised to plan.seg, or "" with the reason in *why.
// h16: 16-bit activation storage at the segment boundaries (BSX_ACT16); u8in: the head reads the 8-bit network input (the step's prep output).
std::string generate_seg_source(const Plan& plan, bool h16, bool u8in, std::string* why);

// the four descriptors as `constexpr SegHead <pre>HEAD = ...; <pre>K2; <pre>K3; <pre>TAIL` (the generator's constants; the debug build's BSX_SEG_DUMP writes the same text)
std::string seg_constants_text(const SegPlan& sp, const char* pre);

}  // namespace bsx
