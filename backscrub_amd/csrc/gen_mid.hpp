// gen_mid.hpp — source generator of the graph-specialised per-frame program (gen_mid.cpp).
#pragma once
#include <string>

#include "plan.hpp"

namespace bsx {

// HIP source of the kernel `bsx_mid(float* arena, long per_frame, const float* weights, unsigned long long* timeline)` specialised to
// plan.program (one 1024-lane workgroup per frame, static LDS of plan.program_lds_floats floats), or "" with the reason in *why.
// act16: the 16-bit activation storage mode — every activation tensor of the program that lives in the arena is read / written as packed halves
// (the segment kernels either side are launched with h16 = true to match).
// opaque_tid: the form in which no lane-derived value outlives its op (mid_prelude.hip: tid_now) — fewer registers, a few more instructions per op; chosen by
// build_mid_kernel (bsx_api.hip) where the plain form spills.
std::string generate_mid_source(const Plan& plan, std::string* why, bool act16 = false, bool opaque_tid = false);

// which ops of plan.program the generator fuses / chunks (shared with the planner's cost model, plan.cpp)
bool mid_dw_chunked(const MicroOp& d);
bool mid_pw_feeds_dw(const Plan& plan, int j);

}  // namespace bsx
