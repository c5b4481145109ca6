// mfma_tile.hpp — device helpers shared by kernels_frame.hip and kernels_nn.hip for v_mfma_f32_16x16x4_f32 tiles.
//
// Accumulator layout of the 16x16 tile (cdna_hip_programming.md §3): lane l = (g = l >> 4, li = l & 15) holds rows
// 4g..4g+3 of column li.  quad_transpose() is a 4x4 transpose inside each quad of adjacent lanes (two DPP exchanges):
// afterwards lane (g, li) holds row 4g + (li & 3), columns 4*(li >> 2) .. +3 — four consecutive channels of one pixel,
// i.e. one 16-byte store (and one 16-byte bias / residual load) instead of four 4-byte ones.
#pragma once
#include <hip/hip_runtime.h>

namespace bsx {

typedef float f4acc __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_quad(float v, const int ctrl_xor) {   // ctrl_xor: 1 → lane^1, 2 → lane^2 (inside a quad)
  const int i = __builtin_bit_cast(int, v);
  const int r = ctrl_xor == 1 ? __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xf, 0xf, true);
  return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ float4 quad_transpose(const f4acc acc, int q) {
  const bool odd = q & 1, hi = q & 2;
  const float rx = dpp_quad(odd ? acc[0] : acc[1], 1), ry = dpp_quad(odd ? acc[2] : acc[3], 1);
  const float t0 = odd ? rx : acc[0], t1 = odd ? acc[1] : rx, t2 = odd ? ry : acc[2], t3 = odd ? acc[3] : ry;
  const float r0 = dpp_quad(hi ? t0 : t2, 2), r1 = dpp_quad(hi ? t1 : t3, 2);
  return hi ? make_float4(r0, r1, t2, t3) : make_float4(t0, t1, r0, r1);
}

}  // namespace bsx
