// mfma_tile.hpp — device helpers shared by kernels_frame.hip and kernels_nn.hip for v_mfma_f32_16x16x4_f32 tiles.
//
// Accumulator layout of the 16x16 tile (cdna_hip_programming.md §3): lane l = (g = l >> 4, li = l & 15) holds rows
// 4g..4g+3 of column li.  quad_transpose() is a 4x4 transpose inside each quad of adjacent lanes (two DPP exchanges):
// afterwards lane (g, li) holds row 4g + (li & 3), columns 4*(li >> 2) .. +3 — four consecutive channels of one pixel,
// i.e. one 16-byte store (and one 16-byte bias / residual load) instead of four 4-byte ones.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

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

// XCD-aware (frame, tile) order for kernels whose workgroups are tiles of independent frames and read a halo of their neighbours' input.  The dispatcher
// places workgroup b of a 1-D grid on XCD b % 8 and every XCD has its own L2: with the plain order the tiles of one frame are spread over all eight L2s and
// every halo line is fetched from HBM once per XCD that touches it (PMC: prep_fused_k fetched 147 MB for 94 MB of touched pixels).  Here XCD k walks frames
// k, k + 8, k + 16, … tile by tile, so a frame's tiles — and their shared halo lines — meet in ONE L2.  Bijective for any frame count (the last n % 8 frames
// keep the plain order); speed only, never correctness.
__device__ __forceinline__ void xcd_frame_tile(unsigned tiles, unsigned n_frames, unsigned* frame, unsigned* tile) {
  const unsigned id = blockIdx.x, full = (n_frames & ~7u) * tiles;
  if (id < full) {
    const unsigned xcd = id & 7u, local = id >> 3, g = local / tiles;
    *frame = 8u * g + xcd;
    *tile = local - g * tiles;
  } else {
    const unsigned r = id - full, f = r / tiles;
    *frame = (n_frames & ~7u) + f;
    *tile = r - f * tiles;
  }
}

}  // namespace bsx
