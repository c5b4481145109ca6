// kernels_img.hip — integer/byte image kernels of the hot path for gfx950 (wave64).
//
// Every kernel here is bit-exact against the CPU oracle's restatement of the OpenCV 8-bit
// paths and of the reference's own loops; all are HBM-bound byte work, so the design rules
// are: 16-byte vector accesses per lane, LDS tiles for neighbourhood ops, no GEMM shapes.
//
//   prep_fused_k         libbackscrub.cc:285-302  ROI crop + cv::resize(INTER_LINEAR) + BGR2RGB + cv::bilateralFilter(5,100,100) + convertTo(CV_32FC3)
//   decode_k             libbackscrub.cc:317-357  argmax / threshold / softmax-2 + temporal IIR
//   mask_upscale_blur_k  libbackscrub.cc:367-371  cv::resize ↑ + cv::blur 5x5 into the persistent mask ROI
//   blend4x4_k           deepseg.cc:108-134       alpha_blend
//   resize_bgr_k         background.cc:186,190    cv::resize of the background
//   yuyv_k               deepseg.cc:87-106        convert_rgb_to_yuyv
#include "debug_switches.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "kernels.hpp"
#include "mfma_tile.hpp"

namespace bsx {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxGridY = 65535;
inline unsigned blocks_for(long total) { return (unsigned)((total + kThreads - 1) / kThreads); }

__device__ __forceinline__ int reflect101(int p, int len) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// One INTER_LINEAR sample of an interleaved u8 image at destination (dx,dy), channel c.
// Horizontal pass in int32 with 11-bit coefficients, vertical pass
// (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2  — OpenCV resize.cpp 8u fixed point.
template <int CN>
__device__ __forceinline__ void sample_linear(const uint8_t* __restrict__ src, long sstride, const ResizeTab& t, int dx, int dy, int* out) {
  if (t.mode == 1) {
    const uint8_t* p = src + (long)dy * sstride + (long)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) out[c] = p[c];
    return;
  }
  if (t.mode == 2) {  // INTER_AREA 2x2 (both scales exactly 2): (s00+s01+s10+s11+2)>>2
    const uint8_t* p = src + (long)(2 * dy) * sstride + (long)(2 * dx) * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) out[c] = (p[c] + p[CN + c] + p[sstride + c] + p[sstride + CN + c] + 2) >> 2;
    return;
  }
  int sx = t.xofs[dx], sx1 = min(sx + 1, t.sw - 1);
  int a0 = t.xa[2 * dx], a1 = t.xa[2 * dx + 1];
  int sy = t.yofs[dy];
  int sy0 = min(max(sy, 0), t.sh - 1), sy1 = min(max(sy + 1, 0), t.sh - 1);
  int b0 = t.ya[2 * dy], b1 = t.ya[2 * dy + 1];
  const uint8_t* r0 = src + (long)sy0 * sstride;
  const uint8_t* r1 = src + (long)sy1 * sstride;
  if constexpr (CN == 3) {
    // both taps of a row are 6 consecutive bytes: ONE (unaligned) 8-byte load per row instead of six byte loads — the texture
    // address unit, not HBM, was the limit of the byte form.  Only when the 8 bytes stay inside the row.
    if (sx1 == sx + 1 && sx * 3 + 8 <= (int)sstride) {
      struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
      const uint64_t q0 = reinterpret_cast<const U8*>(r0 + sx * 3)->v, q1 = reinterpret_cast<const U8*>(r1 + sx * 3)->v;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int h0 = (int)((q0 >> (8 * c)) & 255) * a0 + (int)((q0 >> (8 * (3 + c))) & 255) * a1;
        const int h1 = (int)((q1 >> (8 * c)) & 255) * a0 + (int)((q1 >> (8 * (3 + c))) & 255) * a1;
        out[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < CN; c++) {
    int h0 = r0[sx * CN + c] * a0 + r0[sx1 * CN + c] * a1;
    int h1 = r1[sx * CN + c] * a0 + r1[sx1 * CN + c] * a1;
    out[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  }
}

// ---- prep: bilateral d=5 constants (libbackscrub.cc:295-302).  f32 accumulation in tap order with separate multiply and add (no FMA contraction), then
// cvRound(sum * (1/wsum)) — the association the oracle defines.
constexpr int kBilPix = 4;   // pixels per lane: amortises the 768-entry LUT fill + barrier of every workgroup
// the 13 taps of a radius-2 disc in OpenCV's (dy, dx) row-major order; bsx_api.hip builds bp.space_w in the same order
// and refuses to start if its table ever disagrees (bilateral_taps_match)
__device__ constexpr int kTapY[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
__device__ constexpr int kTapX[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
// OUT bit 0: the network input as f32 [n][inH][inW][3] = convertTo (libbackscrub.cc:302) — what stems without a byte path and the stage tests read;
// bit 1: the filtered pixel itself, R | G<<8 | B<<16 [n][inH][inW] u32 — the stems that take it (seg_head_k, dl_head0_k) apply the SAME two roundings
// `fadd(fmul(float(q), scale), offset)` when they stage their input window, so the 12 B/px tensor never exists (bit-identical by construction).
// ---- YUYV -> BGR inside the kernels that read the camera frame (BSX_STEP_YUYV_IN: cv::COLOR_YUV2BGR_YUYV, app/deepseg.cc:553,725) --------------------------------
// The same integers as yuyv_to_bgr_k below (BT.601 limited range, 20-bit fixed point): a step that takes the camera's raw 2 B/px never materialises
// the 3 B/px BGR frame.  Every product is a full-rate 24-bit multiply (|c| < 2^23, |operand| <= 255).
constexpr int kYuvSH = 20, kYuvCY = 1220542, kYuvCUB = 2116026, kYuvCUG = -409993, kYuvCVG = -852492, kYuvCVR = 1673527;
// Saturating pack: sat_u8(lo >> 20) | sat_u8(hi >> 20) << 8 in ONE instruction (gfx950's v_ashr_pk_u8_i32) — spelled out, with its upper half masked, because the
// compiler's own use of the instruction is wrong on this toolchain (ROCm 7.2, clang 22): it forms it from `min(max(x >> 20, 0), 255)` pairs and then ORs further
// bytes into the result as if bits 31:16 were zero, while the hardware leaves whatever the destination register held there (found by tools/dbg_conv.hip: bytes 2-3
// of every word built that way carried bits of the unshifted sums).  Through the asm the compiler cannot see what the value is, so the mask stays.
__device__ __forceinline__ uint32_t sat_pk2_shr20(int lo, int hi) {
  uint32_t d;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "=v"(d) : "v"(lo), "v"(hi));
  return d;                                                      // bits 15:0 valid; callers mask or shift the rest away
}
__device__ __forceinline__ void yuv_chroma(int u, int v, int* buv, int* guv, int* ruv) {      // u, v already minus 128
  *ruv = (1 << (kYuvSH - 1)) + __mul24(kYuvCVR, v);
  *guv = (1 << (kYuvSH - 1)) + __mul24(kYuvCVG, v) + __mul24(kYuvCUG, u);
  *buv = (1 << (kYuvSH - 1)) + __mul24(kYuvCUB, u);
}
__device__ __forceinline__ int yuv_luma(uint32_t y) { return __mul24(max(0, (int)y - 16), kYuvCY); }
// one tap of prep_fused_k: Y | U << 8 | V << 16 (pulled out of the 8-byte window by v_perm_b32) -> B | G << 8 | R << 16
__device__ __forceinline__ uint32_t yuv_tap_to_bgr(uint32_t t) {
  int buv, guv, ruv;
  yuv_chroma((int)((t >> 8) & 255u) - 128, (int)((t >> 16) & 255u) - 128, &buv, &guv, &ruv);
  const int ya = yuv_luma(t & 255u);
  return (sat_pk2_shr20(ya + buv, ya + guv) & 0xffffu) | ((sat_pk2_shr20(ya + ruv, 0) & 0xffu) << 16);
}
// four pixels = two YUYV words (Y0 U Y1 V) -> the three words of four packed BGR pixels (the operand form of blend_quad): six saturating packs, one per byte pair
// of the output — B0 G0 | R0 B1 | G1 R1 | B2 G2 | R2 B3 | G3 R3
__device__ __forceinline__ void yuyv4_to_bgr3(uint32_t p0, uint32_t p1, uint32_t o[3]) {
  int bu0, gu0, ru0, bu1, gu1, ru1;
  yuv_chroma((int)((p0 >> 8) & 255u) - 128, (int)(p0 >> 24) - 128, &bu0, &gu0, &ru0);
  yuv_chroma((int)((p1 >> 8) & 255u) - 128, (int)(p1 >> 24) - 128, &bu1, &gu1, &ru1);
  const int y0 = yuv_luma(p0 & 255u), y1 = yuv_luma((p0 >> 16) & 255u), y2 = yuv_luma(p1 & 255u), y3 = yuv_luma((p1 >> 16) & 255u);
  o[0] = (sat_pk2_shr20(y0 + bu0, y0 + gu0) & 0xffffu) | (sat_pk2_shr20(y0 + ru0, y1 + bu0) << 16);
  o[1] = (sat_pk2_shr20(y1 + gu0, y1 + ru0) & 0xffffu) | (sat_pk2_shr20(y2 + bu1, y2 + gu1) << 16);
  o[2] = (sat_pk2_shr20(y2 + ru1, y3 + bu1) & 0xffffu) | (sat_pk2_shr20(y3 + gu1, y3 + ru1) << 16);
}

// ---- prep, both steps in ONE kernel: the workgroup resizes the (TW + 4) x (TH + 4) canvas pixels its bilateral tile reads straight into LDS -------
// (every sample of a lane requested before its first use: <= 6 independent tap pairs in flight per lane), filters from LDS and writes the network input.  The
// two-kernel form of rounds 1-2 (resize into a 4 B/px canvas with a stored BORDER_REFLECT_101 apron, then the bilateral reading it back: one write and one read of
// the canvas, a launch boundary, lanes that each waited for one dependent pair of loads) was bit-identical to this and was deleted in round 6; the halo is
// recomputed (1.27x the samples of a 32 x 32 tile, L2 hits).  The stage-0 tests read its output.  Tile sizes are chosen per model so that the tiles cover the
// canvas without a sliver (257 = 9 x 29, not 8 x 32 + 1).
constexpr int kPfS = 36;                    // LDS row stride of the tile (TW <= 32)
// YIN (BSX_STEP_YUYV_IN, LINEAR only): `frames` holds YUYV 4:2:2 (Y0 U Y1 V per pixel pair, 2 B/px) — the two taps of a sample are converted with
// cv::COLOR_YUV2BGR_YUYV's integers (yuv_tap_to_bgr) as they leave the 8-byte window, then resized exactly as BGR taps are: the same network input, bit for bit,
// as bsx_yuyv_to_bgr followed by the BGR step, without the 3 B/px frame in between.  The ROI must start on an even column (a macropixel).
template <int OUT, bool LINEAR, bool YIN = false>             // LINEAR: tab.mode == 0 (cv::resize INTER_LINEAR proper — the other modes are a copy and the exact 2x2 area average)
__global__ __launch_bounds__(kThreads) void prep_fused_k(const uint8_t* __restrict__ frames, int W, int H, Rect4 roi, float* __restrict__ input, uint32_t* __restrict__ input_u8,
                                                        int inW, int inH, Rect4 q, ResizeTab tab, BilateralParams bp, int TW, int TH, int ntx, int nty, int n_frames) {
  __shared__ float lut[768];
  __shared__ uint32_t tile[kPfS * kPfS];
  const int tid = threadIdx.x;
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(ntx * nty), (unsigned)n_frames, &f_, &t_);      // a frame's tiles on ONE XCD: their shared halo lines are L2 hits
  const long n = f_;
  const int tby = (int)t_ / ntx, tbx = (int)t_ - tby * ntx;
  const int tx0 = tbx * TW, ty0 = tby * TH;
  const int SW = TW + 2 * kCanvasPad, total = SW * (TH + 2 * kCanvasPad);
  constexpr int FB = YIN ? 2 : 3;                                         // bytes per frame pixel
  const uint8_t* src = frames + n * (long)W * H * FB + ((long)roi.y * W + roi.x) * FB;
  const unsigned msw = 0xFFFFFFFFu / (unsigned)SW + 1u;                    // i / SW for i < 2^16
  constexpr int kItems = (kPfS * kPfS + kThreads - 1) / kThreads;
  // the colour-weight table of the bilateral filter: requested with the kernel's first loads (round 5) — staged where it is first used, behind the resize phase, its
  // three loads per lane were one more memory round trip in front of the second barrier
  float lut_v[3];
#pragma unroll
  for (int k = 0; k < 3; k++) lut_v[k] = bp.color_lut[tid + k * kThreads];
  if constexpr (LINEAR) {
    // INTER_LINEAR: everything that depends only on the tile COLUMN (reflected canvas x → source byte offset, coefficient pair, where the two taps sit inside the
    // 8 bytes loaded) or only on the tile ROW (source row offsets, coefficient pair) is worked out once per column / row by the first lanes and kept in LDS; an item
    // is then two table reads, four loads and arithmetic.  Two dependent memory round trips per LANE (tables, then all of its <= kItems samples at once) instead of
    // two per SAMPLE — in the per-sample form (sample_linear inside `if (inside)`) the compiler waits for each sample before it starts the next.
    // 8 source bytes [offc, offc + 8) cover both taps: offc = min(3 sx, row_bytes - 8) never reads past the image row; the taps are bytes s0.. and s1.. of them
    // (s1 = s0 + 3, or s0 where cv::resize clamps the second tap onto the first), pulled out by v_perm_b32 with per-column selectors.  Same integers as sample_linear.
    __shared__ int4 colT[kPfS], rowT[kPfS];                                // {offc | -1, a0 | a1 << 16, sel0, sel1}, {o0 | -1, o1, b0, b1}
    const int rowlim = (W - roi.x) * FB, SHt = TH + 2 * kCanvasPad;
    if (tid < SW) {
      const int dx = reflect101(tx0 + tid - kCanvasPad, inW) - q.x;
      int4 e = make_int4(-1, 0, 0, 0);
      if (dx >= 0 && dx < q.w) {
        const int sx = tab.xofs[dx], same = sx + 1 > tab.sw - 1;
        const int a0 = tab.xa[2 * dx], a1 = tab.xa[2 * dx + 1];
        if constexpr (YIN) {
          // the 8 bytes from the macropixel of tap 0 hold both taps' macropixels (tap 1 = pixel sx + 1 sits in the same or in the next one; the window is pulled
          // back by 4 where it would pass the row end — tap 1 is then in tap 0's macropixel).  Selector of a tap: its Y, its macropixel's U and V.
          const int p1 = same ? sx : sx + 1, mb = (sx >> 1) * 4, offc = max(min(mb, rowlim - 8), 0), m0 = mb - offc, m1 = (p1 >> 1) * 4 - offc;
          const int y0 = m0 + 2 * (sx & 1), y1 = m1 + 2 * (p1 & 1);
          e = make_int4(offc, (a0 & 0xffff) | (a1 << 16), 0x0c000000 | ((m0 + 3) << 16) | ((m0 + 1) << 8) | y0, 0x0c000000 | ((m1 + 3) << 16) | ((m1 + 1) << 8) | y1);
        } else {
          const int offb = sx * 3, offc = max(min(offb, rowlim - 8), 0), s0 = offb - offc, s1 = same ? s0 : s0 + 3;
          e = make_int4(offc, (a0 & 0xffff) | (a1 << 16), 0x0c000000 | ((s0 + 2) << 16) | ((s0 + 1) << 8) | s0, 0x0c000000 | ((s1 + 2) << 16) | ((s1 + 1) << 8) | s1);
        }
      }
      colT[tid] = e;
    } else if (tid >= 64 && tid < 64 + SHt) {
      const int ly = tid - 64, dy = reflect101(ty0 + ly - kCanvasPad, inH) - q.y;
      int4 e = make_int4(-1, 0, 0, 0);
      if (dy >= 0 && dy < q.h) {
        const int sy = tab.yofs[dy], sy0 = min(max(sy, 0), tab.sh - 1), sy1 = min(max(sy + 1, 0), tab.sh - 1);
        e = make_int4(sy0 * W * FB, sy1 * W * FB, tab.ya[2 * dy], tab.ya[2 * dy + 1]);
      }
      rowT[ly] = e;
    }
    __syncthreads();
    uint32_t lo0[kItems], hi0[kItems], lo1[kItems], hi1[kItems];
#pragma unroll
    for (int k = 0; k < kItems; k++) {                                     // every load of the lane is requested here
      const int i = min(tid + k * kThreads, total - 1), ly = (int)__umulhi((unsigned)i, msw), lx = i - ly * SW;
      const int co = max(colT[lx].x, 0), o0 = max(rowT[ly].x, 0), o1 = rowT[ly].y;
      struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };            // ONE 8-byte load per source row (byte-aligned: global_load_dwordx2), not two 4-byte ones
      const uint64_t q0 = reinterpret_cast<const U8*>(src + (unsigned)(o0 + co))->v, q1 = reinterpret_cast<const U8*>(src + (unsigned)(o1 + co))->v;
      lo0[k] = (uint32_t)q0; hi0[k] = (uint32_t)(q0 >> 32);
      lo1[k] = (uint32_t)q1; hi1[k] = (uint32_t)(q1 >> 32);
    }
#pragma unroll
    for (int k = 0; k < kItems; k++) {
      const int i = tid + k * kThreads;
      if (i < total) {
        const int ly = (int)__umulhi((unsigned)i, msw), lx = i - ly * SW;
        const int4 c = colT[lx], r = rowT[ly];
        const int a0 = (short)(c.y & 0xffff), a1 = c.y >> 16, b0 = r.z, b1 = r.w;
        uint32_t t00 = __builtin_amdgcn_perm(hi0[k], lo0[k], (uint32_t)c.z), t01 = __builtin_amdgcn_perm(hi0[k], lo0[k], (uint32_t)c.w);   // row 0: tap 0 / tap 1 as B | G << 8 | R << 16
        uint32_t t10 = __builtin_amdgcn_perm(hi1[k], lo1[k], (uint32_t)c.z), t11 = __builtin_amdgcn_perm(hi1[k], lo1[k], (uint32_t)c.w);   // row 1
        if constexpr (YIN) { t00 = yuv_tap_to_bgr(t00); t01 = yuv_tap_to_bgr(t01); t10 = yuv_tap_to_bgr(t10); t11 = yuv_tap_to_bgr(t11); }      // (the taps arrived as Y | U << 8 | V << 16)
        uint32_t v = 0;                                                    // the model canvas outside in_roi (the bars) is 0
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          const int h0 = (int)((t00 >> (8 * ch)) & 255u) * a0 + (int)((t01 >> (8 * ch)) & 255u) * a1;
          const int h1 = (int)((t10 >> (8 * ch)) & 255u) * a0 + (int)((t11 >> (8 * ch)) & 255u) * a1;
          const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
          v |= (uint32_t)o << (8 * (2 - ch));                              // BGR2RGB
        }
        tile[ly * kPfS + lx] = (c.x >= 0 && r.x >= 0) ? v : 0u;
      }
    }
  } else
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const int i = tid + k * kThreads;
    if (i < total) {
      const int ly = (int)__umulhi((unsigned)i, msw), lx = i - ly * SW;
      const int dx = reflect101(tx0 + lx - kCanvasPad, inW) - q.x, dy = reflect101(ty0 + ly - kCanvasPad, inH) - q.y;
      uint32_t v = 0;                                                      // the model canvas outside in_roi (the bars) is 0
      if (dx >= 0 && dx < q.w && dy >= 0 && dy < q.h) {
        int bgr[3];
        sample_linear<3>(src, (long)W * 3, tab, dx, dy, bgr);
        v = (uint32_t)bgr[2] | ((uint32_t)bgr[1] << 8) | ((uint32_t)bgr[0] << 16);  // BGR2RGB
      }
      tile[ly * kPfS + lx] = v;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) lut[tid + k * kThreads] = lut_v[k];
  __syncthreads();
  const int lx = tid & 31, x = tx0 + lx;
  if (lx >= TW || x >= inW) return;
  // two pixels of the lane (rows ly and ly + 8) per pass, their sums in the two halves of packed registers: v_pk_mul_f32 / v_pk_add_f32 do both pixels' multiply
  // (add) of a channel in one instruction — 9 instead of 13 VALU instructions per pixel and tap, the same IEEE operations in the same order (no contraction).
  typedef float f2 __attribute__((ext_vector_type(2)));
  static_assert(kBilPix % 2 == 0, "pixel pairs");
#pragma unroll 1
  for (int it = 0; it < kBilPix; it += 2) {
    const int lyA = (tid >> 5) + 8 * it, lyB = lyA + 8;                   // row B may lie outside the tile: it reads rows < kPfS of the LDS tile and is not stored
    if (lyA >= TH || ty0 + lyA >= inH) return;
    const uint32_t* ta = tile + (lyA + kCanvasPad) * kPfS + (lx + kCanvasPad);      // the centre pixel; tap (dy, dx) at ta[dy * kPfS + dx]
    const uint32_t* tb = ta + 8 * kPfS;
    const uint32_t cA0 = ta[0], cB0 = tb[0];
    f2 sr = {0.f, 0.f}, sg = {0.f, 0.f}, sb = {0.f, 0.f}, ws = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 13; k++) {
      const uint32_t cA = ta[kTapY[k] * kPfS + kTapX[k]], cB = tb[kTapY[k] * kPfS + kTapX[k]];
      f2 w = {lut[__builtin_amdgcn_sad_u8(cA, cA0, 0u)], lut[__builtin_amdgcn_sad_u8(cB, cB0, 0u)]};
      w = w * (f2)(bp.space_w[k]);
      const f2 rr = {(float)(cA & 255), (float)(cB & 255)}, gg = {(float)((cA >> 8) & 255), (float)((cB >> 8) & 255)}, bb = {(float)((cA >> 16) & 255), (float)((cB >> 16) & 255)};
      sr = sr + rr * w;
      sg = sg + gg * w;
      sb = sb + bb * w;
      ws = ws + w;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int y = ty0 + lyA + 8 * h;
      if (h == 1 && (lyB >= TH || y >= inH)) break;
      const unsigned p = (unsigned)(y * inW + x);
      const float wi = __fdiv_rn(1.f, ws[h]);
      int qr = __float2int_rn(__fmul_rn(sr[h], wi)), qg = __float2int_rn(__fmul_rn(sg[h], wi)), qb = __float2int_rn(__fmul_rn(sb[h], wi));
      qr = min(max(qr, 0), 255); qg = min(max(qg, 0), 255); qb = min(max(qb, 0), 255);
      if (OUT & 1) {
        float* o = input + (n * (long)inW * inH + p) * 3;
        o[0] = __fadd_rn(__fmul_rn((float)qr, bp.scale), bp.offset);
        o[1] = __fadd_rn(__fmul_rn((float)qg, bp.scale), bp.offset);
        o[2] = __fadd_rn(__fmul_rn((float)qb, bp.scale), bp.offset);
      }
      if (OUT & 2) input_u8[n * (long)inW * inH + p] = (uint32_t)qr | ((uint32_t)qg << 8) | ((uint32_t)qb << 16);
    }
  }
}

// ---- decode + temporal IIR -------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void decode_k(int type, const float* __restrict__ t, uint8_t* __restrict__ out, long total, int nch) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  uint8_t val = 255;
  if (type == 1) {  // DeepLab: first maximum wins, start value -10000, "person" = 15
    const float* p = t + i * nch;
    float maxval = -10000.f; int maxpos = 0;
    for (int c = 0; c < nch; c++) { float v = p[c]; if (v > maxval) { maxval = v; maxpos = c; } }
    val = maxpos == 15 ? 0 : 255;
  } else if (type == 2) {  // MLKit / BodyPix: float promoted to double against the double literal 0.65
    val = ((double)t[i] > 0.65) ? 0 : 255;
  } else {  // Meet: expf on both logits, normalise, compare (NaN from inf/inf compares false → 255)
    const float2 l = reinterpret_cast<const float2*>(t)[i];
    // Decided-by-margin fast path.  For logits in [-80, 80] (no overflow of exp or of the sum) that differ by >= 1e-4 the
    // outcome of the reference arithmetic is forced: exp is monotone with relative error < 2^-23, so e1/e0 > 1 + 9.9e-5;
    // the two correctly-rounded divisions by the same s perturb that ratio by < 2^-23 more, hence p0 < p1 strictly (and
    // symmetrically p0 > p1).  Only near-ties and out-of-range / NaN logits take the exact path below.
    const float d = l.y - l.x;
    if (fabsf(l.x) <= 80.f && fabsf(l.y) <= 80.f && fabsf(d) >= 1e-4f) {
      val = d > 0.f ? 0 : 255;
    } else {
      float e0 = (float)exp((double)l.x), e1 = (float)exp((double)l.y);  // correctly-rounded stand-in for libm expf
      float s = __fadd_rn(e0, e1);
      float p0 = __fdiv_rn(e0, s), p1 = __fdiv_rn(e1, s);
      val = p0 < p1 ? 0 : 255;
    }
  }
  out[i] = (uint8_t)((val & 0xE0) | (out[i] >> 3));
}

// Meet decode, 4 pixels per lane: two 16-byte logit loads, one 4-byte state word read-modified-written.  Same per-pixel
// arithmetic as decode_k (fast path by margin, exact path for near ties / out-of-range / NaN logits).
__device__ __forceinline__ uint32_t meet_val(float l0, float l1) {
  const float d = l1 - l0;
  if (fabsf(l0) <= 80.f && fabsf(l1) <= 80.f && fabsf(d) >= 1e-4f) return d > 0.f ? 0u : 255u;
  const float e0 = (float)exp((double)l0), e1 = (float)exp((double)l1);
  const float s = __fadd_rn(e0, e1);
  return __fdiv_rn(e0, s) < __fdiv_rn(e1, s) ? 0u : 255u;
}
__global__ __launch_bounds__(kThreads) void decode_meet4_k(const float4* __restrict__ t, uint32_t* __restrict__ out, long quads) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= quads) return;
  const float4 a = t[2 * i], b = t[2 * i + 1];               // (l0,l1) of pixels 0,1 | 2,3
  const uint32_t v = meet_val(a.x, a.y) | (meet_val(a.z, a.w) << 8) | (meet_val(b.x, b.y) << 16) | (meet_val(b.z, b.w) << 24);
  const uint32_t o = out[i];
  out[i] = (v & 0xE0E0E0E0u) | ((o >> 3) & 0x1F1F1F1Fu);      // per byte: (val & 0xE0) | (out >> 3)
}

// MLKit / BodyPix decode, 4 pixels per lane: `p > 0.65` with the float promoted to double (the literal is a double, libbackscrub.cc:338)
__global__ __launch_bounds__(kThreads) void decode_thresh4_k(const float4* __restrict__ t, uint32_t* __restrict__ out, long quads) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= quads) return;
  const float4 p = t[i];
  const uint32_t v = ((double)p.x > 0.65 ? 0u : 255u) | (((double)p.y > 0.65 ? 0u : 255u) << 8) | (((double)p.z > 0.65 ? 0u : 255u) << 16) |
                     (((double)p.w > 0.65 ? 0u : 255u) << 24);
  const uint32_t o = out[i];
  out[i] = (v & 0xE0E0E0E0u) | ((o >> 3) & 0x1F1F1F1Fu);
}

// DeepLab argmax over nch interleaved classes: the 256 pixels of a workgroup are one contiguous block of 256*nch floats —
// read it with coalesced 4-byte loads into LDS, then every lane scans its own nch values (stride nch, conflict-free for odd
// nch).  A lane reading its classes straight from HBM touches 64 cache lines per load instruction.
constexpr int kArgmaxMaxCh = 32;
__global__ __launch_bounds__(kThreads) void decode_argmax_k(const float* __restrict__ t, uint8_t* __restrict__ out, long total, int nch) {
  __shared__ float tile[kThreads * kArgmaxMaxCh];
  const long p0 = (long)blockIdx.x * kThreads;
  const int valid = (int)min((long)kThreads, total - p0);
  const float* src = t + p0 * nch;
  for (int i = threadIdx.x; i < valid * nch; i += kThreads) tile[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x >= valid) return;
  const float* p = tile + threadIdx.x * nch;
  float maxval = -10000.f; int maxpos = 0;                       // libbackscrub.cc:318-332: first maximum wins
  for (int c = 0; c < nch; c++) { const float v = p[c]; if (v > maxval) { maxval = v; maxpos = c; } }
  const uint8_t val = maxpos == 15 ? 0 : 255;                    // class 15 = person
  out[p0 + threadIdx.x] = (uint8_t)((val & 0xE0) | (out[p0 + threadIdx.x] >> 3));
}

// ---- mask: upscale + 5x5 box blur, LDS tiled, separable -------------------------------------------------------------
// Tile = 128x32 output pixels per 256-lane workgroup.  All table lookups happen once per tile column / tile row:
//   1. per column of the (128+4)-wide halo tile: reflected ROI x → (sx, sx1, a0, a1);  per row of the (32+4)-tall
//      halo tile: reflected ROI y → (sy0, sy1, b0, b1)                               [LDS]
//   2. horizontal pass of cv::resize for the <= kMaxSrcRows source rows the tile touches:  hq[sy][x] = (S0*a0 + S1*a1) >> 4
//   3. vertical pass: up[y][x] = (((b0*hq[sy0][x]) >> 16) + ((b1*hq[sy1][x]) >> 16) + 2) >> 2   (exactly OpenCV's 8u formula)
//   4. horizontal 5-sums (u16), 5. vertical 5-sums, (s+12)/25, 4 pixels per 32-bit store.
// Steps 2-3 are the separable form of the per-pixel bilinear sample: identical integers, ~4x fewer operations.
// With BLEND the same workgroup also composites its tile (deepseg.cc:108-134) while the mask bytes are still in
// registers: the mask is written once and never re-read, and the HBM-bound blend traffic of some workgroups overlaps
// the LDS/ALU-bound mask phases of others.  (Used when W, roi.x and roi.w are multiples of 4; outside the ROI the composite
// is the background itself, copied by outside_roi_copy_k.)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));   // v_pk_*_u16 operand
constexpr int kTW = 128, kTH = 32, kHW = kTW + 4, kHH = kTH + 4, kMaxSrcRows = 40;
constexpr int kTileItems = kTH * (kTW / 4) / kThreads;     // 4-pixel groups per lane in the last step
static_assert(kTileItems * kThreads == kTH * (kTW / 4) && kThreads == 8 * (kTW / 4), "tile / lane mapping: item i of a lane sits 8 rows below item i-1");

// ---- packed alpha-blend arithmetic (deepseg.cc:108-134), shared by the mask tile kernels and blend4x4_k --------------------
// Packed form of the same integers (v_pk_*_u16, two bytes per instruction):  a*m + b*(255-m) <= 255*255 fits a u16 lane,
// and floor(t/255) == (t + 1 + (t >> 8)) >> 8 for every t in [0, 65025] (exhaustively checked; the sum stays < 65536).
// Round 5: u = a*m + 1 + b*(255-m) as two v_pk_mad_u16 (u <= 65026 fits a u16 lane) and floor((u-1)/255) == (u + (u >> 8)) >> 8 (exhaustive: tests/test_oracle_image.py;
// u + (u >> 8) <= 65280) — five packed instructions per two bytes instead of six; the odd bytes of a word are pulled into u16 lanes by ONE v_perm_b32 each (was shift + and).
// 67 -> 55 VALU instructions per four pixels: the fused mask + blend's general tiles run their VALU at 75 % busy (profiles/r05z_pmc_sq_lite.md).
__device__ __forceinline__ us2 pk_blend(uint32_t a, uint32_t b, uint32_t m, uint32_t im) {   // operands: two u8 values in the u16 halves; im = 0x00ff00ff ^ m = 255 - m per half
  const us2 av = __builtin_bit_cast(us2, a), bv = __builtin_bit_cast(us2, b), mv = __builtin_bit_cast(us2, m), iv = __builtin_bit_cast(us2, im);
  // (written as `av * mv + one` the compiler moves the constant to the end of the sum: multiply, multiply-add, add — the first fused form is spelled out)
  uint32_t u0;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(u0) : "v"(a), "v"(m), "s"(0x00010001u));
  (void)av; (void)mv;
  us2 u = bv * iv + __builtin_bit_cast(us2, u0);
  return (u + (u >> 8)) >> 8;
}
__device__ __forceinline__ uint32_t blend_word(uint32_t a, uint32_t b, uint32_t m02, uint32_t m13) {
  const uint32_t K = 0x00ff00ffu;
  const uint32_t r02 = __builtin_bit_cast(uint32_t, pk_blend(a & K, b & K, m02, m02 ^ K));
  const uint32_t r13 = __builtin_bit_cast(uint32_t, pk_blend(__builtin_amdgcn_perm(a, a, 0x0c030c01u), __builtin_amdgcn_perm(b, b, 0x0c030c01u), m13, m13 ^ K));
  return r02 | (r13 << 8);
}
typedef uint32_t u3v __attribute__((ext_vector_type(3)));      // four packed BGR pixels: one global_{load,store}_dwordx3 (4-byte aligned)
// 4 pixels = 12 bytes = 3 words; mw holds their 4 mask bytes.  Byte→pixel map of the words: (0,0,0,1) (1,1,2,2) (2,3,3,3).
__device__ __forceinline__ void blend_quad(const uint32_t a[3], const uint32_t b[3], uint32_t mw, uint32_t o[3]) {
  const uint32_t m00 = __builtin_amdgcn_perm(mw, mw, 0x0c000c00u), m01 = __builtin_amdgcn_perm(mw, mw, 0x0c010c00u);
  const uint32_t m12 = __builtin_amdgcn_perm(mw, mw, 0x0c020c01u);
  const uint32_t m23 = __builtin_amdgcn_perm(mw, mw, 0x0c030c02u), m33 = __builtin_amdgcn_perm(mw, mw, 0x0c030c03u);
  o[0] = blend_word(a[0], b[0], m00, m01);
  o[1] = blend_word(a[1], b[1], m12, m12);
  o[2] = blend_word(a[2], b[2], m23, m33);
}

// ---- BGR → YUYV (cv::cvtColor(COLOR_RGB2YUV) on BGR-ordered bytes, then 4:2:2 pack Y0 V Y1 U) -----
__device__ __forceinline__ void rgb2yuv(int R, int G, int B, int* Y, int* U, int* V) {
  const int shift = 14, half = 1 << 13, delta = 128 << 14;
  int y = (R * 4899 + G * 9617 + B * 1868 + half) >> shift;
  int u = ((B - y) * 8061 + delta + half) >> shift;
  int v = ((R - y) * 14369 + delta + half) >> shift;
  *Y = min(max(y, 0), 255); *U = min(max(u, 0), 255); *V = min(max(v, 0), 255);
}
// two neighbouring pixels (bytes as stored: channel 0 is taken as "R", deepseg.cc:90) → one YUYV word Y0 | V<<8 | Y1<<16 | U<<24
__device__ __forceinline__ uint32_t yuyv_pair(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2) {
  int y0, u0, v0, y1, u1, v1;
  rgb2yuv((int)a0, (int)a1, (int)a2, &y0, &u0, &v0);
  rgb2yuv((int)b0, (int)b1, (int)b2, &y1, &u1, &v1);
  return (uint32_t)y0 | ((uint32_t)((v0 + v1) / 2) << 8) | ((uint32_t)y1 << 16) | ((uint32_t)((u0 + u1) / 2) << 24);
}

// ---- pieces shared by the two mask tile kernels ------------------------------------------------------------------------------
// Composite operands of a lane's kTileItems 4-pixel groups (12 B of background + 12 B of frame each), requested at the very
// top of the kernel so that their HBM latency hides behind the LDS phases.
struct TileBlendOperands { uint32_t a[kTileItems][3], b[kTileItems][3]; };
// `uniform` (wave-uniform): 0 = both operands; 1 = the tile's mask is 255 everywhere → only the background is needed; 2 = 0 everywhere → only the frame
// `parts` (wave-uniform): bit 0 = request the background operand, bit 1 = the frame operand — 3 = whatever `uniform` needs in one call; mask_tile_k requests a SHARED
// background before it knows the tile's class (parts = 1) and the frame after (parts = 2)
// `yin` (wave-uniform; BSX_STEP_YUYV_IN): `frames` holds YUYV 4:2:2 — a lane's four pixels are 8 bytes (b[i][0..1]), converted where they are consumed
// (tile_frame_bgr below); b[i][2] is then unused
template <bool BLEND, bool WHOLE = false>      // WHOLE: every tile of the launch lies inside the ROI (roi.w % 128 == 0, roi.h % 32 == 0): no per-item edge tests
__device__ __forceinline__ void tile_load_blend_operands(TileBlendOperands& o, const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ frames,
                                                         int n, int W, int H, Rect4 roi, int tx0, int ty0, int tid, int uniform = 0, int parts = 3, bool yin = false) {
  if constexpr (BLEND) {
    const int ly0 = tid / (kTW / 4), gx = tx0 + (tid % (kTW / 4)) * 4;
    const long pix0 = (long)(roi.y + ty0 + ly0) * W + roi.x + gx;        // frame coordinates of the ROI-relative tile pixel
    const int fb = yin ? 2 : 3;                                          // bytes per frame pixel
    const uint8_t* const a0 = bg + (bg_stride ? n * bg_stride : 0) + pix0 * 3;
    const uint8_t* const b0 = frames + ((long)n * W * H + pix0) * fb;
#pragma unroll
    for (int i = 0; i < kTileItems; i++) {
      if (parts & 1) o.a[i][0] = o.a[i][1] = o.a[i][2] = 0;
      if (parts & 2) o.b[i][0] = o.b[i][1] = o.b[i][2] = 0;
      if (WHOLE || (ty0 + ly0 + 8 * i < roi.h && gx < roi.w)) {
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(a0 + (long)(8 * i) * W * 3);
        const uint32_t* bp = reinterpret_cast<const uint32_t*>(b0 + (long)(8 * i) * W * fb);
        if ((parts & 1) && uniform != 2) { o.a[i][0] = ap[0]; o.a[i][1] = ap[1]; o.a[i][2] = ap[2]; }
        if ((parts & 2) && uniform != 1) {                               // streamed once
          o.b[i][0] = __builtin_nontemporal_load(bp); o.b[i][1] = __builtin_nontemporal_load(bp + 1);
          if (!yin) o.b[i][2] = __builtin_nontemporal_load(bp + 2);
        }
      }
    }
  }
}
// the frame operand of item i as three BGR words, whatever form it was loaded in
__device__ __forceinline__ void tile_frame_bgr(const TileBlendOperands& o, int i, bool yin, uint32_t b3[3]) {
  if (yin) yuyv4_to_bgr3(o.b[i][0], o.b[i][1], b3);
  else { b3[0] = o.b[i][0]; b3[1] = o.b[i][1]; b3[2] = o.b[i][2]; }
}
// Step 3: vertical pass of cv::resize, 4 pixels per lane (64-bit LDS reads of the two source rows, one 32-bit write).
// row0 / row1: LDS tables of the two hq rows of every halo row; (y, xg) advance without a division.
template <typename RowT>
__device__ __forceinline__ void tile_vertical_pass(const uint16_t* hq, uint8_t* up, const RowT* row0, const RowT* row1, int row_bias, const short* row_b0,
                                                   const short* row_b1, int tid) {
  constexpr int kG = kHW / 4;
  for (int y = tid / kG, xg = tid % kG; y < kHH;) {
    const int x = xg * 4;
    const int b0 = row_b0[y], b1 = row_b1[y];                  // 16-bit coefficients: the products below are 24-bit multiplies
    const uint2 r0 = *reinterpret_cast<const uint2*>(&hq[((int)row0[y] - row_bias) * kHW + x]);
    const uint2 r1 = *reinterpret_cast<const uint2*>(&hq[((int)row1[y] - row_bias) * kHW + x]);
    const int h0[4] = {(int)(r0.x & 0xffff), (int)(r0.x >> 16), (int)(r0.y & 0xffff), (int)(r0.y >> 16)};
    const int h1[4] = {(int)(r1.x & 0xffff), (int)(r1.x >> 16), (int)(r1.y & 0xffff), (int)(r1.y >> 16)};
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) packed |= (uint32_t)((((b0 * h0[j]) >> 16) + ((b1 * h1[j]) >> 16) + 2) >> 2) << (8 * j);
    *reinterpret_cast<uint32_t*>(&up[y * kHW + x]) = packed;
    xg += kThreads % kG; y += kThreads / kG;
    if (xg >= kG) { xg -= kG; y++; }
  }
}
// Step 4: horizontal 5-sums, 4 per lane: 8 consecutive bytes in, 4 u16 out.  v_sad_u8 against 0 adds the four bytes of a
// word (+ an accumulator) in one instruction; the sliding windows come from v_alignbyte.
__device__ __forceinline__ void tile_hsum5(const uint8_t* up, uint16_t* hs, int tid) {
  for (int k = tid; k < kHH * (kTW / 4); k += kThreads) {
    const int ly = k / (kTW / 4), lx = (k - ly * (kTW / 4)) * 4;
    const uint2 v = *reinterpret_cast<const uint2*>(&up[ly * kHW + lx]);            // bytes b0..b3 | b4..b7
    const uint32_t s0 = __builtin_amdgcn_sad_u8(v.x, 0u, v.y & 255u);                                                  // b0..b4
    const uint32_t s1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(v.y, v.x, 1), 0u, (v.y >> 8) & 255u);      // b1..b5
    const uint32_t s2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(v.y, v.x, 2), 0u, (v.y >> 16) & 255u);     // b2..b6
    const uint32_t s3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(v.y, v.x, 3), 0u, v.y >> 24);              // b3..b7
    *reinterpret_cast<uint2*>(&hs[ly * kTW + lx]) = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
  }
}
// Step 5: vertical 5-sums (packed u16 adds: five sums <= 25*255 stay inside a u16 lane), (s + 12) / 25 as
// ((s + 12) * 5243) >> 17 (exhaustively checked for s <= 25*255), mask store, and with BLEND the composite of the same
// 4 pixels (W, roi.x and roi.w multiples of 4, checked by the launcher: 12 bytes = 3 aligned words per image).
// four packed BGR pixels (12 bytes in three words) in reverse pixel order: cv::flip(.., 1) of a 4-pixel group
__device__ __forceinline__ void reverse4px(uint32_t (&w)[3]) {
  const uint32_t n0 = (w[2] >> 8) | ((w[1] & 0x00FF0000u) << 8);
  const uint32_t n1 = (w[1] >> 24) | ((w[2] & 255u) << 8) | ((w[0] >> 24) << 16) | (w[1] << 24);
  const uint32_t n2 = ((w[1] >> 8) & 255u) | (w[0] << 8);
  w[0] = n0; w[1] = n1; w[2] = n2;
}
// flip (yuyv bits 1-2: 2 = horizontal, 4 = vertical): cv::flip of the COMPOSITE (deepseg.cc:667-673) folded into where the tile stores it — the lane's four
// pixels go to the mirrored column group in reverse order, the row to the mirrored row; the persistent mask is the unflipped frame's and stays put.
// `uniform` (wave-uniform; see mask_tile_k): 1 / 2 = every mask byte of the tile is 255 / 0 — no sums to form, and the composite IS the background / the frame
// ((a*255 + b*0)/255 == a for every byte: the exhaustive blend test covers m = 0 and 255)
template <bool BLEND, bool WHOLE = false>
__device__ __forceinline__ void tile_vsum5_store(const uint16_t* hs, uint8_t* __restrict__ mask, uint8_t* __restrict__ outp, const TileBlendOperands& o,
                                                 int n, int W, int H, Rect4 roi, int tx0, int ty0, int tid, int yuyv_flip, int uniform = 0) {
  const int ly0 = tid / (kTW / 4), lx = (tid % (kTW / 4)) * 4;
  const int gx = tx0 + lx;
  const int yuyv = yuyv_flip & 1;
  const bool fh = (yuyv_flip & 2) != 0, fv = (yuyv_flip & 4) != 0, yin = (yuyv_flip & 16) != 0;      // bit 4 = BSX_STEP_YUYV_IN: the frame operand arrived as YUYV
  const int obpp = yuyv ? 2 : 3;                                 // composite written as packed BGR or as YUYV 4:2:2 (convert_rgb_to_yuyv fused in)
  uint8_t* const dst0 = mask + (long)n * W * H + (long)(roi.y + ty0 + ly0) * W + roi.x + gx;
  const int oy0 = fv ? H - 1 - (roi.y + ty0 + ly0) : roi.y + ty0 + ly0, ox = fh ? W - 4 - (roi.x + gx) : roi.x + gx;
  const long orow = fv ? -(long)W : (long)W;                     // output row step per tile row
  uint8_t* const out0 = BLEND ? outp + ((long)n * W * H + (long)oy0 * W + ox) * obpp : nullptr;
#pragma unroll
  for (int i = 0; i < kTileItems; i++) {
    const int ly = ly0 + 8 * i, gy = ty0 + ly;
    if (!WHOLE && (gy >= roi.h || gx >= roi.w)) continue;
    uint32_t packed = uniform == 1 ? 0xFFFFFFFFu : 0u;
    if (!uniform) {
      uint2 acc = *reinterpret_cast<const uint2*>(&hs[ly * kTW + lx]);
#pragma unroll
      for (int r = 1; r < 5; r++) {
        const uint2 v = *reinterpret_cast<const uint2*>(&hs[(ly + r) * kTW + lx]);
        acc.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, acc.x) + __builtin_bit_cast(us2, v.x));
        acc.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, acc.y) + __builtin_bit_cast(us2, v.y));
      }
      const uint32_t m0 = (__umul24(acc.x & 0xffffu, 5243u) + 12u * 5243u) >> 17, m1 = (__umul24(acc.x >> 16, 5243u) + 12u * 5243u) >> 17;
      const uint32_t m2 = (__umul24(acc.y & 0xffffu, 5243u) + 12u * 5243u) >> 17, m3 = (__umul24(acc.y >> 16, 5243u) + 12u * 5243u) >> 17;
      packed = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
    }
    uint8_t* dst = dst0 + (long)(8 * i) * W;
    if (BLEND && (yuyv_flip & 8)) { /* composite only (BSX_STEP_NO_MASK): the full-resolution mask stays in registers */ }
    else if (WHOLE || (gx + 3 < roi.w && ((uintptr_t)dst & 3) == 0)) *reinterpret_cast<uint32_t*>(dst) = packed;      // (nontemporal here: measured, no difference — profiles/r05d)
    else for (int j = 0; j < 4 && gx + j < roi.w; j++) dst[j] = (uint8_t)(packed >> (8 * j));
    if constexpr (BLEND) {
      uint32_t* op = reinterpret_cast<uint32_t*>(out0 + (long)(8 * i) * orow * obpp);
      uint32_t o3[3];
      if (uniform == 1) { o3[0] = o.a[i][0]; o3[1] = o.a[i][1]; o3[2] = o.a[i][2]; }
      else if (uniform == 2) tile_frame_bgr(o, i, yin, o3);
      else { uint32_t b3[3]; tile_frame_bgr(o, i, yin, b3); blend_quad(o.a[i], b3, packed, o3); }
      if (fh) reverse4px(o3);
      if (yuyv) {                                                  // deepseg.cc:87-106 on the four composited pixels: 8 bytes instead of 12
        __builtin_nontemporal_store(yuyv_pair(o3[0] & 255u, (o3[0] >> 8) & 255u, (o3[0] >> 16) & 255u, o3[0] >> 24, o3[1] & 255u, (o3[1] >> 8) & 255u), op);
        __builtin_nontemporal_store(yuyv_pair((o3[1] >> 16) & 255u, o3[1] >> 24, o3[2] & 255u, (o3[2] >> 8) & 255u, (o3[2] >> 16) & 255u, o3[2] >> 24), op + 1);
      } else {
        // ONE 12-byte store (global_store_dwordx3): the 32 lanes of a tile row then write 384 contiguous bytes per instruction — as three dword stores each
        // instruction wrote 4 of every 12 bytes (a third of every line, three times over)
        if (yuyv_flip & 128) *reinterpret_cast<u3v*>(op) = u3v{o3[0], o3[1], o3[2]};      // (debug build: plain instead of nontemporal stores, A/B — round 6)
        else __builtin_nontemporal_store(u3v{o3[0], o3[1], o3[2]}, reinterpret_cast<u3v*>(op));
      }
    }
  }
}

template <bool BLEND, bool YIN = false>      // YIN (BSX_STEP_YUYV_IN): `frames` is YUYV 4:2:2 — a template parameter so that the BGR instantiation carries none of the conversion
__global__ __launch_bounds__(kThreads) void mask_upscale_blur_k(const uint8_t* __restrict__ ofinal, int outW, int outH, Rect4 q, ResizeTab tab,
                                                               uint8_t* __restrict__ mask, int W, int H, Rect4 roi,
                                                               const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ frames,
                                                               uint8_t* __restrict__ outp, int yuyv, int ntx, int nty, int n_frames) {
  // coefficients are 0..2048: kept as 16-bit so that every product below is a full-rate 24-bit multiply
  __shared__ int col_sx[kHW], col_sx1[kHW], row_s0[kHH], row_s1[kHH];
  __shared__ short col_a0[kHW], col_a1[kHW], row_b0[kHH], row_b1[kHH];
  // hq (steps 2-3) and hs (steps 4-5) are never live together: one buffer, more workgroups per CU
  __shared__ __attribute__((aligned(16))) uint16_t hq_hs[kMaxSrcRows * kHW > kHH * kTW ? kMaxSrcRows * kHW : kHH * kTW];
  __shared__ __attribute__((aligned(16))) uint8_t up[kHH * kHW + 8];
  uint16_t* const hq = hq_hs;
  uint16_t* const hs = hq_hs;
  __shared__ int s_min, s_max;
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(ntx * nty), (unsigned)n_frames, &f_, &t_);
  const int n = (int)f_, tby = (int)t_ / ntx, tbx = (int)t_ - tby * ntx;
  const int tx0 = tbx * kTW, ty0 = tby * kTH;
  const uint8_t* src = ofinal + (long)n * outW * outH + (long)q.y * outW + q.x;
  const int tid = threadIdx.x;
  TileBlendOperands ops;
  yuyv = YIN ? (yuyv | 16) : (yuyv & ~16);                                  // the helpers read bit 4: a compile-time constant per instantiation
  tile_load_blend_operands<BLEND>(ops, bg, bg_stride, frames, n, W, H, roi, tx0, ty0, tid, 0, 3, YIN);
  if (tid == 0) { s_min = 1 << 30; s_max = -1; }
  __syncthreads();
  // 1. column / row tables
  if (tid < kHW) {
    const int gx = reflect101(min(tx0 + tid - 2, roi.w + 1), roi.w);
    int sx, sx1, a0, a1;
    if (tab.mode == 1) { sx = sx1 = gx; a0 = 2048; a1 = 0; }
    else if (tab.mode == 2) { sx = 2 * gx; sx1 = 2 * gx + 1; a0 = a1 = 0; }
    else { sx = tab.xofs[gx]; sx1 = min(sx + 1, tab.sw - 1); a0 = tab.xa[2 * gx]; a1 = tab.xa[2 * gx + 1]; }
    col_sx[tid] = sx; col_sx1[tid] = sx1; col_a0[tid] = (short)a0; col_a1[tid] = (short)a1;
  } else if (tid >= 192 && tid < 192 + kHH) {
    const int r = tid - 192;
    const int gy = reflect101(min(ty0 + r - 2, roi.h + 1), roi.h);
    int s0, s1, b0, b1;
    if (tab.mode == 1) { s0 = s1 = gy; b0 = 2048; b1 = 0; }
    else if (tab.mode == 2) { s0 = 2 * gy; s1 = 2 * gy + 1; b0 = b1 = 0; }
    else { const int sy = tab.yofs[gy]; s0 = min(max(sy, 0), tab.sh - 1); s1 = min(max(sy + 1, 0), tab.sh - 1); b0 = tab.ya[2 * gy]; b1 = tab.ya[2 * gy + 1]; }
    row_s0[r] = s0; row_s1[r] = s1; row_b0[r] = (short)b0; row_b1[r] = (short)b1;
    atomicMin(&s_min, s0);
    atomicMax(&s_max, s1);
  }
  __syncthreads();
  const int smin = s_min, nsr = s_max - smin + 1;
  if (tab.mode == 0 && nsr <= kMaxSrcRows) {
    // 2. horizontal pass on the touched source rows.  A lane keeps ONE column (its table entries stay in registers) and
    //    walks the source rows — columns 0..127 two rows per pass, the four halo columns 128..131 by the first 4*nsr
    //    lanes.  No index division, 32-bit offsets from a uniform base, 24-bit multiplies.
    static_assert(kThreads == 2 * kTW && 4 * kMaxSrcRows <= kThreads, "lane mapping of step 2");
    {
      const uint8_t* const base = src + (long)smin * outW;
      const int x = tid & (kTW - 1), half = tid >> 7;
      const int a0 = col_a0[x], a1 = col_a1[x];
      unsigned o0 = (unsigned)(half * outW + col_sx[x]), o1 = (unsigned)(half * outW + col_sx1[x]);
      for (int r = half; r < nsr; r += 2, o0 += 2u * outW, o1 += 2u * outW)
        hq[r * kHW + x] = (uint16_t)((base[o0] * a0 + base[o1] * a1) >> 4);
      if (tid < 4 * nsr) {
        const int xr = kTW + (tid & 3), r = tid >> 2;
        hq[r * kHW + xr] = (uint16_t)((base[(unsigned)(r * outW + col_sx[xr])] * col_a0[xr] + base[(unsigned)(r * outW + col_sx1[xr])] * col_a1[xr]) >> 4);
      }
    }
    __syncthreads();
    tile_vertical_pass(hq, up, row_s0, row_s1, smin, row_b0, row_b1, tid);   // 3.
  } else {
    // copy / exact-2x area / very strong down-scale: direct per-pixel sample
    for (int k = tid; k < kHH * kHW; k += kThreads) {
      const int y = k / kHW, x = k - y * kHW;
      int v;
      if (tab.mode == 1) v = src[(long)row_s0[y] * outW + col_sx[x]];
      else if (tab.mode == 2) v = (src[(long)row_s0[y] * outW + col_sx[x]] + src[(long)row_s0[y] * outW + col_sx1[x]] + src[(long)row_s1[y] * outW + col_sx[x]] +
                                   src[(long)row_s1[y] * outW + col_sx1[x]] + 2) >> 2;
      else {
        const int h0 = src[(long)row_s0[y] * outW + col_sx[x]] * col_a0[x] + src[(long)row_s0[y] * outW + col_sx1[x]] * col_a1[x];
        const int h1 = src[(long)row_s1[y] * outW + col_sx[x]] * col_a0[x] + src[(long)row_s1[y] * outW + col_sx1[x]] * col_a1[x];
        v = (((row_b0[y] * (h0 >> 4)) >> 16) + ((row_b1[y] * (h1 >> 4)) >> 16) + 2) >> 2;
      }
      up[k] = (uint8_t)v;
    }
  }
  __syncthreads();
  tile_hsum5(up, hs, tid);                                                                       // 4.
  __syncthreads();
  tile_vsum5_store<BLEND>(hs, mask, outp, ops, n, W, H, roi, tx0, ty0, tid, yuyv);               // 5.
}

// ---- mask tile, single-round-trip form ---------------------------------------------------------------------------------
// Same integers as mask_upscale_blur_k, restructured around what actually bounds it: with the composite fused in, a
// workgroup's life is a chain of dependent memory round trips taken while HBM is saturated by the blend traffic
// (loaded latency of several microseconds each).  Here every global load of the tile is issued in the first few
// instructions: the source-block extents come from four uniform (scalar) table reads, then the raw ofinal block, this
// lane's table entries and its composite operands are all requested back to back.  Steps 2-5 then run from LDS only.
// Used when the host verified that every tile's source block fits (ResizeTab::tile_ok); other cases take the kernel above.
constexpr int kSrcBlockBytes = kHH * kHW;        // the raw block lives in `up` until step 3 overwrites it
// F0: the launch has no flag set (no YUYV out, no flip, mask stored, default load order) — the default step.  A template parameter like YIN because this kernel pays
// for every wave-uniform branch it carries: with the YUYV-in conversion behind a run-time flag the BGR step's launch was 12-15 % slower (profiles/r06l: 105-111 ->
// 94-97 us at configs[1], 1064-1114 -> 946-955 us at the configs[4] slice), code that never ran.
// WH: whole tiles only (see tile_load_blend_operands) and a 4-byte aligned mask — decided by the launcher from the geometry
template <bool BLEND, bool YIN = false, bool F0 = false, bool WH = false>
__global__ __launch_bounds__(kThreads) void mask_tile_k(const uint8_t* __restrict__ ofinal, int outW, int outH, Rect4 q, ResizeTab tab,
                                                       uint8_t* __restrict__ mask, int W, int H, Rect4 roi,
                                                       const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ frames,
                                                       uint8_t* __restrict__ outp, int yuyv, int ntx, int nty, int n_frames, int ty_base, int nty_all) {
  // (ty_base, nty_all: this launch covers tile rows [ty_base, ty_base + nty) of the nty_all rows of a frame — the launcher cuts a frame whose last tile row is partial
  //  into the whole rows, run by the WH instantiation, and that last row)
  __shared__ short col_c0[kHW], col_c1[kHW], col_a0[kHW], col_a1[kHW];     // block-relative tap columns, coefficients
  __shared__ short row_r0[kHH], row_r1[kHH], row_b0[kHH], row_b1[kHH];     // block-relative tap rows, coefficients
  __shared__ __attribute__((aligned(16))) uint16_t hq_hs[kMaxSrcRows * kHW > kHH * kTW ? kMaxSrcRows * kHW : kHH * kTW];
  __shared__ __attribute__((aligned(16))) uint8_t up[kHH * kHW + 8];
  uint16_t* const hq = hq_hs;
  uint16_t* const hs = hq_hs;
  uint8_t* const blk = up;
  // XCD-aware order: a frame's tiles run on ONE XCD.  Where the ROI does not start on a 128-byte line (roi.x = 80 / 280: DeepLab, MLKit) the 384-byte row
  // segments of neighbouring tiles share cache lines — fetched once per L2 that touches them (PMC: 1.8x the frame bytes in the plain order).
  unsigned f_, t_;
  xcd_frame_tile((unsigned)(ntx * nty), (unsigned)n_frames, &f_, &t_);
  const int n = (int)f_, tid = threadIdx.x, tby0 = (int)t_ / ntx, tbx = (int)t_ - tby0 * ntx, tby = tby0 + ty_base;
  const int tx0 = tbx * kTW, ty0 = tby * kTH;
  // UNIFORM TILES (round 4).  In the steady state of the temporal filter the model-resolution mask is exactly 0x00 or 0xFF wherever the person's outline is not
  //     (lib/libbackscrub.cc:330-355: three equal decisions in a row), and a tile whose whole source block (taps of the halo included) holds one of those two values
  //     gets that value in every pixel: cv::resize of a constant is the constant ((b0 * 32640 >> 16) + (b1 * 32640 >> 16) + 2) >> 2 == 255 for every b0 + b1 == 2048,
  //     tests/test_oracle_image.py), and so is the box blur ((25 v + 12) / 25).  Such a tile skips steps 1-5, and its composite is a copy: of the background where the
  //     mask is 255 — the frame is not even read — and of the frame where it is 0.  tile_class_k (below) classified every tile of the launch a moment ago: one
  //     scalar load here, no vote, no barrier, nothing serialised in front of the operand loads (the first version voted inside this kernel: block load → ballot →
  //     barrier → operand loads cost the general path 6 %, profiles/r04e).  Bit-identical by construction; BSX_NO_UNIFORM_TILES=1 (read when the context is
  //     created) keeps every tile on the general path (A/B timing; the parity tests run both).
  int uniform = 0;
  // A SHARED background (bg_stride == 0: one image for all streams, L2-resident) is requested BEFORE the tile's class is known (round 5): the class byte is a second
  // dependent round trip behind the kernel arguments, and two thirds of the tiles (uniform 255: composite = background) then need nothing else.  A per-stream
  // background is HBM traffic a uniform-0 tile must not pay: it keeps the order class -> operands.
  if (F0) yuyv = 0;                                                                                 // flags as compile-time constants: F0 = none, YIN = bit 4 (read again by
  yuyv = YIN ? (yuyv | 16) : (yuyv & ~16);                                                          // tile_vsum5_store)
  const bool early_bg = BLEND && bg_stride == 0 && tab.tile_class != nullptr && !(yuyv & 64);      // (bit 6: the debug build's A/B switch for this order)
  constexpr bool yin = YIN;                                                                         // BSX_STEP_YUYV_IN
  TileBlendOperands ops;
  if (early_bg) tile_load_blend_operands<BLEND, WH>(ops, bg, bg_stride, frames, n, W, H, roi, tx0, ty0, tid, 0, 1, yin);
  if (tab.tile_class) {                                     // the aligned word that holds the byte: a SCALAR load (uniform address), not a vector load + readfirstlane
    const uintptr_t ca = (uintptr_t)tab.tile_class + (size_t)n * (size_t)(ntx * nty_all) + (size_t)(tby * ntx + tbx);
    uniform = (int)((*reinterpret_cast<const uint32_t*>(ca & ~(uintptr_t)3) >> (8 * (unsigned)(ca & 3))) & 255u);
  }
  if (uniform) {                                           // wave-uniform: nothing of the general path below is even requested
    tile_load_blend_operands<BLEND, WH>(ops, bg, bg_stride, frames, n, W, H, roi, tx0, ty0, tid, uniform, early_bg ? 2 : 3, yin);
    tile_vsum5_store<BLEND, WH>(hq_hs, mask, outp, ops, n, W, H, roi, tx0, ty0, tid, yuyv, uniform);
    return;
  }
  // extents of the source block: xofs / yofs are monotonic, so the extreme destination rows / columns give them
  // (requesting these four scalar table reads together with the class byte — one scalar round trip instead of two in front of the general path — measured: no
  //  difference on any configuration, profiles/r05e)
  const int gy_lo = max(ty0 - 2, 0), gy_hi = min(ty0 + kTH + 1, roi.h - 1);
  const int gx_lo = max(tx0 - 2, 0), gx_hi = min(tx0 + kTW + 1, roi.w - 1);
  const int smin = min(max(tab.yofs[gy_lo], 0), tab.sh - 1), smax = min(max(tab.yofs[gy_hi] + 1, 0), tab.sh - 1);
  const int cmin = tab.xofs[gx_lo], cmax = min(tab.xofs[gx_hi] + 1, tab.sw - 1);
  const int nsr = smax - smin + 1, ncol = cmax - cmin + 1;
  const uint8_t* const base = ofinal + (long)n * outW * outH + (long)(q.y + smin) * outW + q.x + cmin;
  // (a) raw block: 12 rows x 64 columns in three loads per lane cover the usual 5x up-scale; anything larger loops below
  uint32_t raw[3];
  const int br = tid >> 6, bc = tid & 63;
#pragma unroll
  for (int j = 0; j < 3; j++) { raw[j] = 0; if (br + 4 * j < nsr && bc < ncol) raw[j] = base[(unsigned)((br + 4 * j) * outW + bc)]; }
  // (b) this lane's table entries
  int t_s = 0, t_a0 = 0, t_a1 = 0;
  if (tid < kHW) {
    const int gx = reflect101(min(tx0 + tid - 2, roi.w + 1), roi.w);
    t_s = tab.xofs[gx]; t_a0 = tab.xa[2 * gx]; t_a1 = tab.xa[2 * gx + 1];
  } else if (tid >= 192 && tid < 192 + kHH) {
    const int gy = reflect101(min(ty0 + (tid - 192) - 2, roi.h + 1), roi.h);
    t_s = tab.yofs[gy]; t_a0 = tab.ya[2 * gy]; t_a1 = tab.ya[2 * gy + 1];
  }
  // (c) composite operands (the shared background is already on its way)
  tile_load_blend_operands<BLEND, WH>(ops, bg, bg_stride, frames, n, W, H, roi, tx0, ty0, tid, 0, early_bg ? 2 : 3, yin);
  // 1. block and tables into LDS
#pragma unroll
  for (int j = 0; j < 3; j++) if (br + 4 * j < nsr && bc < ncol) blk[(br + 4 * j) * ncol + bc] = (uint8_t)raw[j];
  if (nsr > 12 || ncol > 64)
    for (int r = br; r < nsr; r += 4)
      for (int cc = bc; cc < ncol; cc += 64)
        if (r >= 12 || cc >= 64) blk[r * ncol + cc] = base[(unsigned)(r * outW + cc)];
  if (tid < kHW) {
    col_c0[tid] = (short)(t_s - cmin); col_c1[tid] = (short)(min(t_s + 1, tab.sw - 1) - cmin);
    col_a0[tid] = (short)t_a0; col_a1[tid] = (short)t_a1;
  } else if (tid >= 192 && tid < 192 + kHH) {
    const int r = tid - 192;
    row_r0[r] = (short)(min(max(t_s, 0), tab.sh - 1) - smin); row_r1[r] = (short)(min(max(t_s + 1, 0), tab.sh - 1) - smin);
    row_b0[r] = (short)t_a0; row_b1[r] = (short)t_a1;
  }
  __syncthreads();
  // 2. horizontal pass of the block rows: hq[r][x] = (S0*a0 + S1*a1) >> 4, all 132 columns, k = r * 132 + x walks without a division
  for (int r = tid / kHW, x = tid % kHW; r < nsr;) {
    hq[r * kHW + x] = (uint16_t)((blk[r * ncol + col_c0[x]] * col_a0[x] + blk[r * ncol + col_c1[x]] * col_a1[x]) >> 4);
    x += kThreads % kHW; r += kThreads / kHW;
    if (x >= kHW) { x -= kHW; r++; }
  }
  __syncthreads();
  tile_vertical_pass(hq, up, row_r0, row_r1, 0, row_b0, row_b1, tid);                             // 3.
  __syncthreads();
  tile_hsum5(up, hs, tid);                                                                       // 4.
  __syncthreads();
  tile_vsum5_store<BLEND, WH>(hs, mask, outp, ops, n, W, H, roi, tx0, ty0, tid, yuyv);               // 5.
}

// ---- alpha blend (deepseg.cc:108-134), stand-alone: bsx_composite_batch ---------------------------------------------------------------
// Every memory instruction COALESCED ACROSS LANES (round 4; the 16-consecutive-pixels-per-lane form it replaced — 16-byte accesses 48 bytes apart between
// neighbouring lanes, every line visited by three instructions — moved 4.8 TB/s and was deleted in round 6).  A lane owns kB4 groups of FOUR pixels, group j of lane l at group index (block * kB4 + j) * 256 + l: one wave instruction = 64 lanes x 12 contiguous bytes
// = 768 bytes = six whole lines (global_load_dwordx3 / global_store_dwordx3, 4-byte aligned), the mask 64 x 4 bytes = two lines.  tools/microbench_mix.hip: a
// plain streaming kernel with this mix and coalesced accesses moves 5.7-6.2 TB/s through HBM.
constexpr int kB4 = 4;
__global__ __launch_bounds__(kThreads) void blend4x4_k(const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ fr,
                                                      const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, unsigned quads_per_frame, long npix) {
  const long n = blockIdx.y;
  const unsigned q0 = blockIdx.x * (kThreads * kB4) + threadIdx.x;      // first 4-pixel group of this lane
  const uint8_t* const a0 = bg + (bg_stride ? n * bg_stride : 0);
  const uint8_t* const b0 = fr + n * npix * 3;
  const uint8_t* const m0 = mask + n * npix;
  uint8_t* const o0 = out + n * npix * 3;
  u3v av[kB4], bv[kB4];
  uint32_t mw[kB4];
#pragma unroll
  for (int j = 0; j < kB4; j++) {
    const unsigned q = min(q0 + j * kThreads, quads_per_frame - 1);     // groups past the end re-read the last one (never stored)
    av[j] = *reinterpret_cast<const u3v*>(a0 + (size_t)q * 12);
    bv[j] = __builtin_nontemporal_load(reinterpret_cast<const u3v*>(b0 + (size_t)q * 12));
    mw[j] = *reinterpret_cast<const uint32_t*>(m0 + (size_t)q * 4);
  }
#pragma unroll
  for (int j = 0; j < kB4; j++) {
    const unsigned q = q0 + j * kThreads;
    const uint32_t a3[3] = {av[j].x, av[j].y, av[j].z}, b3[3] = {bv[j].x, bv[j].y, bv[j].z};
    uint32_t o3[3];
    blend_quad(a3, b3, mw[j], o3);
    if (q < quads_per_frame) __builtin_nontemporal_store(u3v{o3[0], o3[1], o3[2]}, reinterpret_cast<u3v*>(o0 + (size_t)q * 12));
  }
}

// scalar tail / unaligned fallback: one pixel per lane
__global__ __launch_bounds__(kThreads) void blend1_k(const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ fr,
                                                    const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, long npix) {
  const long p = (long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const long n = blockIdx.y;
  int m = mask[n * npix + p];
  const uint8_t* a = bg + (bg_stride ? n * bg_stride : 0) + p * 3;
  const uint8_t* b = fr + (n * npix + p) * 3;
  uint8_t* o = out + (n * npix + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = (uint8_t)((a[c] * m + b[c] * (255 - m)) / 255);
}

// composite outside the ROI = background: copies of the rows above / below the ROI and of the row segments left / right of it (W and roi.x, roi.w multiples of 4 →
// every segment boundary is word aligned).  Lane = FOUR consecutive outside words of frame blockIdx.y (round 4: one word per lane — a division and a 4-byte access per lane —
// ran this pure copy at 2.8 TB/s of writes, 110 us per 256 HD frames with the MLKit ROI): where the four words are adjacent in memory (always, except across a strip or ROI
// boundary) they move as one 16-byte access, otherwise word by word.
__device__ __forceinline__ long outside_word_offset(unsigned i, unsigned wpr, unsigned lw, unsigned rw0, unsigned sw, unsigned full, int W, Rect4 roi) {
  unsigned row, word;
  if (i < full) {
    const unsigned r = i / wpr;
    word = i - r * wpr;
    row = r < (unsigned)roi.y ? r : r + (unsigned)roi.h;
  } else {
    const unsigned j = i - full, r = j / sw, k = j - r * sw;
    row = (unsigned)roi.y + r;
    word = k < lw ? k : rw0 + (k - lw);
  }
  return ((long)row * W) * 3 + (long)word * 4;
}
__global__ __launch_bounds__(kThreads) void outside_roi_copy_k(const uint8_t* __restrict__ bg, long bg_stride, uint8_t* __restrict__ out, int W, int H, Rect4 roi) {
  const unsigned wpr = (unsigned)W * 3 / 4;                                   // words per row
  const unsigned lw = (unsigned)roi.x * 3 / 4, rw0 = (unsigned)(roi.x + roi.w) * 3 / 4, sw = lw + (wpr - rw0);   // strip words per ROI row
  const unsigned full = (unsigned)(H - roi.h) * wpr;                           // words of the rows entirely outside
  const unsigned total = full + (unsigned)roi.h * sw;
  const unsigned i = (blockIdx.x * kThreads + threadIdx.x) * 4u;
  if (i >= total) return;
  const long n = blockIdx.y;
  const uint8_t* src = bg + (bg_stride ? n * bg_stride : 0);
  uint8_t* dst = out + n * (long)W * H * 3;
  const long o0 = outside_word_offset(i, wpr, lw, rw0, sw, full, W, roi);
  if (i + 3 < total && outside_word_offset(i + 3, wpr, lw, rw0, sw, full, W, roi) == o0 + 12) {
    struct __attribute__((packed, aligned(4))) W4 { uint32_t v[4]; };          // 4-byte aligned 16-byte access (a strip need not start on a 16-byte boundary)
    *reinterpret_cast<W4*>(dst + o0) = *reinterpret_cast<const W4*>(src + o0);
    return;
  }
  for (unsigned k = 0; k < 4 && i + k < total; k++) {
    const long o = k ? outside_word_offset(i + k, wpr, lw, rw0, sw, full, W, roi) : o0;
    *reinterpret_cast<uint32_t*>(dst + o) = *reinterpret_cast<const uint32_t*>(src + o);
  }
}

// ---- generic BGR resize -------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void resize_bgr_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ResizeTab tab) {
  const unsigned p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= (unsigned)(tab.dw * tab.dh)) return;
  const int y = (int)(p / (unsigned)tab.dw), x = (int)(p - (unsigned)y * (unsigned)tab.dw);
  const long n = blockIdx.y;
  int v[3];
  sample_linear<3>(src + n * (long)tab.sw * tab.sh * 3, (long)tab.sw * 3, tab, x, y, v);
  uint8_t* o = dst + (n * (long)tab.dw * tab.dh + p) * 3;
  o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
}

__global__ __launch_bounds__(kThreads) void yuyv_k(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, long pairs) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= pairs) return;
  const uint8_t* p = in + i * 6;
  out[i] = yuyv_pair(p[0], p[1], p[2], p[3], p[4], p[5]);
}
// composite outside the ROI in YUYV form = the background converted: lane = one pixel pair of the frame, pairs inside the ROI belong to the tiles
// outside the ROI when the composite is flipped: one lane per group of four pixels of the frame (the ROI's groups return at once); BGR or YUYV out
__global__ __launch_bounds__(kThreads) void outside_roi_flip_k(const uint8_t* __restrict__ bg, long bg_stride, uint8_t* __restrict__ out, int W, int H, Rect4 roi, int yuyv_flip) {
  const unsigned i = blockIdx.x * kThreads + threadIdx.x, gpr = (unsigned)W / 4;
  if (i >= gpr * (unsigned)H) return;
  const int row = (int)(i / gpr), x = (int)(i - (unsigned)row * gpr) * 4;
  if (row >= roi.y && row < roi.y + roi.h && x >= roi.x && x < roi.x + roi.w) return;
  const long n = blockIdx.y;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(bg + (bg_stride ? n * bg_stride : 0) + ((long)row * W + x) * 3);
  uint32_t w[3] = {p[0], p[1], p[2]};
  const bool fh = (yuyv_flip & 2) != 0, fv = (yuyv_flip & 4) != 0;
  if (fh) reverse4px(w);
  const int oy = fv ? H - 1 - row : row, ox = fh ? W - 4 - x : x;
  if (yuyv_flip & 1) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (n * (long)W * H + (long)oy * W + ox) * 2);
    o[0] = yuyv_pair(w[0] & 255u, (w[0] >> 8) & 255u, (w[0] >> 16) & 255u, w[0] >> 24, w[1] & 255u, (w[1] >> 8) & 255u);
    o[1] = yuyv_pair((w[1] >> 16) & 255u, w[1] >> 24, w[2] & 255u, (w[2] >> 8) & 255u, (w[2] >> 16) & 255u, w[2] >> 24);
  } else {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (n * (long)W * H + (long)oy * W + ox) * 3);
    o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
  }
}
__global__ __launch_bounds__(kThreads) void outside_roi_yuyv_k(const uint8_t* __restrict__ bg, long bg_stride, uint32_t* __restrict__ out, int W, int H, Rect4 roi) {
  const unsigned i = blockIdx.x * kThreads + threadIdx.x, ppr = (unsigned)W / 2;
  if (i >= ppr * (unsigned)H) return;
  const unsigned row = i / ppr, x = (i - row * ppr) * 2;
  if ((int)row >= roi.y && (int)row < roi.y + roi.h && (int)x >= roi.x && (int)x < roi.x + roi.w) return;
  const long n = blockIdx.y;
  const uint8_t* p = bg + (bg_stride ? n * bg_stride : 0) + ((long)row * W + x) * 3;
  out[n * (long)(W / 2) * H + i] = yuyv_pair(p[0], p[1], p[2], p[3], p[4], p[5]);
}

// ---- YUYV → BGR ingest (cv::COLOR_YUV2BGR_YUYV, BT.601 limited range, 20-bit fixed point) --------------------------------
__global__ __launch_bounds__(kThreads) void yuyv_to_bgr_k(const uint32_t* __restrict__ in, uint8_t* __restrict__ out, long pairs) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= pairs) return;
  const int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
  const uint32_t p = in[i];                       // Y0 | U<<8 | Y1<<16 | V<<24
  const int y0 = p & 255, u = (int)((p >> 8) & 255) - 128, y1 = (p >> 16) & 255, v = (int)(p >> 24) - 128;
  const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
  const int ya = max(0, y0 - 16) * CY, yb = max(0, y1 - 16) * CY;
  uint8_t* o = out + 6 * i;
  o[0] = (uint8_t)min(max((ya + buv) >> SH, 0), 255); o[1] = (uint8_t)min(max((ya + guv) >> SH, 0), 255); o[2] = (uint8_t)min(max((ya + ruv) >> SH, 0), 255);
  o[3] = (uint8_t)min(max((yb + buv) >> SH, 0), 255); o[4] = (uint8_t)min(max((yb + guv) >> SH, 0), 255); o[5] = (uint8_t)min(max((yb + ruv) >> SH, 0), 255);
}

// ---- cv::flip on packed BGR (deepseg.cc:667-673): code 0 = around the x axis (rows reversed), > 0 = around the y axis
// (columns reversed), < 0 = both.  Out of place; lane = one pixel of frame blockIdx.y.
__global__ __launch_bounds__(kThreads) void flip_bgr_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, int code) {
  const unsigned p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= (unsigned)(w * h)) return;
  const int y = (int)(p / (unsigned)w), x = (int)(p - (unsigned)y * (unsigned)w);
  const int sy = code <= 0 ? h - 1 - y : y, sx = code != 0 ? w - 1 - x : x;
  const long base = (long)blockIdx.y * w * h * 3;
  const uint8_t* s = src + base + ((long)sy * w + sx) * 3;
  uint8_t* d = dst + base + (long)p * 3;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

__global__ __launch_bounds__(kThreads) void fill_k(uint4* p, uint4 v, long n16) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i < n16) p[i] = v;
}

}  // namespace

// resize + bilateral in one launch (prep_fused_k); input (f32 [n][inH][inW][3]) and / or input_u8 (R|G<<8|B<<16 [n][inH][inW]): whichever is non-null is written
bool prep_yuyv_fusable(int W, Rect4 roi, const ResizeTab& tab) {      // the YIN form of prep_fused_k: INTER_LINEAR proper, whole macropixels, an 8-byte window inside the row
  return tab.mode == 0 && (W & 1) == 0 && (roi.x & 1) == 0 && (W - roi.x) * 2 >= 8;
}

hipError_t launch_prep_fused(const uint8_t* frames, int W, int H, Rect4 roi, float* input, uint32_t* input_u8, int inW, int inH, Rect4 in_roi, ResizeTab tab,
                             BilateralParams bp, int n, hipStream_t s, bool yuyv_in) {
  if (!input && !input_u8) return hipErrorInvalidValue;
  if (yuyv_in && !prep_yuyv_fusable(W, roi, tab)) return hipErrorInvalidValue;      // (the caller converts the frames first: bsx_api.hip step_impl)
  const int ntx = (inW + 31) / 32, nty = (inH + 31) / 32, TW = (inW + ntx - 1) / ntx, TH = (inH + nty - 1) / nty;      // even tiles, <= 32 x 32
  const long per_frame = (long)inW * inH;
  static const bool xcd_on = !(BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 0);      // A/B timing: 0 = plain frame-major workgroup order
  const int chunk = 1 << 20;                                            // frames per launch: keeps the 1-D grid far below 2^31 workgroups
  for (int n0 = 0; n0 < n; n0 += chunk) {
    const int nn = n - n0 < chunk ? n - n0 : chunk;
    const dim3 grid((unsigned)(ntx * nty) * (unsigned)nn);
    const uint8_t* fr = frames + (size_t)n0 * W * H * (yuyv_in ? 2 : 3);
    float* f = input ? input + (size_t)n0 * per_frame * 3 : nullptr;
    uint32_t* u = input_u8 ? input_u8 + (size_t)n0 * per_frame : nullptr;
#define BSX_PF(O, L) prep_fused_k<O, L><<<grid, kThreads, 0, s>>>(fr, W, H, roi, f, u, inW, inH, in_roi, tab, bp, TW, TH, ntx, nty, xcd_on ? nn : 0)
#define BSX_PFY(O) prep_fused_k<O, true, true><<<grid, kThreads, 0, s>>>(fr, W, H, roi, f, u, inW, inH, in_roi, tab, bp, TW, TH, ntx, nty, xcd_on ? nn : 0)
    if (yuyv_in) { if (f && u) BSX_PFY(3); else if (u) BSX_PFY(2); else BSX_PFY(1); }
    else
    if (tab.mode == 0 && (W - roi.x) * 3 >= 8) { if (f && u) BSX_PF(3, true); else if (u) BSX_PF(2, true); else BSX_PF(1, true); }   // (the 8-byte tap window needs an 8-byte row)
    else { if (f && u) BSX_PF(3, false); else if (u) BSX_PF(2, false); else BSX_PF(1, false); }
#undef BSX_PF
#undef BSX_PFY
  }
  return hipGetLastError();
}

bool bilateral_taps_match(const BilateralParams& bp) {
  static const int ty[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2}, tx[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
  for (int k = 0; k < 13; k++) if (bp.off_y[k] != ty[k] || bp.off_x[k] != tx[k]) return false;
  return true;
}


hipError_t launch_decode(int model_type, const float* logits, uint8_t* ofinal, int npix, int nch, int n, hipStream_t s) {
  long total = (long)n * npix;
  int type = model_type == 1 ? 1 : (model_type == 3 ? 3 : 2);
  if (type == 3 && nch == 2 && (total & 3) == 0 && ((((uintptr_t)logits) & 15) | (((uintptr_t)ofinal) & 3)) == 0)
    decode_meet4_k<<<blocks_for(total / 4), kThreads, 0, s>>>(reinterpret_cast<const float4*>(logits), reinterpret_cast<uint32_t*>(ofinal), total / 4);
  else if (type == 2 && nch == 1 && (total & 3) == 0 && ((((uintptr_t)logits) & 15) | (((uintptr_t)ofinal) & 3)) == 0)
    decode_thresh4_k<<<blocks_for(total / 4), kThreads, 0, s>>>(reinterpret_cast<const float4*>(logits), reinterpret_cast<uint32_t*>(ofinal), total / 4);
  else if (type == 1 && nch <= kArgmaxMaxCh) decode_argmax_k<<<blocks_for(total), kThreads, 0, s>>>(logits, ofinal, total, nch);
  else decode_k<<<blocks_for(total), kThreads, 0, s>>>(type, logits, ofinal, total, nch);
  return hipGetLastError();
}

// Every tile's source block must fit the LDS staging area of mask_tile_k (host tables, checked once per ResizeTab).
int mask_tile_width() { return kTW; }
int mask_tile_height() { return kTH; }
bool mask_tile_fits(const int* xofs, const int* yofs, int sw, int sh, int dw, int dh) {
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  int max_rows = 0, max_cols = 0;
  for (int ty0 = 0; ty0 < dh; ty0 += kTH) {
    const int lo = std::max(ty0 - 2, 0), hi = std::min(ty0 + kTH + 1, dh - 1);
    max_rows = std::max(max_rows, clampi(yofs[hi] + 1, 0, sh - 1) - clampi(yofs[lo], 0, sh - 1) + 1);
  }
  for (int tx0 = 0; tx0 < dw; tx0 += kTW) {
    const int lo = std::max(tx0 - 2, 0), hi = std::min(tx0 + kTW + 1, dw - 1);
    max_cols = std::max(max_cols, std::min(xofs[hi] + 1, sw - 1) - xofs[lo] + 1);
  }
  return max_rows <= kMaxSrcRows && max_rows * max_cols <= kSrcBlockBytes;
}

// BSX_NO_MASK_TILE=1 (read ONCE, when the context builds its resize tables: it clears ResizeTab::tile_ok) forces the generic kernel: the parity tests use it to cover both.
static bool mask_tile_usable(const ResizeTab& tab) { return tab.mode == 0 && tab.tile_ok; }

hipError_t launch_mask_upscale_blur(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H, Rect4 roi,
                                    int n, hipStream_t s) {
  const int ntx = (roi.w + kTW - 1) / kTW, nty = (roi.h + kTH - 1) / kTH;
  if ((unsigned long long)ntx * nty * (unsigned long long)n >= (1ull << 31)) return hipErrorInvalidValue;
  static const bool xcd_on = !(BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 0);      // A/B timing: 0 = plain frame-major workgroup order
  // one-XCD-per-frame order only where neighbouring tiles SHARE cache lines, i.e. the ROI's rows do not start on a 128-byte line (roi.x = 80 / 280: measured
  // mask+blend 0.480 -> 0.464 ms at 256 HD MLKit streams); on line-aligned geometries nothing is shared and the plain order measured 2-3 % faster (profiles/r03o)
  const bool shared_lines = ((roi.x * 3) & 127) != 0 || ((W * 3) & 127) != 0;
  const int nf = (xcd_on && shared_lines) ? n : 0;
  dim3 grid((unsigned)(ntx * nty) * (unsigned)n);
  if (hipError_t e = launch_tile_class(ofinal, outW, outH, in_roi, tab, roi, n, s)) return e;
  if (mask_tile_usable(tab)) mask_tile_k<false><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, nullptr, 0, nullptr, nullptr, 0, ntx, nty, nf, 0, nty);
  else mask_upscale_blur_k<false><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, nullptr, 0, nullptr, nullptr, 0, ntx, nty, nf);
  return hipGetLastError();
}

// Tile classes of one frame per workgroup: is a tile's source block — the extents mask_tile_k computes — all 0xFF (→ 1), all 0x00 (→ 2) or neither (→ 0)?
// Two steps through LDS so that every model-resolution byte is looked at once per tile COLUMN, with wide loads: (1) item (row r, tile column tbx) scans the bytes
// [cmin(tbx), cmax(tbx)] of row r as unaligned dwords, AND / OR accumulated → two flags; (2) tile (tby, tbx) ANDs the flags of its rows [smin(tby), smax(tby)].
// (The first version gave every tile a wave that walked its block row by row, byte by byte: 17 us at 256 lite/VGA streams, 151 us at 1024 DeepLab streams —
// more than the shortcut saved; profiles/r04g.)
constexpr int kClsMaxItems = 8192, kClsMaxTx = 16;
__global__ __launch_bounds__(kThreads) void tile_class_k(const uint8_t* __restrict__ ofinal, int outW, int outH, Rect4 q, ResizeTab tab, Rect4 roi, int ntx, int nty) {
  __shared__ uint8_t f255[kClsMaxItems], f0[kClsMaxItems];
  __shared__ int cmn[kClsMaxTx], cmx[kClsMaxTx];
  const int n = blockIdx.x, tid = threadIdx.x;
  const uint8_t* const fr = ofinal + (long)n * outW * outH + (long)q.y * outW + q.x;
  if (tid < ntx) {
    const int tx0 = tid * kTW, gx_lo = max(tx0 - 2, 0), gx_hi = min(tx0 + kTW + 1, roi.w - 1);
    cmn[tid] = tab.xofs[gx_lo]; cmx[tid] = min(tab.xofs[gx_hi] + 1, tab.sw - 1);
  }
  __syncthreads();
  struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
  const int items = tab.sh * ntx;
  for (int i = tid; i < items; i += kThreads) {
    const int r = i / ntx, tbx = i - r * ntx, c0 = cmn[tbx], len = cmx[tbx] - c0 + 1;
    const uint8_t* p = fr + (unsigned)(r * outW + c0);
    uint32_t a = 0xFFFFFFFFu, o = 0u;
    if (len >= 4) {
#pragma unroll 4
      for (int k = 0; k + 4 <= len; k += 4) { const uint32_t w = reinterpret_cast<const U4*>(p + k)->v; a &= w; o |= w; }
      const uint32_t w = reinterpret_cast<const U4*>(p + len - 4)->v;      // the last four bytes (overlapping the loop's: AND / OR do not care)
      a &= w; o |= w;
    } else {
      for (int k = 0; k < len; k++) { const uint32_t w = p[k] * 0x01010101u; a &= w; o |= w; }
    }
    f255[i] = a == 0xFFFFFFFFu; f0[i] = o == 0u;
  }
  __syncthreads();
  for (int t = tid; t < ntx * nty; t += kThreads) {
    const int tby = t / ntx, tbx = t - tby * ntx, ty0 = tby * kTH;
    const int gy_lo = max(ty0 - 2, 0), gy_hi = min(ty0 + kTH + 1, roi.h - 1);
    const int smin = min(max(tab.yofs[gy_lo], 0), tab.sh - 1), smax = min(max(tab.yofs[gy_hi] + 1, 0), tab.sh - 1);
    int all255 = 1, all0 = 1;
    for (int r = smin; r <= smax; r++) { all255 &= f255[r * ntx + tbx]; all0 &= f0[r * ntx + tbx]; }
    tab.tile_class[(size_t)n * (size_t)(ntx * nty) + t] = (uint8_t)(all255 ? 1 : (all0 ? 2 : 0));
  }
}

hipError_t launch_tile_class(const uint8_t* ofinal, int outW, int outH, const Rect4& in_roi, const ResizeTab& tab, const Rect4& roi, int n, hipStream_t s) {
  if (!tab.tile_class || !mask_tile_usable(tab)) return hipSuccess;
  const int ntx = (roi.w + kTW - 1) / kTW, nty = (roi.h + kTH - 1) / kTH;
  if (ntx > kClsMaxTx || (long)tab.sh * ntx > kClsMaxItems) return hipMemsetAsync(tab.tile_class, 0, (size_t)ntx * nty * n, s);      // out of the classifier's range: all general
  tile_class_k<<<dim3((unsigned)n), kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, roi, ntx, nty);
  return hipGetLastError();
}

bool mask_blend_fusable(int W, int H, Rect4 roi, const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* out) {
  return (W % 4) == 0 && (roi.x % 4) == 0 && (roi.w % 4) == 0 && (bg_stride % 4) == 0 &&
         ((((uintptr_t)bg) | ((uintptr_t)frames) | ((uintptr_t)out)) & 3) == 0;
}

hipError_t launch_mask_blend(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H, Rect4 roi,
                             const uint8_t* bg, size_t bg_stride, const uint8_t* frames, uint8_t* out, int n, hipStream_t s, int yuyv, int lds_pad) {
  // lds_pad (bytes of dynamic LDS nobody uses): an OCCUPANCY CAP for the two-deep pipeline (bsx_step_batch_pipelined) — left alone this HBM-bound launch takes every wave
  // slot of every CU and the latency-bound network kernels of the other stream run on what is left (seg_head 2.65x slower, profiles/r04k); with the pad only
  // 160 KB / (static + pad) workgroups fit a CU
  // outside the ROI the persistent mask is 255 forever (libbackscrub.cc:248-249), i.e. the composite there IS the background
  // ((a*255 + b*0)/255 == a): those strips are copied, the ROI is composited by the mask tiles
  // `yuyv`: bit 0 = YUYV output, bits 1-2 = horizontal / vertical flip of the composite, bit 3 = do not store the full-resolution mask, bit 4 = `frames` is YUYV
  // 4:2:2 (bsx.h: BSX_STEP_*); outside the ROI no frame pixel is read
  if ((yuyv & 6) && (roi.x != 0 || roi.y != 0 || roi.w != W || roi.h != H))
    outside_roi_flip_k<<<dim3(blocks_for((long)(W / 4) * H), n), kThreads, 0, s>>>(bg, (long)bg_stride, out, W, H, roi, yuyv);
  else if ((yuyv & 1) && (roi.x != 0 || roi.y != 0 || roi.w != W || roi.h != H))
    outside_roi_yuyv_k<<<dim3(blocks_for((long)(W / 2) * H), n), kThreads, 0, s>>>(bg, (long)bg_stride, reinterpret_cast<uint32_t*>(out), W, H, roi);
  else if (roi.x != 0 || roi.y != 0 || roi.w != W || roi.h != H)
    outside_roi_copy_k<<<dim3(blocks_for(((long)(H - roi.h) * (W * 3 / 4) + (long)roi.h * ((W - roi.w) * 3 / 4) + 3) / 4), n), kThreads, (size_t)lds_pad, s>>>(bg, (long)bg_stride, out, W, H,
                                                                                                                                 roi);
  const int ntx = (roi.w + kTW - 1) / kTW, nty = (roi.h + kTH - 1) / kTH;
  if ((unsigned long long)ntx * nty * (unsigned long long)n >= (1ull << 31)) return hipErrorInvalidValue;
  static const bool xcd_on = !(BSX_DBG_ENV("BSX_XCD_TILES") && atoi(BSX_DBG_ENV("BSX_XCD_TILES")) == 0);      // A/B timing: 0 = plain frame-major workgroup order
  // one-XCD-per-frame order only where neighbouring tiles SHARE cache lines, i.e. the ROI's rows do not start on a 128-byte line (roi.x = 80 / 280: measured
  // mask+blend 0.480 -> 0.464 ms at 256 HD MLKit streams); on line-aligned geometries nothing is shared and the plain order measured 2-3 % faster (profiles/r03o)
  const bool shared_lines = ((roi.x * 3) & 127) != 0 || ((W * 3) & 127) != 0;
  const int nf = (xcd_on && shared_lines) ? n : 0;
  dim3 grid((unsigned)(ntx * nty) * (unsigned)n);
  if (hipError_t e = launch_tile_class(ofinal, outW, outH, in_roi, tab, roi, n, s)) return e;
  static const bool no_early_bg = BSX_DBG_ENV("BSX_NO_EARLY_BG") != nullptr;      // A/B timing: bit 6 of the flag word = request a shared background only after the tile's class is known (rounds 1-4)
  if (no_early_bg) yuyv |= 64;
  static const bool plain_stores = BSX_DBG_ENV("BSX_TILE_PLAIN_STORES") != nullptr;      // A/B timing: bit 7 = the composite leaves with plain instead of nontemporal stores
  if (plain_stores) yuyv |= 128;
  const bool yin = (yuyv & 16) != 0;
  if (mask_tile_usable(tab)) {
    const bool f0 = (yuyv & ~16) == 0;
    // whole tile rows (every item inside the ROI: the WH instantiation) and, where the ROI height is not a multiple of the tile height (720 = 22.5 x 32), the last,
    // partial row as a second launch of the edge-testing instantiation
    const bool whole_x = f0 && roi.w % kTW == 0 && ((uintptr_t)mask & 3) == 0;      // (W, roi.x multiples of 4: mask_blend_fusable)
    const int nty_whole = whole_x ? roi.h / kTH : 0;
#define BSX_MT(Y, F, WHL, GRID, NTY, TYB) mask_tile_k<true, Y, F, WHL><<<GRID, kThreads, (size_t)lds_pad, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, bg, (long)bg_stride, frames, out, yuyv, ntx, NTY, nf, TYB, nty)
    if (nty_whole > 0) {
      const dim3 gw((unsigned)(ntx * nty_whole) * (unsigned)n), gr((unsigned)(ntx * (nty - nty_whole)) * (unsigned)n);
      if (yin) { BSX_MT(true, true, true, gw, nty_whole, 0); if (nty > nty_whole) BSX_MT(true, true, false, gr, nty - nty_whole, nty_whole); }
      else { BSX_MT(false, true, true, gw, nty_whole, 0); if (nty > nty_whole) BSX_MT(false, true, false, gr, nty - nty_whole, nty_whole); }
    } else if (yin) { if (f0) BSX_MT(true, true, false, grid, nty, 0); else BSX_MT(true, false, false, grid, nty, 0); }
    else { if (f0) BSX_MT(false, true, false, grid, nty, 0); else BSX_MT(false, false, false, grid, nty, 0); }
#undef BSX_MT
  } else if (yin) mask_upscale_blur_k<true, true><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, bg, (long)bg_stride, frames, out, yuyv, ntx, nty, nf);
  else mask_upscale_blur_k<true, false><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, bg, (long)bg_stride, frames, out, yuyv, ntx, nty, nf);
  return hipGetLastError();
}

hipError_t launch_blend(const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* masks, uint8_t* out, size_t npix, int n,
                        hipStream_t s) {
  // lane-coalesced form: 4-byte alignment and whole 4-pixel groups are enough; anything else takes the per-pixel kernel
  const bool quad_ok = (((uintptr_t)bg | (uintptr_t)frames | (uintptr_t)masks | (uintptr_t)out) & 3) == 0 && (npix % 4 == 0) && (bg_stride % 4 == 0) && npix / 4 < (1l << 31);
  const long quads = quad_ok ? (long)(npix / 4) : 0;
  for (int n0 = 0; n0 < n; n0 += kMaxGridY) {
    const int nn = n - n0 < kMaxGridY ? n - n0 : kMaxGridY;
    const uint8_t* bgp = bg + (size_t)n0 * bg_stride;
    const uint8_t* fp = frames + (size_t)n0 * npix * 3;
    const uint8_t* mp = masks + (size_t)n0 * npix;
    uint8_t* op = out + (size_t)n0 * npix * 3;
    if (quads) blend4x4_k<<<dim3((unsigned)((quads + kThreads * kB4 - 1) / (kThreads * kB4)), nn), kThreads, 0, s>>>(bgp, (long)bg_stride, fp, mp, op, (unsigned)quads, (long)npix);
    else blend1_k<<<dim3(blocks_for((long)npix), nn), kThreads, 0, s>>>(bgp, (long)bg_stride, fp, mp, op, (long)npix);
  }
  return hipGetLastError();
}

hipError_t launch_resize_bgr(const uint8_t* src, uint8_t* dst, ResizeTab tab, int n, hipStream_t s) {
  for (int n0 = 0; n0 < n; n0 += kMaxGridY) {
    const int nn = n - n0 < kMaxGridY ? n - n0 : kMaxGridY;
    resize_bgr_k<<<dim3(blocks_for((long)tab.dw * tab.dh), nn), kThreads, 0, s>>>(src + (size_t)n0 * tab.sw * tab.sh * 3, dst + (size_t)n0 * tab.dw * tab.dh * 3, tab);
  }
  return hipGetLastError();
}

// ---- cv::GaussianBlur(bg, bg, Size(n, n), 0) on packed BGR u8 (app/deepseg.cc:657-658, `-p bgblur:<n>`) ------------------------------
// OpenCV's 8-bit Gaussian is an integer separable filter: u8-valued coefficients c[k] = cvRound(kernel * 256) (host table), the
// horizontal pass Σ c·src in u16 (8 fractional bits), the vertical pass Σ c·h in u32 (16 fractional bits), (v + 2^15) >> 16,
// BORDER_REFLECT_101.  With Σc <= 257 neither intermediate can overflow, so the saturating adds of the reference never trigger.
// Tile = 64 x 16 output pixels per workgroup.  The source tile (+ radius r <= 15 on every side) is de-interleaved into three byte
// planes in LDS so that consecutive taps of one channel are consecutive bytes: the horizontal pass then does 4 taps per
// v_dot4_u32_u8 on windows shifted out of aligned dwords with v_alignbyte; its u16 results are stored transposed ([col][row]) so the
// vertical pass does 2 taps per v_dot2_u32_u16 the same way.  HBM traffic = the image once in, once out.
constexpr int kGTW = 64, kGTH = 16, kGMaxR = 15, kGSW = kGTW + 2 * kGMaxR, kGSH = kGTH + 2 * kGMaxR;    // 94 x 46 source tile
constexpr int kGSrcStride = 96;                  // bytes per plane row (multiple of 4)
constexpr int kGHStride = 48;                    // u16 per hbuf column (rows of the source tile, even)
// hbuf column c of a plane starts 2 * (c / 8) dwords late: the horizontal pass writes 16 column groups (4 columns = 96 dwords apart: banks 0 / 32 only, an 8-way
// conflict on every store) at once, the shift spreads them over 16 banks; the vertical pass reads 8 CONSECUTIVE columns per wave (24 dwords apart: conflict-free),
// which share one shift.
constexpr int kGHPlane = kGTW * kGHStride + 32;  // u16 per plane: 64 columns + the largest shift (14 dwords) rounded up
__device__ __forceinline__ int gh_col(int plane, int col) { return plane * kGHPlane + col * kGHStride + 4 * (col >> 3); }
// c[k] packed 4 per dword (bytes) and 2 per dword (u16), zero padded — once per output phase: c4[j] is the tap sequence delayed by j bytes, c2[s] by s halves, so that
// the 4 neighbouring outputs of a horizontal item (2 of a vertical item) are dot products of the SAME aligned data words with different coefficient words
// (one v_dot per word and output) instead of realigned windows with one coefficient set (v_alignbyte + v_dot per word and output).
struct GaussCoef { uint32_t c4[4][9]; uint32_t c2[2][17]; int n, r, sh; };   // sh: the LDS planes start sh (0..3) pixels left of the tile's first source column
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int reflect101_far(int p, int len) {   // cv::borderInterpolate(BORDER_REFLECT_101) for overshoots beyond one period
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}
// MODE 0: the vertical pass stores its bytes itself (any geometry).  MODE 1 (W % 4 == 0, 4-byte aligned images): the vertical pass leaves the blurred tile in LDS
// as packed BGR rows and the workgroup writes it as whole dwords, one 4-pixel group (12 B) per lane — each lane of the vertical pass owns two ROWS of one column of
// one plane, i.e. byte stores scattered over 16 image rows.  MODE 2 = MODE 1 with the alpha blend (deepseg.cc:108-134) applied to the group on its way out:
// `-p bgblur` without `-b` (deepseg.cc:652-661) composites the camera frame over ITS OWN blur, so the blurred image is consumed where it is produced — it never
// exists in HBM, and neither does the blend's second read of the frame (the lane's frame / mask words are requested at the top of the kernel).
// NT = ceil((ksize + 3 + sh) / 4), a template parameter: the tap loops are straight-line code over NT horizontal / 2 NT - 1 vertical data words (the coefficient bytes past
// ksize are zero, so the surplus taps add 0 whatever bytes they read), not loops predicated tap by tap.
// The kernel is bound by VALU issue, not by memory (ksize 3 costs half of ksize 25), so the per-byte index arithmetic is what there is to save:
//   * items advance by precomputed (row, column) steps instead of a runtime division per item, addresses are 32-bit offsets from a uniform base;
//   * opt bit 0 (rows dword aligned), tiles whose source columns lie inside the image: the source tile is staged 4 pixels (three dwords) per item and
//     de-interleaved with six v_perm_b32 into one dword per plane, instead of three byte loads + three byte LDS stores per pixel (rows still reflect).
constexpr int kGOStride = 196;                   // bytes per packed output row in LDS: 64 px x 3 B + 4 (49 dwords: the 16 rows start on distinct banks)
static_assert(kGTH * (kGTW / 4) == kThreads && kGTH * kGOStride <= 3 * kGSH * kGSrcStride, "one 4-pixel group per lane; the output tile aliases the source planes");
template <int MODE, int NT>
__global__ __launch_bounds__(kThreads) void gauss_blur_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint8_t* __restrict__ mask, int W, int H, GaussCoef gc,
                                                         int opts) {
  __shared__ __attribute__((aligned(16))) uint8_t s_src[3 * kGSH * kGSrcStride];          // [plane][row][col]; MODE >= 1: later the packed output tile [row][kGOStride]
  __shared__ __attribute__((aligned(16))) uint16_t s_h[3 * kGHPlane];                     // [plane][col][row], gh_col()
  const size_t img = (size_t)blockIdx.z * (size_t)W * H * 3;
  const uint8_t* in = src + img;
  uint8_t* out = dst + img;
  const int x0 = blockIdx.x * kGTW, y0 = blockIdx.y * kGTH, r = gc.r;
  const int SW = kGTW + 2 * r, SH = kGTH + 2 * r;                                          // live part of the source tile
  // MODE >= 1: this lane's 4-pixel group of the tile
  const int orow = threadIdx.x >> 4, ogx = x0 + 4 * (threadIdx.x & 15), ogy = y0 + orow;
  const bool olive = MODE >= 1 && ogx < W && ogy < H;
  uint32_t fr[3] = {0u, 0u, 0u}, mw = 0u;
  if (MODE == 2 && olive) {
    const uint32_t* fp = reinterpret_cast<const uint32_t*>(in + (unsigned)(ogy * W + ogx) * 3u);
    fr[0] = fp[0]; fr[1] = fp[1]; fr[2] = fp[2];
    mw = *reinterpret_cast<const uint32_t*>(mask + (size_t)blockIdx.z * (size_t)W * H + (unsigned)(ogy * W + ogx));
  }
  // 1. stage + de-interleave (reflected at the image border)
  //    The planes start at source column x0 - r - sh, sh = (-r) & 3 when 4-pixel staging is on (a multiple of 4 → aligned 12-byte groups), else 0; the horizontal
  //    pass never sees the difference: its coefficient words are delayed by sh more bytes (gauss_coefficients).
  const int sh = gc.sh, G = (SW + sh + 3) >> 2;                        // 4-pixel groups per staged row
  const bool xin = (opts & 1) && x0 - r - sh >= 0 && x0 - r - sh + 4 * G <= W;   // uniform: every group of every row lies inside the image row
  if (xin) {                                                          // item = (row, group of 4 pixels): 12 source bytes → one dword per plane
    int row = (int)threadIdx.x / G, grp = (int)threadIdx.x - row * G;
    const int dr = kThreads / G, dc = kThreads - dr * G;
    const unsigned xb = (unsigned)(x0 - r - sh);
    while (row < SH) {
      const int gy = reflect101_far(y0 - r + row, H);
      const uint32_t* p = reinterpret_cast<const uint32_t*>(in + ((unsigned)(gy * W) + xb + 4u * (unsigned)grp) * 3u);
      const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];                 // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
      const uint32_t pb = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00060300u), 0x05020100u);
      const uint32_t pg = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00070401u), 0x06020100u);
      const uint32_t pr = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00000502u), 0x07040100u);
      uint32_t* o = reinterpret_cast<uint32_t*>(s_src + row * kGSrcStride) + grp;
      o[0] = pb; o[kGSH * kGSrcStride / 4] = pg; o[2 * kGSH * kGSrcStride / 4] = pr;
      grp += dc; row += dr;
      if (grp >= G) { grp -= G; row++; }
    }
  } else {
    int row = (int)threadIdx.x / SW, col = (int)threadIdx.x - row * SW;
    const int dr = kThreads / SW, dc = kThreads - dr * SW;
    while (row < SH) {
      const int gy = reflect101_far(y0 - r + row, H), gx = reflect101_far(x0 - r + col, W);
      const uint8_t* p = in + (unsigned)(gy * W + gx) * 3u;
      const uint8_t b = p[0], g = p[1], rr = p[2];
      s_src[(0 * kGSH + row) * kGSrcStride + col + sh] = b;
      s_src[(1 * kGSH + row) * kGSrcStride + col + sh] = g;
      s_src[(2 * kGSH + row) * kGSrcStride + col + sh] = rr;
      col += dc; row += dr;
      if (col >= SW) { col -= SW; row++; }
    }
  }
  __syncthreads();
  // 2. horizontal pass: item = (plane, source row, group of 4 output columns); consecutive items of a lane are kThreads / 16 = 16 rows apart (SH > 16: one wrap at most)
  static_assert(kGTW / 4 == 16 && kThreads / 16 == kGTH && NT >= 2 && NT <= 9, "lane → (row, column group) mapping of the horizontal pass");
  for (int grp = threadIdx.x & 15, row = threadIdx.x >> 4, plane = 0; plane < 3;) {
    const uint32_t* rowp = reinterpret_cast<const uint32_t*>(s_src + (plane * kGSH + row) * kGSrcStride) + grp;   // bytes 4*grp ..
    uint32_t d[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) d[t] = rowp[t];                           // stays inside the 96-byte row: 4 * (15 + NT) <= 96
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_udot4(d[t], gc.c4[j][t], acc[j], false);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) s_h[gh_col(plane, 4 * grp + j) + row] = (uint16_t)acc[j];      // sum(c) <= 257 (gauss_coefficients): 255 * 257 fits 16 bits
    row += kGTH;
    if (row >= SH) { row -= SH; plane++; }
  }
  __syncthreads();
  // 3. vertical pass: item = (plane, output column, pair of output rows)
  constexpr int NP = 2 * NT - 1;                                           // ksize <= 4 NT - 3: ceil((ksize + 1) / 2) data words
  for (int i = threadIdx.x; i < 3 * kGTW * (kGTH / 2); i += kThreads) {
    const int m = i & (kGTH / 2 - 1), pc = i / (kGTH / 2), col = pc & (kGTW - 1), plane = pc / kGTW;
    const uint32_t* colp = reinterpret_cast<const uint32_t*>(s_h + gh_col(plane, col)) + m;                        // rows 2m ..
    uint32_t hd[NP];
#pragma unroll
    for (int t = 0; t < NP; t++) hd[t] = colp[t];                           // 2 (m + NP - 1) + 1 <= 47 < kGHStride
    uint32_t a0 = 0, a1 = 0;
#pragma unroll
    for (int t = 0; t < NP; t++) {
      a0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, hd[t]), __builtin_bit_cast(us2v, gc.c2[0][t]), a0, false);
      a1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, hd[t]), __builtin_bit_cast(us2v, gc.c2[1][t]), a1, false);
    }
    const uint8_t v0 = (uint8_t)min((a0 + (1u << 15)) >> 16, 255u), v1 = (uint8_t)min((a1 + (1u << 15)) >> 16, 255u);
    if (MODE >= 1) {
      s_src[(2 * m) * kGOStride + 3 * col + plane] = v0;
      s_src[(2 * m + 1) * kGOStride + 3 * col + plane] = v1;
    } else {
      const int gx = x0 + col, gy = y0 + 2 * m;
      if (gx < W) {
        if (gy < H) out[(unsigned)(gy * W + gx) * 3u + plane] = v0;
        if (gy + 1 < H) out[(unsigned)((gy + 1) * W + gx) * 3u + plane] = v1;
      }
    }
  }
  if (MODE >= 1) {
    __syncthreads();
    if (olive) {
      const uint32_t* bp = reinterpret_cast<const uint32_t*>(s_src + orow * kGOStride + 12 * (threadIdx.x & 15));
      uint32_t o3[3] = {bp[0], bp[1], bp[2]};
      if (MODE == 2) { const uint32_t a[3] = {o3[0], o3[1], o3[2]}; blend_quad(a, fr, mw, o3); }    // background = the blur, foreground = the frame itself
      uint32_t* op = reinterpret_cast<uint32_t*>(out + (unsigned)(ogy * W + ogx) * 3u);
      op[0] = o3[0]; op[1] = o3[1]; op[2] = o3[2];
    }
  }
}

// Coefficients of cv::GaussianBlur(ksize = n, sigma = 0) for 8-bit images (OpenCV 3.4 - 4.4 rule; DESIGN.md records the
// version ambiguity): fixed tables for n <= 7, else cvRound(exp(-x^2 / (2 sigma^2)) / sum * 256), sigma = 0.3 * ((n-1)/2 - 1) + 0.8.
bool gauss_coefficients(int n, GaussCoef* gc, int sh) {
  if (n < 1 || n > 2 * kGMaxR + 1 || !(n & 1)) return false;
  unsigned c[32] = {0};
  if (n == 1) c[0] = 256;
  else if (n == 3) { c[0] = 64; c[1] = 128; c[2] = 64; }
  else if (n == 5) { const unsigned t[] = {16, 64, 96, 64, 16}; for (int i = 0; i < 5; i++) c[i] = t[i]; }
  else if (n == 7) { const unsigned t[] = {8, 28, 56, 72, 56, 28, 8}; for (int i = 0; i < 7; i++) c[i] = t[i]; }
  else {
    const double sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8, scale2x = (-0.5 * 0.25) / (sigma * sigma);
    double v[32], sum = 0;
    for (int i = 0, x = 1 - n; i < n; i++, x += 2) { v[i] = std::exp((double)(x * x) * scale2x); sum += v[i]; }
    const double inv = 1.0 / sum;
    for (int i = 0; i < n; i++) c[i] = (unsigned)std::lrint(v[i] * inv * 256.0);
  }
  unsigned total = 0;
  for (int i = 0; i < n; i++) { if (c[i] > 255 && n > 1) return false; total += c[i]; }
  if (total > 257) return false;                          // the kernel relies on Σc·255 fitting 16 bits
  *gc = GaussCoef{};
  gc->n = n; gc->r = n / 2; gc->sh = sh;
  if (n == 1) return true;                                // 256 does not fit a byte: n = 1 is served as a copy by the caller
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < n; i++) gc->c4[j][(i + j + sh) >> 2] |= c[i] << (8 * ((i + j + sh) & 3));
  for (int h = 0; h < 2; h++)
    for (int i = 0; i < n; i++) gc->c2[h][(i + h) >> 1] |= c[i] << (16 * ((i + h) & 1));
  return true;
}
bool gauss_coeff_words(int ksize, int shift, uint32_t* c4, uint32_t* c2) {
  GaussCoef gc;
  if (shift < 0 || shift > 3 || ksize + shift > 32 || !gauss_coefficients(ksize, &gc, shift)) return false;
  memcpy(c4, gc.c4, sizeof(gc.c4));
  memcpy(c2, gc.c2, sizeof(gc.c2));
  return true;
}
static bool gauss_words(const void* a, const void* b, const void* c, int w) {      // whole-dword output groups: 4 pixels = 12 bytes at 4-byte aligned addresses
  static const bool off = [] { const char* e = BSX_DBG_ENV("BSX_GAUSS_BYTE_STORE"); return e && *e == '1'; }();
  return !off && (w & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 3) == 0;
}
static int gauss_opts(const void* src, int w) {                                    // bit 0: 4-pixel staging of interior tiles (dword aligned rows)
  static const bool off = [] { const char* e = BSX_DBG_ENV("BSX_GAUSS_BYTE_STAGE"); return e && *e == '1'; }();
  return !off && (w & 3) == 0 && (((uintptr_t)src) & 3) == 0 ? 1 : 0;
}
template <int MODE>
static void gauss_launch(dim3 grid, hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* mask, int w, int h, const GaussCoef& gc, int opts) {
  switch ((gc.n + gc.sh + 6) >> 2) {                      // NT = ceil((ksize + 3 + sh) / 4): 2 .. 9
#define BSX_G(NT) case NT: gauss_blur_k<MODE, NT><<<grid, kThreads, 0, s>>>(src, dst, mask, w, h, gc, opts); break;
    BSX_G(2) BSX_G(3) BSX_G(4) BSX_G(5) BSX_G(6) BSX_G(7) BSX_G(8) BSX_G(9)
#undef BSX_G
    default: break;
  }
}
hipError_t launch_gauss_blur(const uint8_t* src, uint8_t* dst, int w, int h, int ksize, int n, hipStream_t s) {
  GaussCoef gc;
  const int opts = gauss_opts(src, w);
  if (!gauss_coefficients(ksize, &gc, (opts & 1) ? (-(ksize / 2)) & 3 : 0)) return hipErrorInvalidValue;
  if (ksize == 1) return hipMemcpyAsync(dst, src, (size_t)n * w * h * 3, hipMemcpyDeviceToDevice, s);
  const bool words = gauss_words(src, dst, nullptr, w);
  for (int n0 = 0; n0 < n; n0 += kMaxGridY) {
    const int nn = n - n0 < kMaxGridY ? n - n0 : kMaxGridY;
    const dim3 grid((w + kGTW - 1) / kGTW, (h + kGTH - 1) / kGTH, nn);
    if (words) gauss_launch<1>(grid, s, src + (size_t)n0 * w * h * 3, dst + (size_t)n0 * w * h * 3, nullptr, w, h, gc, opts);
    else gauss_launch<0>(grid, s, src + (size_t)n0 * w * h * 3, dst + (size_t)n0 * w * h * 3, nullptr, w, h, gc, opts);
  }
  return hipGetLastError();
}
// out = alpha_blend(GaussianBlur(frames, ksize), frames, masks) in one pass over the frames (gauss_blur_k<2, ..>)
bool gauss_blend_fusable(const uint8_t* frames, const uint8_t* masks, const uint8_t* out, int w, int ksize) {
  return ksize >= 3 && ksize <= 2 * kGMaxR + 1 && (ksize & 1) && frames != out && gauss_words(frames, masks, out, w);
}
hipError_t launch_gauss_blend(const uint8_t* frames, const uint8_t* masks, uint8_t* out, int w, int h, int ksize, int n, hipStream_t s) {
  GaussCoef gc;
  const int opts = gauss_opts(frames, w);
  if (!gauss_coefficients(ksize, &gc, (opts & 1) ? (-(ksize / 2)) & 3 : 0) || !gauss_blend_fusable(frames, masks, out, w, ksize)) return hipErrorInvalidValue;
  for (int n0 = 0; n0 < n; n0 += kMaxGridY) {
    const int nn = n - n0 < kMaxGridY ? n - n0 : kMaxGridY;
    gauss_launch<2>(dim3((w + kGTW - 1) / kGTW, (h + kGTH - 1) / kGTH, nn), s, frames + (size_t)n0 * w * h * 3, out + (size_t)n0 * w * h * 3, masks + (size_t)n0 * w * h, w, h, gc, opts);
  }
  return hipGetLastError();
}

hipError_t launch_flip_bgr(const uint8_t* src, uint8_t* dst, int w, int h, int code, int n, hipStream_t s) {
  for (int n0 = 0; n0 < n; n0 += kMaxGridY) {
    const int nn = n - n0 < kMaxGridY ? n - n0 : kMaxGridY;
    flip_bgr_k<<<dim3(blocks_for((long)w * h), nn), kThreads, 0, s>>>(src + (size_t)n0 * w * h * 3, dst + (size_t)n0 * w * h * 3, w, h, code);
  }
  return hipGetLastError();
}

hipError_t launch_bgr_to_yuyv(const uint8_t* bgr, uint8_t* yuyv, int w, int h, int n, hipStream_t s) {
  long pairs = (long)n * w * h / 2;
  yuyv_k<<<blocks_for(pairs), kThreads, 0, s>>>(bgr, reinterpret_cast<uint32_t*>(yuyv), pairs);
  return hipGetLastError();
}

hipError_t launch_yuyv_to_bgr(const uint8_t* yuyv, uint8_t* bgr, int w, int h, int n, hipStream_t s) {
  long pairs = (long)n * w * h / 2;
  yuyv_to_bgr_k<<<blocks_for(pairs), kThreads, 0, s>>>(reinterpret_cast<const uint32_t*>(yuyv), bgr, pairs);
  return hipGetLastError();
}

hipError_t launch_fill_u8(uint8_t* p, uint8_t v, size_t bytes, hipStream_t s) {
  return hipMemsetAsync(p, v, bytes, s);
}

}  // namespace bsx
