// kernels_img.hip — integer/byte image kernels of the hot path for gfx950 (wave64).
//
// Every kernel here is bit-exact against the CPU oracle's restatement of the OpenCV 8-bit
// paths and of the reference's own loops; all are HBM-bound byte work, so the design rules
// are: 16-byte vector accesses per lane, LDS tiles for neighbourhood ops, no GEMM shapes.
//
//   prep_resize_k        libbackscrub.cc:285-290  ROI crop + cv::resize(INTER_LINEAR) + BGR2RGB
//   prep_bilateral_k     libbackscrub.cc:295-302  cv::bilateralFilter(5,100,100) + convertTo(CV_32FC3)
//   decode_k             libbackscrub.cc:317-357  argmax / threshold / softmax-2 + temporal IIR
//   mask_upscale_blur_k  libbackscrub.cc:367-371  cv::resize ↑ + cv::blur 5x5 into the persistent mask ROI
//   blend16_k            deepseg.cc:108-134       alpha_blend
//   resize_bgr_k         background.cc:186,190    cv::resize of the background
//   yuyv_k               deepseg.cc:87-106        convert_rgb_to_yuyv
#include "kernels.hpp"

namespace bsx {
namespace {

constexpr int kThreads = 256;
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
#pragma unroll
  for (int c = 0; c < CN; c++) {
    int h0 = r0[sx * CN + c] * a0 + r0[sx1 * CN + c] * a1;
    int h1 = r1[sx * CN + c] * a0 + r1[sx1 * CN + c] * a1;
    out[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  }
}

// ---- prep 1: frame ROI → model canvas, packed R | G<<8 | B<<16 (bars stay 0) -----------------
__global__ __launch_bounds__(kThreads) void prep_resize_k(const uint8_t* __restrict__ frames, int W, int H, Rect4 roi,
                                                         uint32_t* __restrict__ canvas, int inW, int inH, Rect4 q, ResizeTab tab, long total) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % inW);
  long r = i / inW;
  int y = (int)(r % inH);
  long n = r / inH;
  uint32_t v = 0;
  int dx = x - q.x, dy = y - q.y;
  if (dx >= 0 && dx < q.w && dy >= 0 && dy < q.h) {
    const uint8_t* src = frames + n * (long)W * H * 3 + ((long)roi.y * W + roi.x) * 3;
    int bgr[3];
    sample_linear<3>(src, (long)W * 3, tab, dx, dy, bgr);
    v = (uint32_t)bgr[2] | ((uint32_t)bgr[1] << 8) | ((uint32_t)bgr[0] << 16);  // BGR2RGB
  }
  canvas[i] = v;
}

// ---- prep 2: bilateral d=5 on the RGB canvas + u8→f32 normalise -----------------------------
// f32 accumulation in tap order with separate multiply and add (no FMA contraction), then
// cvRound(sum * (1/wsum)) — the association the oracle defines.
__global__ __launch_bounds__(kThreads) void prep_bilateral_k(const uint32_t* __restrict__ canvas, float* __restrict__ input, int inW, int inH,
                                                            BilateralParams bp, long total) {
  __shared__ float lut[768];
  for (int k = threadIdx.x; k < 768; k += kThreads) lut[k] = bp.color_lut[k];
  __syncthreads();
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % inW);
  long r = i / inW;
  int y = (int)(r % inH);
  long n = r / inH;
  const uint32_t* img = canvas + n * (long)inW * inH;
  uint32_t c0 = img[(long)y * inW + x];
  int r0 = c0 & 255, g0 = (c0 >> 8) & 255, b0 = (c0 >> 16) & 255;
  float sr = 0.f, sg = 0.f, sb = 0.f, ws = 0.f;
#pragma unroll
  for (int k = 0; k < 13; k++) {
    int yy = reflect101(y + bp.off_y[k], inH), xx = reflect101(x + bp.off_x[k], inW);
    uint32_t c = img[(long)yy * inW + xx];
    int rr = c & 255, gg = (c >> 8) & 255, bb = (c >> 16) & 255;
    float w = __fmul_rn(bp.space_w[k], lut[abs(rr - r0) + abs(gg - g0) + abs(bb - b0)]);
    sr = __fadd_rn(sr, __fmul_rn((float)rr, w));
    sg = __fadd_rn(sg, __fmul_rn((float)gg, w));
    sb = __fadd_rn(sb, __fmul_rn((float)bb, w));
    ws = __fadd_rn(ws, w);
  }
  ws = __fdiv_rn(1.f, ws);
  int qr = __float2int_rn(__fmul_rn(sr, ws)), qg = __float2int_rn(__fmul_rn(sg, ws)), qb = __float2int_rn(__fmul_rn(sb, ws));
  qr = min(max(qr, 0), 255); qg = min(max(qg, 0), 255); qb = min(max(qb, 0), 255);
  float* o = input + i * 3;
  o[0] = __fadd_rn(__fmul_rn((float)qr, bp.scale), bp.offset);
  o[1] = __fadd_rn(__fmul_rn((float)qg, bp.scale), bp.offset);
  o[2] = __fadd_rn(__fmul_rn((float)qb, bp.scale), bp.offset);
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
    float2 l = reinterpret_cast<const float2*>(t)[i];
    float e0 = (float)exp((double)l.x), e1 = (float)exp((double)l.y);  // correctly-rounded stand-in for libm expf
    float s = __fadd_rn(e0, e1);
    float p0 = __fdiv_rn(e0, s), p1 = __fdiv_rn(e1, s);
    val = p0 < p1 ? 0 : 255;
  }
  out[i] = (uint8_t)((val & 0xE0) | (out[i] >> 3));
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
// the LDS/ALU-bound mask phases of others.  (Used when the ROI is the whole frame.)
constexpr int kTW = 128, kTH = 32, kHW = kTW + 4, kHH = kTH + 4, kMaxSrcRows = 40;
__device__ __forceinline__ uint32_t blend4w(uint32_t a, uint32_t b, int m0, int m1, int m2, int m3);
template <bool BLEND>
__global__ __launch_bounds__(kThreads) void mask_upscale_blur_k(const uint8_t* __restrict__ ofinal, int outW, int outH, Rect4 q, ResizeTab tab,
                                                               uint8_t* __restrict__ mask, int W, int H, Rect4 roi,
                                                               const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ frames,
                                                               uint8_t* __restrict__ outp) {
  __shared__ int col_sx[kHW], col_sx1[kHW], col_a0[kHW], col_a1[kHW];
  __shared__ int row_s0[kHH], row_s1[kHH], row_b0[kHH], row_b1[kHH];
  __shared__ __attribute__((aligned(16))) uint16_t hq[kMaxSrcRows * kHW];
  __shared__ __attribute__((aligned(16))) uint8_t up[kHH * kHW + 8];
  __shared__ __attribute__((aligned(16))) uint16_t hs[kHH * kTW];
  __shared__ int s_min, s_max;
  const int n = blockIdx.z;
  const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * kTH;
  const uint8_t* src = ofinal + (long)n * outW * outH + (long)q.y * outW + q.x;
  const int tid = threadIdx.x;
  if (tid == 0) { s_min = 1 << 30; s_max = -1; }
  __syncthreads();
  // 1. column / row tables
  if (tid < kHW) {
    const int gx = reflect101(min(tx0 + tid - 2, roi.w + 1), roi.w);
    int sx, sx1, a0, a1;
    if (tab.mode == 1) { sx = sx1 = gx; a0 = 2048; a1 = 0; }
    else if (tab.mode == 2) { sx = 2 * gx; sx1 = 2 * gx + 1; a0 = a1 = 0; }
    else { sx = tab.xofs[gx]; sx1 = min(sx + 1, tab.sw - 1); a0 = tab.xa[2 * gx]; a1 = tab.xa[2 * gx + 1]; }
    col_sx[tid] = sx; col_sx1[tid] = sx1; col_a0[tid] = a0; col_a1[tid] = a1;
  } else if (tid >= 192 && tid < 192 + kHH) {
    const int r = tid - 192;
    const int gy = reflect101(min(ty0 + r - 2, roi.h + 1), roi.h);
    int s0, s1, b0, b1;
    if (tab.mode == 1) { s0 = s1 = gy; b0 = 2048; b1 = 0; }
    else if (tab.mode == 2) { s0 = 2 * gy; s1 = 2 * gy + 1; b0 = b1 = 0; }
    else { const int sy = tab.yofs[gy]; s0 = min(max(sy, 0), tab.sh - 1); s1 = min(max(sy + 1, 0), tab.sh - 1); b0 = tab.ya[2 * gy]; b1 = tab.ya[2 * gy + 1]; }
    row_s0[r] = s0; row_s1[r] = s1; row_b0[r] = b0; row_b1[r] = b1;
    atomicMin(&s_min, s0);
    atomicMax(&s_max, s1);
  }
  __syncthreads();
  const int smin = s_min, nsr = s_max - smin + 1;
  if (tab.mode == 0 && nsr <= kMaxSrcRows) {
    // 2. horizontal pass on the touched source rows
    for (int k = tid; k < nsr * kHW; k += kThreads) {
      const int r = k / kHW, x = k - r * kHW;
      const uint8_t* sr = src + (long)(smin + r) * outW;
      hq[k] = (uint16_t)((sr[col_sx[x]] * col_a0[x] + sr[col_sx1[x]] * col_a1[x]) >> 4);
    }
    __syncthreads();
    // 3. vertical pass, 4 pixels per lane (64-bit LDS reads of the two source rows, one 32-bit write)
    for (int k = tid; k < kHH * (kHW / 4); k += kThreads) {
      const int y = k / (kHW / 4), x = (k - y * (kHW / 4)) * 4;
      const int b0 = row_b0[y], b1 = row_b1[y];
      const uint2 r0 = *reinterpret_cast<const uint2*>(&hq[(row_s0[y] - smin) * kHW + x]);
      const uint2 r1 = *reinterpret_cast<const uint2*>(&hq[(row_s1[y] - smin) * kHW + x]);
      const int h0[4] = {(int)(r0.x & 0xffff), (int)(r0.x >> 16), (int)(r0.y & 0xffff), (int)(r0.y >> 16)};
      const int h1[4] = {(int)(r1.x & 0xffff), (int)(r1.x >> 16), (int)(r1.y & 0xffff), (int)(r1.y >> 16)};
      uint32_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) packed |= (uint32_t)((((b0 * h0[j]) >> 16) + ((b1 * h1[j]) >> 16) + 2) >> 2) << (8 * j);
      *reinterpret_cast<uint32_t*>(&up[y * kHW + x]) = packed;
    }
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
  // 4. horizontal 5-sums, 4 per lane: 8 consecutive bytes in, 4 u16 out
  for (int k = tid; k < kHH * (kTW / 4); k += kThreads) {
    const int ly = k / (kTW / 4), lx = (k - ly * (kTW / 4)) * 4;
    const uint2 v = *reinterpret_cast<const uint2*>(&up[ly * kHW + lx]);
    const int b[8] = {(int)(v.x & 255), (int)((v.x >> 8) & 255), (int)((v.x >> 16) & 255), (int)(v.x >> 24),
                      (int)(v.y & 255), (int)((v.y >> 8) & 255), (int)((v.y >> 16) & 255), (int)(v.y >> 24)};
    const int s0 = b[0] + b[1] + b[2] + b[3] + b[4];
    const int s1 = s0 - b[0] + b[5], s2 = s1 - b[1] + b[6], s3 = s2 - b[2] + b[7];
    *reinterpret_cast<uint2*>(&hs[ly * kTW + lx]) = make_uint2((uint32_t)s0 | ((uint32_t)s1 << 16), (uint32_t)s2 | ((uint32_t)s3 << 16));
  }
  __syncthreads();
  // 5. vertical 5-sums: 32 rows x 32 groups of 4 pixels = 1024 items, 4 per lane
  for (int k = tid; k < kTH * (kTW / 4); k += kThreads) {
    const int ly = k / (kTW / 4), lx = (k - ly * (kTW / 4)) * 4;
    const int gy = ty0 + ly, gx = tx0 + lx;
    if (gy >= roi.h || gx >= roi.w) continue;
    int sum[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const uint2 v = *reinterpret_cast<const uint2*>(&hs[(ly + r) * kTW + lx]);
      sum[0] += (int)(v.x & 0xffff); sum[1] += (int)(v.x >> 16); sum[2] += (int)(v.y & 0xffff); sum[3] += (int)(v.y >> 16);
    }
    uint32_t packed = 0;
    uint8_t vals[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      vals[j] = (uint8_t)((sum[j] + 12) / 25);
      packed |= (uint32_t)vals[j] << (8 * j);
    }
    uint8_t* dst = mask + (long)n * W * H + (long)(roi.y + gy) * W + roi.x + gx;
    if (gx + 3 < roi.w && ((uintptr_t)dst & 3) == 0) *reinterpret_cast<uint32_t*>(dst) = packed;
    else for (int j = 0; j < 4 && gx + j < roi.w; j++) dst[j] = vals[j];
    if constexpr (BLEND) {
      // roi == whole frame and W % 4 == 0 (checked by the launcher): 4 pixels = 12 bytes = 3 aligned words per image
      const long pix = (long)gy * W + gx;
      const uint32_t* ap = reinterpret_cast<const uint32_t*>(bg + (bg_stride ? n * bg_stride : 0) + pix * 3);
      const uint32_t* bp = reinterpret_cast<const uint32_t*>(frames + ((long)n * W * H + pix) * 3);
      uint32_t* op = reinterpret_cast<uint32_t*>(outp + ((long)n * W * H + pix) * 3);
      const uint32_t a0 = ap[0], a1 = ap[1], a2 = ap[2], b0 = bp[0], b1 = bp[1], b2 = bp[2];
      const int m0 = vals[0], m1 = vals[1], m2 = vals[2], m3 = vals[3];
      op[0] = blend4w(a0, b0, m0, m0, m0, m1);
      op[1] = blend4w(a1, b1, m1, m1, m2, m2);
      op[2] = blend4w(a2, b2, m2, m3, m3, m3);
    }
  }
}

// ---- alpha blend: 16 pixels (16 mask bytes, 48+48 source bytes, 48 output bytes) per lane -----
__device__ __forceinline__ uint32_t blend4(uint32_t a, uint32_t b, int m0, int m1, int m2, int m3);
__device__ __forceinline__ uint32_t blend4w(uint32_t a, uint32_t b, int m0, int m1, int m2, int m3) { return blend4(a, b, m0, m1, m2, m3); }
__device__ __forceinline__ uint32_t blend4(uint32_t a, uint32_t b, int m0, int m1, int m2, int m3) {
  // four consecutive bytes of the packed BGR stream; mX = mask of the pixel byte X belongs to
  uint32_t r;
  int a0 = a & 255, a1 = (a >> 8) & 255, a2 = (a >> 16) & 255, a3 = a >> 24;
  int b0 = b & 255, b1 = (b >> 8) & 255, b2 = (b >> 16) & 255, b3 = b >> 24;
  r = (uint32_t)((a0 * m0 + b0 * (255 - m0)) / 255);
  r |= (uint32_t)((a1 * m1 + b1 * (255 - m1)) / 255) << 8;
  r |= (uint32_t)((a2 * m2 + b2 * (255 - m2)) / 255) << 16;
  r |= (uint32_t)((a3 * m3 + b3 * (255 - m3)) / 255) << 24;
  return r;
}

__global__ __launch_bounds__(kThreads) void blend16_k(const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ fr,
                                                     const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, long groups_per_frame,
                                                     long npix, long total_groups) {
  long gi = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gi >= total_groups) return;
  long n = gi / groups_per_frame, g = gi % groups_per_frame;
  long pix = n * npix + g * 16;
  const uint4 mv = *reinterpret_cast<const uint4*>(mask + pix);
  const uint4* ap = reinterpret_cast<const uint4*>(bg + (bg_stride ? n * bg_stride : 0) + g * 48);
  const uint4* bp = reinterpret_cast<const uint4*>(fr + pix * 3);
  uint4* op = reinterpret_cast<uint4*>(out + pix * 3);
  uint32_t mw[4] = {mv.x, mv.y, mv.z, mv.w};
  uint32_t aw[12], bw[12], ow[12];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    uint4 a = ap[k], b = bp[k];
    aw[4 * k] = a.x; aw[4 * k + 1] = a.y; aw[4 * k + 2] = a.z; aw[4 * k + 3] = a.w;
    bw[4 * k] = b.x; bw[4 * k + 1] = b.y; bw[4 * k + 2] = b.z; bw[4 * k + 3] = b.w;
  }
  // word j of the 12 covers bytes 4j..4j+3 → pixels (4j)/3 .. (4j+3)/3 ; every 3 words = 4 pixels = 1 mask word
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int m0 = mw[q] & 255, m1 = (mw[q] >> 8) & 255, m2 = (mw[q] >> 16) & 255, m3 = mw[q] >> 24;
    ow[3 * q + 0] = blend4(aw[3 * q + 0], bw[3 * q + 0], m0, m0, m0, m1);
    ow[3 * q + 1] = blend4(aw[3 * q + 1], bw[3 * q + 1], m1, m1, m2, m2);
    ow[3 * q + 2] = blend4(aw[3 * q + 2], bw[3 * q + 2], m2, m3, m3, m3);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) op[k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
}

// scalar tail / unaligned fallback: one pixel per lane
__global__ __launch_bounds__(kThreads) void blend1_k(const uint8_t* __restrict__ bg, long bg_stride, const uint8_t* __restrict__ fr,
                                                    const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, long npix, long first, long total) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  long per = npix - first;
  long n = i / per, p = first + i % per;
  int m = mask[n * npix + p];
  const uint8_t* a = bg + (bg_stride ? n * bg_stride : 0) + p * 3;
  const uint8_t* b = fr + (n * npix + p) * 3;
  uint8_t* o = out + (n * npix + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = (uint8_t)((a[c] * m + b[c] * (255 - m)) / 255);
}

// ---- generic BGR resize -------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void resize_bgr_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, ResizeTab tab, long total) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % tab.dw);
  long r = i / tab.dw;
  int y = (int)(r % tab.dh);
  long n = r / tab.dh;
  int v[3];
  sample_linear<3>(src + n * (long)tab.sw * tab.sh * 3, (long)tab.sw * 3, tab, x, y, v);
  uint8_t* o = dst + i * 3;
  o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
}

// ---- BGR → YUYV (cv::cvtColor(COLOR_RGB2YUV) on BGR-ordered bytes, then 4:2:2 pack Y0 V Y1 U) -----
__device__ __forceinline__ void rgb2yuv(int R, int G, int B, int* Y, int* U, int* V) {
  const int shift = 14, half = 1 << 13, delta = 128 << 14;
  int y = (R * 4899 + G * 9617 + B * 1868 + half) >> shift;
  int u = ((B - y) * 8061 + delta + half) >> shift;
  int v = ((R - y) * 14369 + delta + half) >> shift;
  *Y = min(max(y, 0), 255); *U = min(max(u, 0), 255); *V = min(max(v, 0), 255);
}
__global__ __launch_bounds__(kThreads) void yuyv_k(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, long pairs) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= pairs) return;
  const uint8_t* p = in + i * 6;
  int y0, u0, v0, y1, u1, v1;
  rgb2yuv(p[0], p[1], p[2], &y0, &u0, &v0);
  rgb2yuv(p[3], p[4], p[5], &y1, &u1, &v1);
  uint32_t u = (uint32_t)((u0 + u1) / 2), v = (uint32_t)((v0 + v1) / 2);
  out[i] = (uint32_t)y0 | (v << 8) | ((uint32_t)y1 << 16) | (u << 24);
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

__global__ __launch_bounds__(kThreads) void fill_k(uint4* p, uint4 v, long n16) {
  long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i < n16) p[i] = v;
}

}  // namespace

hipError_t launch_prep_resize(const uint8_t* frames, int W, int H, Rect4 roi, uint32_t* canvas, int inW, int inH, Rect4 in_roi, ResizeTab tab,
                              int n, hipStream_t s) {
  long total = (long)n * inW * inH;
  prep_resize_k<<<blocks_for(total), kThreads, 0, s>>>(frames, W, H, roi, canvas, inW, inH, in_roi, tab, total);
  return hipGetLastError();
}

hipError_t launch_prep_bilateral(const uint32_t* canvas, float* input, int inW, int inH, BilateralParams bp, int n, hipStream_t s) {
  long total = (long)n * inW * inH;
  prep_bilateral_k<<<blocks_for(total), kThreads, 0, s>>>(canvas, input, inW, inH, bp, total);
  return hipGetLastError();
}

hipError_t launch_decode(int model_type, const float* logits, uint8_t* ofinal, int npix, int nch, int n, hipStream_t s) {
  long total = (long)n * npix;
  int type = model_type == 1 ? 1 : (model_type == 3 ? 3 : 2);
  decode_k<<<blocks_for(total), kThreads, 0, s>>>(type, logits, ofinal, total, nch);
  return hipGetLastError();
}

hipError_t launch_mask_upscale_blur(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H, Rect4 roi,
                                    int n, hipStream_t s) {
  dim3 grid((roi.w + kTW - 1) / kTW, (roi.h + kTH - 1) / kTH, n);
  mask_upscale_blur_k<false><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, nullptr, 0, nullptr, nullptr);
  return hipGetLastError();
}

bool mask_blend_fusable(int W, int H, Rect4 roi, const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* out) {
  return roi.x == 0 && roi.y == 0 && roi.w == W && roi.h == H && (W % 4) == 0 && (bg_stride % 4) == 0 &&
         ((((uintptr_t)bg) | ((uintptr_t)frames) | ((uintptr_t)out)) & 3) == 0;
}

hipError_t launch_mask_blend(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H, Rect4 roi,
                             const uint8_t* bg, size_t bg_stride, const uint8_t* frames, uint8_t* out, int n, hipStream_t s) {
  dim3 grid((roi.w + kTW - 1) / kTW, (roi.h + kTH - 1) / kTH, n);
  mask_upscale_blur_k<true><<<grid, kThreads, 0, s>>>(ofinal, outW, outH, in_roi, tab, mask, W, H, roi, bg, (long)bg_stride, frames, out);
  return hipGetLastError();
}

hipError_t launch_blend(const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* masks, uint8_t* out, size_t npix, int n,
                        hipStream_t s) {
  bool aligned = (((uintptr_t)bg | (uintptr_t)frames | (uintptr_t)masks | (uintptr_t)out) & 15) == 0 && (npix % 16 == 0) && (bg_stride % 16 == 0);
  long groups = aligned ? (long)(npix / 16) : 0;
  if (groups) {
    long total = groups * n;
    blend16_k<<<blocks_for(total), kThreads, 0, s>>>(bg, (long)bg_stride, frames, masks, out, groups, (long)npix, total);
  } else {
    long total = (long)npix * n;
    blend1_k<<<blocks_for(total), kThreads, 0, s>>>(bg, (long)bg_stride, frames, masks, out, (long)npix, 0, total);
  }
  return hipGetLastError();
}

hipError_t launch_resize_bgr(const uint8_t* src, uint8_t* dst, ResizeTab tab, int n, hipStream_t s) {
  long total = (long)n * tab.dw * tab.dh;
  resize_bgr_k<<<blocks_for(total), kThreads, 0, s>>>(src, dst, tab, total);
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
