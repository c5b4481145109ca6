// kernels.hpp — launch interface of the hand-written gfx950 kernels (kernels_nn.hip,
// kernels_img.hip).  Host-only declarations; everything takes an explicit hipStream_t.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "plan.hpp"
#include "segments.hpp"

namespace bsx {

// ---- network -------------------------------------------------------------------------
// Executes one fused step for `n` streams.  `arena` holds every activation tensor at
// arena + plan.tensor_off[t] * n_cap (frame i of tensor t at + i * elems(t)).
// The network input / output tensors live in their own batch-major buffers (net_in / net_out).
// weights16 / f16_terms: the split-f16 MFMA form of the large pointwise convolutions (Step::w16_off, plan.weights16):
// 3 = hi/lo split of both operands (f32-grade results, the default), 1 = plain f16 inputs (IoU-gated fast mode), 0 = f32 MFMA
// f16_terms: low nibble = MFMA terms (0: f32 MFMA kernels, 1: plain f16 operands, 3: split f16); bit 4 (with 1 term only) = the fused expand+depthwise
// kernels store their output as f16 and the project GEMM that consumes it reads f16 (opt-in reduced-precision storage, BSX_F16_GEMM=fast16)
hipError_t launch_step(const Step& st, const Plan& plan, float* arena, float* net_in, float* net_out, const float* weights, int n, int n_cap,
                       hipStream_t s, const uint16_t* weights16 = nullptr, int f16_terms = 0, const uint32_t* net_in_u8 = nullptr, float in_scale = 0.f, float in_offset = 0.f);
// net_in_u8: the network input as 8-bit pixels (prep_fused_k<2>) — taken by the fused DeepLab head (dl_head0_k<true>), which normalises on load
inline bool head0_u8_ok(const Plan& plan) {
  if (plan.steps.empty() || !plan.steps[0].fuse_head0) return false;
  const Step& st = plan.steps[0];
  return st.in0 == plan.input && (long)(2 * (head0_band_rows(st.W, st.OW) + 2) + 1) * st.W <= 4 * 3 * 512;
}

// Does the chain of three 1x1 convolutions around step `mid` (Step::chain_first / chain_last) run as ONE launch for n streams?  Asked by the three steps'
// launches and by the profile's slot labels: split-f16 MFMA mode, and enough pixels for the GEMM forms (below that the three steps run on their own).
inline bool chain3_on(const Plan& plan, int mid, int n, const uint16_t* weights16, int f16_terms) {
  if (mid < 0 || mid >= (int)plan.steps.size() || plan.steps[mid].chain_first < 0) return false;
  const Step& b = plan.steps[mid];
  const long M = (long)n * b.OH * b.OW;                 // (the kernel indexes its input with 32-bit element offsets: past 2^31 elements the three steps run on their own)
  return weights16 && (f16_terms & 15) == 3 && M >= 8192 && M * plan.steps[b.chain_first].Cin < (1l << 31);
}

// DeepLab tail: the graph's final RESIZE_BILINEAR fused with the 21-way argmax + temporal IIR (the full-resolution logits never exist)
bool resize_argmax_fusable(const Step& st);
// generic = the scalar first-maximum scan (what more than 24 classes take; tests force it for the 21-class graph)
hipError_t launch_resize_argmax_iir(const Step& st, const float* lowres_logits, uint8_t* ofinal, int n, hipStream_t s, bool generic = false);

// Whole-network per-frame program (kernels_frame.hip): one 1024-lane workgroup per stream.
hipError_t frame_program_prepare(int lds_floats);
hipError_t launch_frame_program(const MicroOp* d_ops, int n_ops, int lds_floats, float* arena, long per_frame_floats, float* net_in, float* net_out,
                                const float* weights, int n, hipStream_t s, unsigned long long* timeline = nullptr);

// Spatially-parallel segment kernels around the per-frame program (kernels_seg.hip, segments.hpp)
hipError_t seg_prepare();
hipError_t nn_prepare();          // dynamic-LDS limits of the fused per-launch kernels (kernels_nn.hip), for the current device
hipError_t launch_seg_head(const SegHead& d, float* arena, long per_frame, const void* net_in, const float* weights, int n, hipStream_t s, bool h16 = false, bool u8 = false,
                           float in_scale = 0.f, float in_offset = 0.f);
hipError_t launch_seg_k2(const SegK2& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s, bool h16 = false);
hipError_t launch_seg_k3(const SegK3& d, float* arena, long per_frame, const float* weights, int n, hipStream_t s, bool h16 = false);
// logits = true: write the network output tensor (debug / stage tests); false: decode + temporal IIR straight into `ofinal`
hipError_t launch_seg_gate(const SegGate& gt, float* arena, long per_frame, const float* weights, long long out_off, int n, hipStream_t s);
hipError_t launch_seg_tail(const SegTail& d, float* arena, long per_frame, float* net_out, uint8_t* ofinal, const float* weights, bool logits, int n, hipStream_t s, bool h16 = false);

// ---- image path ----------------------------------------------------------------------
// Fixed-point bilinear tables of cv::resize(INTER_LINEAR, 8u) for one (src,dst) size pair
// (device arrays; built on the host with the same float/double arithmetic OpenCV uses).
struct ResizeTab {
  const int* xofs = nullptr;     // [dw]  source column (already clamped)
  const short* xa = nullptr;     // [2*dw] horizontal coefficients (a0,a1), sum 2048
  const int* yofs = nullptr;     // [dh]  source row before clamping
  const short* ya = nullptr;     // [2*dh] vertical coefficients (b0,b1)
  int sw = 0, sh = 0, dw = 0, dh = 0;
  int mode = 0;                  // 0 linear, 1 copy (same size), 2 INTER_AREA 2x2 (both scales exactly 2)
  int tile_ok = 0;               // mask_tile_fits(): every mask tile's source block fits the LDS staging area
  // mask up-scale table only: device scratch of one byte per (stream, mask tile), written by tile_class_k right before the tile kernel reads it (1 = the tile's
  // whole source block is 0xFF, 2 = 0x00, 0 = anything else).  nullptr (BSX_NO_UNIFORM_TILES at bsx_new: A/B timing, parity tests) = every tile on the general path
  uint8_t* tile_class = nullptr;
};
bool mask_tile_fits(const int* xofs, const int* yofs, int sw, int sh, int dw, int dh);
int mask_tile_width();             // the mask tile kernel's tile geometry (kernels_img.hip: kTW x kTH)
int mask_tile_height();
// classify the mask tiles of n streams into tab.tile_class (see ResizeTab); launch_mask_upscale_blur / launch_mask_blend run it themselves
struct Rect4;
hipError_t launch_tile_class(const uint8_t* ofinal, int outW, int outH, const Rect4& in_roi, const ResizeTab& tab, const Rect4& roi, int n, hipStream_t s);

struct Rect4 { int x, y, w, h; };

struct BilateralParams {
  float space_w[13];
  int off_y[13], off_x[13];
  const float* color_lut;        // [768] device
  float scale, offset;
};

constexpr int kCanvasPad = 2;        // radius of the bilateral filter: the halo of prep_fused_k's tiles (BORDER_REFLECT_101)
bool bilateral_taps_match(const BilateralParams& bp);   // host table order == the kernel's hard-wired 13 taps
// frame ROI ↓ → model canvas → bilateral(5,100,100) → convertTo: ONE kernel, no canvas in memory.  libbackscrub.cc:285-302
// input (f32 [n][inH][inW][3]) and / or input_u8 (R|G<<8|B<<16 [n][inH][inW]): whichever is non-null is written
// yuyv_in: `frames` is YUYV 4:2:2 (2 B/px; BSX_STEP_YUYV_IN) — only where prep_yuyv_fusable() holds
hipError_t launch_prep_fused(const uint8_t* frames, int W, int H, Rect4 roi, float* input, uint32_t* input_u8, int inW, int inH, Rect4 in_roi, ResizeTab tab,
                             BilateralParams bp, int n, hipStream_t s, bool yuyv_in = false);
bool prep_yuyv_fusable(int W, Rect4 roi, const ResizeTab& tab);
// decode + temporal IIR on the model-resolution mask.  libbackscrub.cc:317-357
hipError_t launch_decode(int model_type, const float* logits, uint8_t* ofinal, int npix, int nch, int n, hipStream_t s);
// ofinal(in_roi) ↑ roi size, 5x5 box blur (REFLECT_101 on the ROI), write into mask(roi).  libbackscrub.cc:367-371
hipError_t launch_mask_upscale_blur(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H,
                                    Rect4 roi, int n, hipStream_t s);
// mask upscale + blur AND alpha blend of the same tile in one launch (W, roi.x, roi.w multiples of 4, 4-byte aligned images;
// pixels outside the ROI — mask 255 forever — get the background copied)
bool mask_blend_fusable(int W, int H, Rect4 roi, const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* out);
hipError_t launch_mask_blend(const uint8_t* ofinal, int outW, int outH, Rect4 in_roi, ResizeTab tab, uint8_t* mask, int W, int H, Rect4 roi,
                             const uint8_t* bg, size_t bg_stride, const uint8_t* frames, uint8_t* out, int n, hipStream_t s, int yuyv = 0, int lds_pad = 0);
// alpha blend.  deepseg.cc:108-134
hipError_t launch_blend(const uint8_t* bg, size_t bg_stride, const uint8_t* frames, const uint8_t* masks, uint8_t* out, size_t npix,
                        int n, hipStream_t s);
// generic BGR resize (background → frame size).  background.cc:186,190
hipError_t launch_resize_bgr(const uint8_t* src, uint8_t* dst, ResizeTab tab, int n, hipStream_t s);
// BGR → YUYV.  deepseg.cc:87-106
hipError_t launch_bgr_to_yuyv(const uint8_t* bgr, uint8_t* yuyv, int w, int h, int n, hipStream_t s);
// YUYV → BGR ingest.  deepseg.cc:553 (CAP_PROP_CONVERT_RGB), :725
hipError_t launch_yuyv_to_bgr(const uint8_t* yuyv, uint8_t* bgr, int w, int h, int n, hipStream_t s);
// cv::flip of the composited frame (code as cv::flip: 0 vertical, >0 horizontal, <0 both).  deepseg.cc:667-673
hipError_t launch_flip_bgr(const uint8_t* src, uint8_t* dst, int w, int h, int code, int n, hipStream_t s);
// cv::GaussianBlur(Size(ksize, ksize), sigma 0) of packed BGR u8 images (dst != src), ksize odd <= 31.  deepseg.cc:657-658 (-p bgblur:N)
hipError_t launch_gauss_blur(const uint8_t* src, uint8_t* dst, int w, int h, int ksize, int n, hipStream_t s);
bool gauss_coeff_words(int ksize, int shift, uint32_t* c4 /* [4][9] */, uint32_t* c2 /* [2][17] */);   // host: the tables launch_gauss_* pass to the kernel
// blur + alpha blend of the frames over their own blur (deepseg.cc:652-661 without -b), the blurred image never stored; fusable = 4-byte aligned images, w % 4 == 0
bool gauss_blend_fusable(const uint8_t* frames, const uint8_t* masks, const uint8_t* out, int w, int ksize);
hipError_t launch_gauss_blend(const uint8_t* frames, const uint8_t* masks, uint8_t* out, int w, int h, int ksize, int n, hipStream_t s);
// fill
hipError_t launch_fill_u8(uint8_t* p, uint8_t v, size_t bytes, hipStream_t s);

}  // namespace bsx
