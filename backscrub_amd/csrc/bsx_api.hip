// bsx_api.hip — the C ABI of libbsx.so (include/bsx.h): context, geometry, per-batch launch
// sequence.  Host-side mirror of lib/libbackscrub.cc (bs_maskgen_new :161-259,
// bs_maskgen_process :279-376, bs_maskgen_delete :261-277) and of the compositing step of
// app/deepseg.cc (:108-134, :649-661), re-designed for batches of independent streams
// resident in HBM.
#include "debug_switches.hpp"
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bsx.h"
#include "gen_mid.hpp"
#include "gen_seg.hpp"
#include "kernels.hpp"
#include "roctx_ranges.hpp"
#include "rtc.hpp"
#include "plan.hpp"
#include "tflite_model.hpp"

using namespace bsx;

namespace {
thread_local std::string g_last_error;

struct HostResizeTab {
  std::vector<int> xofs, yofs;
  std::vector<short> xa, ya;
  int sw = 0, sh = 0, dw = 0, dh = 0, mode = 0;
};

inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline short sat_short(long v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// Coefficient tables of cv::resize(INTER_LINEAR) for 8-bit images (OpenCV imgproc resize.cpp):
// scale = 1/(dst/src) in double; fx = (float)((dx+0.5)*scale-0.5); sx = floor(fx); fx -= sx;
// left/right clamps zero the fraction; coefficients = saturate_cast<short>(w*2048) (round-half-even).
HostResizeTab make_resize_tab(int sw, int sh, int dw, int dh) {
  HostResizeTab t;
  t.sw = sw; t.sh = sh; t.dw = dw; t.dh = dh;
  if (sw == dw && sh == dh) { t.mode = 1; return t; }
  double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
  long isx = lrint(scale_x), isy = lrint(scale_y);
  bool area_fast = std::fabs(scale_x - (double)isx) < 2.220446049250313e-16 && std::fabs(scale_y - (double)isy) < 2.220446049250313e-16;
  if (area_fast && isx == 2 && isy == 2) { t.mode = 2; return t; }
  t.xofs.resize(dw); t.xa.resize(2 * (size_t)dw); t.yofs.resize(dh); t.ya.resize(2 * (size_t)dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    t.xofs[dx] = sx;
    t.xa[2 * dx] = sat_short(lrintf((1.f - fx) * 2048.f));
    t.xa[2 * dx + 1] = sat_short(lrintf(fx * 2048.f));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    t.yofs[dy] = sy;
    t.ya[2 * dy] = sat_short(lrintf((1.f - fy) * 2048.f));
    t.ya[2 * dy + 1] = sat_short(lrintf(fy * 2048.f));
  }
  return t;
}

struct DevResizeTab {
  ResizeTab tab;
  void* mem = nullptr;
};

}  // namespace

struct bsx_ctx {
  // callbacks (lib/libbackscrub.cc:37-41)
  bsx_debug_fn ondebug = nullptr;
  bsx_stage_fn onprep = nullptr, oninfer = nullptr, onmask = nullptr;
  void* caller_ctx = nullptr;
  // model
  Graph graph;
  Plan plan;
  int model_type = BSX_MODEL_UNKNOWN;
  float norm_scale = 0, norm_offset = 0;
  // geometry
  int width = 0, height = 0, n_streams = 0, device = 0;
  int inW = 0, inH = 0, inC = 0, outW = 0, outH = 0, outC = 0;
  Rect4 roi{}, in_roi{};
  size_t threads = 0;
  // device state
  hipStream_t own_stream = nullptr;
  float* d_arena = nullptr;
  float* d_net_in = nullptr;        // network input  [n][inH][inW][inC] f32 (written by the prep kernels)
  uint32_t* d_net_in_u8 = nullptr;  // the same tensor before convertTo: [n][inH][inW] R | G<<8 | B<<16 — what the stems with a byte path read (in_u8)
  // bs_maskgen_process for ONE frame is a chain of 7 (Meet / MLKit) to ~45 (DeepLab) dependent launches of a few microseconds each: launch-bound.  The chain of
  // a stream slot is captured once into a hipGraph and replayed (BSX_NO_GRAPH=1, or any stage callback — they need host synchronisation between stages — keeps
  // the direct launches).  graph_state: 0 = not tried, 1 = usable, -1 = capture failed on this runtime (direct launches from then on).
  std::map<int, hipGraphExec_t> host_graphs;
  int graph_state = 0;
  bool in_u8 = false;               // the step's prep writes ONLY the 8-bit form and the stem normalises on load (seg_head_k / dl_head0_k; bit-identical).
                                    //   BSX_F32_INPUT=1 (read at bsx_new) keeps the f32 tensor for A/B timing; the stage-debug entry writes both.
  float* d_net_out = nullptr;       // network output [n][outH][outW][outC] f32 (read by the decode kernel)
  float* d_weights = nullptr;
  uint16_t* d_weights16 = nullptr;   // split-f16 copies of the large pointwise-conv weights (Plan::weights16)
  int f16_terms = 3;                 // per-launch path: 3 = split-f16 MFMA GEMM (f32-grade, default), 1 = plain f16 inputs (BSX_F16_GEMM=fast), 0 = f32 MFMA (BSX_F16_GEMM=off)
  uint8_t* d_ofinal = nullptr;
  uint8_t* d_masks = nullptr;
  uint8_t* d_host_frame = nullptr;  // staging for bsx_process_host
  uint8_t* d_bgr_scratch = nullptr; // BGR composite of bsx_step_batch_yuyv / _ex when the fused epilogue does not apply (lazy)
  uint8_t* d_bgr_scratch2 = nullptr; // ... its flipped copy when a YUYV pack follows (lazy)
  uint8_t* d_bgblur_scratch = nullptr; // BSX_STEP_BGBLUR when the single pass does not apply: the blurred frames (lazy)
  uint8_t* d_bgr_in_scratch = nullptr; // BSX_STEP_YUYV_IN where the fused kernels do not apply: the batch converted to BGR (lazy)
  float* d_color_lut = nullptr;
  MicroOp* d_program = nullptr;     // per-frame network program (kernels_frame.hip)
  bool use_program = false;
  RtcKernel mid;                    // the same program as ONE graph-specialised kernel, compiled by hipRTC when the context is created
  RtcKernel seg_mod;                // the segment kernels specialised to this graph (gen_seg.cpp; one hipRTC module: seg_mod.fn = bsx_seg_head, seg_fn[] = k2, k3, tail)
  hipFunction_t seg_fn[3] = {nullptr, nullptr, nullptr};
  std::string seg_note;
  std::string mid_note;             //   (gen_mid.cpp, mid_prelude.hip); mid.fn == nullptr: the interpreter runs (BSX_NO_RTC=1, or why in mid_note)
  BilateralParams bilateral{};
  DevResizeTab tab_down, tab_up;
  uint8_t* d_tile_class = nullptr;    // [n_streams][mask tiles]: 1 / 2 = the tile's source block is all 0xFF / 0x00 (tile_class_k), read by mask_tile_k
  size_t tiles_per_frame = 0;
  std::map<std::pair<std::pair<int, int>, std::pair<int, int>>, DevResizeTab> bg_tabs;
  std::string last_error, plan_text;
  bool keep_logits = false;            // BSX_KEEP_LOGITS: segmented plans write the logits and run the stand-alone decode (A/B, debugging)
  bool no_mask_blend_fusion = false;   // BSX_NO_MASK_BLEND_FUSION, read once at bsx_new (no getenv on the per-step path)
  bool no_bgblur_fusion = false;       // BSX_NO_BGBLUR_FUSION: BSX_STEP_BGBLUR always as blur pass + step (the A/B switch of the single-pass form)
  bool no_mask_tile = false;           // BSX_NO_MASK_TILE (tests: the generic mask kernel), likewise
  bool no_uniform_tiles = false;       // BSX_NO_UNIFORM_TILES (A/B timing, tests: every mask tile on the general path), likewise
  bool tail_generic = false;           // BSX_TAIL_GENERIC (tests: the scalar argmax scan of the DeepLab tail), likewise
  // Lanes (BSX_LANES=k, experiment): the fused step splits its batch into k contiguous groups of streams and runs each group's launch sequence on its own
  // HIP stream — streams are independent, so the HBM-bound tail of one group (mask + blend) can overlap the latency-bound network kernels of another.
  int lanes = 1;
  hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  // Two-deep pipeline (bsx_step_batch_pipelined): the composite of the batch handed over by the PREVIOUS call (mask up-scale + blur + alpha blend: HBM-bound) runs
  // on comp_stream while the caller's stream runs the mask pipeline of THIS call's batch (prep → network: latency-bound) — the batch-level form of the reference's
  // CalcMask worker thread next to its capture / blend loop (app/deepseg.cc:159-285, 634-661).  The one shared object is the model-resolution temporal state:
  // the kernel that advances it (tail / argmax tail / decode) waits for ev_pcomp first (state_write_fence).
  struct PendingComposite { bool active = false; const uint8_t* frames = nullptr; const uint8_t* bg = nullptr; size_t bg_stride = 0; uint8_t* out = nullptr; int n = 0; unsigned flags = 0; } pend;
  hipStream_t comp_stream = nullptr;
  hipEvent_t ev_pdone = nullptr, ev_pcomp = nullptr;   // ev_pdone: end of the mask pipeline of the PENDING batch, recorded on the stream of the call that enqueued it — the composite
                                                       // (and the next call's network) wait for THAT event, so a caller may alternate streams between calls
  hipEvent_t wait_before_state = nullptr;   // consumed by the next launch that writes d_ofinal
  int pipe_wgs = 0;                         // BSX_PIPE_WGS (read at bsx_new, experiment): workgroups per CU the pipelined composite may hold (occupancy cap by an LDS pad; 0 = no cap,
                                            // the default: every cap measured slower — the tile kernel needs its waves, profiles/r04l)
  bool act16 = false;                  // BSX_ACT16=1: 16-bit activation STORAGE for the segmented Meet / MLKit networks (g1) — opt-in, IoU-gated; needs the specialised middle kernel

  // stream-0 view of a graph tensor (network input/output have dedicated buffers; intermediates are batch-major in
  // the per-launch path and frame-major — frame 0 first — in the per-frame program)
  float* tensor_ptr(int t) const {
    if (t == plan.input) return d_net_in;
    if (t == plan.output) return d_net_out;
    return use_program ? d_arena + plan.tensor_off[t] : d_arena + (size_t)plan.tensor_off[t] * (size_t)n_streams;
  }
};

namespace {

// HIP's current device is per THREAD and defaults to 0: the reference application creates the context on its main thread
// (app/deepseg.cc:246) and calls bs_maskgen_process on a worker thread (:203,258), so every entry point that touches the
// GPU switches to the context's device for its duration and restores the caller's device afterwards.
struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

void report(bsx_ctx* c, bsx_debug_fn fn, void* user, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->last_error = buf;
  g_last_error = buf;
  if (fn) fn(user, buf); else fputs(buf, stderr);   // same routing as _dbg(), lib/libbackscrub.cc:72-83
}

#define BSX_HIP(c, expr)                                                                                    \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      report((c), (c) ? (c)->ondebug : nullptr, (c) ? (c)->caller_ctx : nullptr, "error: HIP %s at %s:%d (%s)\n", \
             hipGetErrorString(e_), __FILE__, __LINE__, #expr);                                             \
      return BSX_EDEVICE;                                                                                   \
    }                                                                                                       \
  } while (0)

int upload_tab(bsx_ctx* c, const HostResizeTab& h, DevResizeTab* d) {
  d->tab.sw = h.sw; d->tab.sh = h.sh; d->tab.dw = h.dw; d->tab.dh = h.dh; d->tab.mode = h.mode;
  if (h.mode != 0) return BSX_OK;
  d->tab.tile_ok = (mask_tile_fits(h.xofs.data(), h.yofs.data(), h.sw, h.sh, h.dw, h.dh) && !c->no_mask_tile) ? 1 : 0;
  d->tab.tile_class = nullptr;
  size_t b_xofs = h.xofs.size() * 4, b_yofs = h.yofs.size() * 4, b_xa = h.xa.size() * 2, b_ya = h.ya.size() * 2;
  auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
  size_t total = up16(b_xofs) + up16(b_yofs) + up16(b_xa) + up16(b_ya);
  BSX_HIP(c, hipMalloc(&d->mem, total));
  char* p = (char*)d->mem;
  BSX_HIP(c, hipMemcpy(p, h.xofs.data(), b_xofs, hipMemcpyHostToDevice)); d->tab.xofs = (const int*)p; p += up16(b_xofs);
  BSX_HIP(c, hipMemcpy(p, h.yofs.data(), b_yofs, hipMemcpyHostToDevice)); d->tab.yofs = (const int*)p; p += up16(b_yofs);
  BSX_HIP(c, hipMemcpy(p, h.xa.data(), b_xa, hipMemcpyHostToDevice)); d->tab.xa = (const short*)p; p += up16(b_xa);
  BSX_HIP(c, hipMemcpy(p, h.ya.data(), b_ya, hipMemcpyHostToDevice)); d->tab.ya = (const short*)p;
  return BSX_OK;
}

// The graph-specialised SEGMENT kernels of `plan` (gen_seg.cpp), compiled or fetched from the cache for `arch`: "" and the code object, or why there is none
std::string build_seg_kernels(const Plan& plan, bool h16, bool u8in, const std::string& arch, std::vector<char>* code, bool* cached, size_t* src_bytes = nullptr) {
  std::string why, log;
  if (BSX_DBG_ENV("BSX_NO_SEG_RTC")) return "ahead-of-time kernels (BSX_NO_SEG_RTC)";
  const std::string src = generate_seg_source(plan, h16, u8in, &why);
  if (src.empty()) return "ahead-of-time kernels (" + why + ")";
  if (src_bytes) *src_bytes = src.size();
  if (!rtc_build(src, arch, code, &log, cached)) { if (BSX_DBG_ENV("BSX_RTC_DEBUG")) fprintf(stderr, "%s\n", log.c_str()); return "ahead-of-time kernels (hipRTC: " + log.substr(0, 600) + ")"; }
  return "";
}

// The graph-specialised middle kernel of `plan`, compiled (or fetched from the cache) for `arch` — in the form that spills least.  The plain form lets the compiler share
// every lane-derived value between ops; where that pushes the kernel into scratch (MLKit: 352 bytes at 128 registers) the opaque-lane-index form (mid_prelude.hip:
// tid_now; 92 registers, no scratch) is compiled too and taken if its scratch is smaller — read from the code objects' kernel descriptors, so the choice needs no GPU and
// bsx_model_precompile makes the same one a context makes later (both code objects sit in the cache).  Returns "" and fills *code, or the reason there is no kernel.
struct MidBuild { std::vector<char> code; std::string source, note; bool cached = false, opaque_tid = false; long scratch = -1; };
std::string build_mid_kernel(const Plan& plan, bool act16, const std::string& arch, MidBuild* out) {
  std::string why, log;
  int force = -1;                                                 // debug build: BSX_RTC_TID=0 | 1 forces the plain / the opaque form (A/B timing)
  if (const char* e = BSX_DBG_ENV("BSX_RTC_TID")) force = atoi(e) != 0;
  auto build = [&](bool opaque, MidBuild* b) -> std::string {
    b->source = generate_mid_source(plan, &why, act16, opaque);
    if (b->source.empty()) return "interpreted (" + why + ")";
    if (!rtc_build(b->source, arch, &b->code, &log, &b->cached)) { if (BSX_DBG_ENV("BSX_RTC_DEBUG")) fprintf(stderr, "%s\n", log.c_str()); return "interpreted (hipRTC: " + log.substr(0, 400) + ")"; }
    b->opaque_tid = opaque;
    b->scratch = code_object_scratch_bytes(b->code, "bsx_mid");
    return "";
  };
  std::string err = build(force == 1, out);
  if (!err.empty() || force >= 0 || out->scratch <= 0) return err;
  MidBuild alt;
  if (build(true, &alt).empty() && alt.scratch >= 0 && alt.scratch < out->scratch) *out = std::move(alt);
  return "";
}

int model_type_from_name(const std::string& n) {  // lib/libbackscrub.cc:116-130 (same precedence)
  if (n.find("body-pix") != n.npos) return BSX_MODEL_BODYPIX;
  if (n.find("deeplab") != n.npos) return BSX_MODEL_DEEPLAB;
  if (n.find("segm_") != n.npos) return BSX_MODEL_MEET;
  if (n.find("selfie") != n.npos) return BSX_MODEL_MLKIT;
  return BSX_MODEL_UNKNOWN;
}

int init_device_state(bsx_ctx* c) {
  BSX_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  const size_t N = (size_t)c->n_streams;
  BSX_HIP(c, hipMalloc(&c->d_arena, c->plan.arena_floats_per_stream * N * sizeof(float)));
  BSX_HIP(c, hipMalloc(&c->d_net_in, N * c->inW * c->inH * c->inC * sizeof(float)));
  BSX_HIP(c, hipMalloc(&c->d_net_in_u8, N * c->inW * c->inH * sizeof(uint32_t)));
  BSX_HIP(c, hipMalloc(&c->d_net_out, N * c->outW * c->outH * c->outC * sizeof(float)));
  BSX_HIP(c, hipMalloc(&c->d_weights, std::max<size_t>(c->plan.weights.size(), 4) * sizeof(float)));
  BSX_HIP(c, hipMemcpy(c->d_weights, c->plan.weights.data(), c->plan.weights.size() * sizeof(float), hipMemcpyHostToDevice));
  BSX_HIP(c, hipMalloc(&c->d_weights16, std::max<size_t>(c->plan.weights16.size(), 8) * sizeof(uint16_t)));
  BSX_HIP(c, hipMemcpy(c->d_weights16, c->plan.weights16.data(), c->plan.weights16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  // off = f32 MFMA; fast = plain f16 operands (1 term); fast16 = fast + the depthwise outputs of the fused blocks STORED as f16 (bit 4: kernels.hpp)
  if (const char* m = getenv("BSX_F16_GEMM")) c->f16_terms = !strcmp(m, "off") ? 0 : (!strcmp(m, "fast") ? 1 : (!strcmp(m, "fast16") ? 17 : 3));
  // The per-frame program pays off when most tensors stay in LDS (Meet / MLKit families); graphs whose tensors mostly
  // spill (DeepLab: 33x33x480) run faster as one batch-wide launch per step.  BSX_FORCE_FRAME_PROGRAM / BSX_NO_FRAME_PROGRAM override.
  c->use_program = !c->plan.program.empty() && BSX_DBG_ENV("BSX_NO_FRAME_PROGRAM") == nullptr &&
                   (c->plan.seg.on || c->plan.program_lds_tensors >= c->plan.program_global_tensors || BSX_DBG_ENV("BSX_FORCE_FRAME_PROGRAM") != nullptr);
  if (c->use_program) {
    BSX_HIP(c, hipMalloc(&c->d_program, c->plan.program.size() * sizeof(MicroOp)));
    BSX_HIP(c, hipMemcpy(c->d_program, c->plan.program.data(), c->plan.program.size() * sizeof(MicroOp), hipMemcpyHostToDevice));
    BSX_HIP(c, frame_program_prepare(c->plan.program_lds_floats));
    if (c->plan.seg.on) BSX_HIP(c, seg_prepare());
    // Specialise the program to this graph: straight-line code with compile-time geometry instead of the interpreted micro-op table.
    // Compiled by hipRTC for this device's architecture (cached on disk; bsx_model_precompile fills the cache without a GPU).
    // Anything the generator does not cover, or a failed compilation, leaves the interpreter in charge — never an error.
    const char* a16 = getenv("BSX_ACT16");
    c->act16 = a16 && atoi(a16) != 0 && c->plan.seg.on;
    if (!BSX_DBG_ENV("BSX_NO_RTC")) {
      hipDeviceProp_t prop;
      MidBuild mb;
      if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) c->mid_note = "interpreted (no device properties)";
      else {
        const std::string err = build_mid_kernel(c->plan, c->act16, prop.gcnArchName, &mb);
        if (!err.empty()) c->mid_note = err;
        else if (rtc_load(mb.code, "bsx_mid", &c->mid) != hipSuccess) { c->mid_note = "interpreted (code object did not load)"; (void)hipGetLastError(); }
        else c->mid_note = std::string("specialised kernel (hipRTC") + (mb.cached ? ", from the cache" : ", compiled now") + (mb.opaque_tid ? ", lane indices re-derived per op" : "") +
                           (mb.scratch > 0 ? ", " + std::to_string(mb.scratch) + " B of scratch" : "") + ")";
      }
    } else c->mid_note = "interpreted (BSX_NO_RTC)";
    if (c->act16 && !c->mid.fn) {            // the interpreter has f32 tensors only: the mode needs the generated kernel
      c->last_error = "BSX_ACT16: the specialised middle kernel is not available (" + c->mid_note + ")";
      return BSX_EDEVICE;
    }
    if (c->act16) c->mid_note += ", 16-bit activation storage";
  } else {
    BSX_HIP(c, nn_prepare());                 // per-launch path: the fused kernels' dynamic-LDS limits on this device
  }
  if (const char* l = BSX_DBG_ENV("BSX_PIPE_WGS")) c->pipe_wgs = std::min(8, std::max(0, atoi(l)));
  if (const char* l = BSX_DBG_ENV("BSX_LANES")) c->lanes = std::min(4, std::max(1, atoi(l)));
  if (c->lanes > 1) {
    BSX_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int k = 1; k < c->lanes; k++) {
      BSX_HIP(c, hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
      BSX_HIP(c, hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
    }
  }
  // stems with a byte path take the 8-bit network input: the segmented Meet / MLKit head and DeepLab's fused head kernel
  if (BSX_DBG_ENV("BSX_NO_GRAPH")) c->graph_state = -1;
  c->in_u8 = BSX_DBG_ENV("BSX_F32_INPUT") == nullptr && ((c->use_program && c->plan.seg.on) || (!c->use_program && head0_u8_ok(c->plan)));
  // The segment kernels specialised to this graph (gen_seg.cpp): the same source as the ahead-of-time kernels with this plan's descriptors as compile-time constants,
  // compiled by hipRTC (cached on disk; bsx_model_precompile fills the cache without a GPU).  Anything that goes wrong leaves the ahead-of-time kernels in charge.
  if (c->use_program && c->plan.seg.on) {
    if (BSX_DBG_ENV("BSX_NO_RTC")) c->seg_note = "ahead-of-time kernels (BSX_NO_RTC)";
    else {
      hipDeviceProp_t prop;
      std::vector<char> code;
      bool cached = false;
      if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) c->seg_note = "ahead-of-time kernels (no device properties)";
      else {
        c->seg_note = build_seg_kernels(c->plan, c->act16, c->in_u8, prop.gcnArchName, &code, &cached);
        if (c->seg_note.empty()) {
          if (rtc_load(code, "bsx_seg_head", &c->seg_mod) != hipSuccess || rtc_function(c->seg_mod, "bsx_seg_k2", &c->seg_fn[0]) != hipSuccess ||
              rtc_function(c->seg_mod, "bsx_seg_k3", &c->seg_fn[1]) != hipSuccess || rtc_function(c->seg_mod, "bsx_seg_tail", &c->seg_fn[2]) != hipSuccess) {
            rtc_unload(&c->seg_mod);
            (void)hipGetLastError();
            c->seg_note = "ahead-of-time kernels (code object did not load)";
          } else c->seg_note = std::string("specialised kernels (hipRTC") + (cached ? ", from the cache)" : ", compiled now)");
        }
      }
    }
  }
  BSX_HIP(c, hipMalloc(&c->d_ofinal, N * c->outW * c->outH));
  BSX_HIP(c, hipMalloc(&c->d_masks, N * c->width * c->height));
  BSX_HIP(c, hipMemset(c->d_ofinal, 0, N * c->outW * c->outH));            // :257 leaves it uninitialised; defined as 0
  BSX_HIP(c, hipMemset(c->d_masks, 255, N * c->width * c->height));        // :248
  // bilateralFilter(d=5, sigmaColor=100, sigmaSpace=100) tables (OpenCV bilateral_filter):
  {
    const double sigma_color = 100.0, sigma_space = 100.0;
    const double gc = -0.5 / (sigma_color * sigma_color), gs = -0.5 / (sigma_space * sigma_space);
    const int radius = 2;
    std::vector<float> lut(768);
    for (int i = 0; i < 768; i++) lut[i] = (float)std::exp(i * i * gc);
    BSX_HIP(c, hipMalloc(&c->d_color_lut, 768 * sizeof(float)));
    BSX_HIP(c, hipMemcpy(c->d_color_lut, lut.data(), 768 * sizeof(float), hipMemcpyHostToDevice));
    int k = 0;
    for (int i = -radius; i <= radius; i++) for (int j = -radius; j <= radius; j++) {
      double r = std::sqrt((double)i * i + (double)j * j);
      if (r > radius) continue;
      c->bilateral.space_w[k] = (float)std::exp(r * r * gs);
      c->bilateral.off_y[k] = i; c->bilateral.off_x[k] = j;
      k++;
    }
    c->bilateral.color_lut = c->d_color_lut;
    c->bilateral.scale = c->norm_scale; c->bilateral.offset = c->norm_offset;
    if (k != 13 || !bilateral_taps_match(c->bilateral)) { c->last_error = "bilateral tap table does not match the kernel"; return BSX_EDEVICE; }
  }
  int rc = upload_tab(c, make_resize_tab(c->roi.w, c->roi.h, c->in_roi.w, c->in_roi.h), &c->tab_down);
  if (rc) return rc;
  rc = upload_tab(c, make_resize_tab(c->in_roi.w, c->in_roi.h, c->roi.w, c->roi.h), &c->tab_up);
  if (rc) return rc;
  // one class byte per (stream, mask tile) for the tile kernel's uniform-tile shortcut (kernels_img.hip: tile_class_k); absent = every tile on the general path
  c->tiles_per_frame = (size_t)((c->roi.w + mask_tile_width() - 1) / mask_tile_width()) * (size_t)((c->roi.h + mask_tile_height() - 1) / mask_tile_height());
  if (!c->no_uniform_tiles && c->tab_up.tab.mode == 0 && c->tab_up.tab.tile_ok) {
    BSX_HIP(c, hipMalloc(&c->d_tile_class, N * c->tiles_per_frame + 4));          // (+4: the tile kernel reads the aligned word around a byte)
    BSX_HIP(c, hipMemset(c->d_tile_class, 0, N * c->tiles_per_frame + 4));
    c->tab_up.tab.tile_class = c->d_tile_class;
  }
  // canvas outside in_roi is written as 0 by the prep kernel on every frame (the reference keeps a
  // persistent zeroed in_u8_bgr, :251); nothing else to initialise.
  return BSX_OK;
}

// NULL means the HIP default (null) stream, exactly as in every HIP API; the context's own
// non-blocking stream is only used by the synchronous host path.
hipStream_t pick(bsx_ctx*, void* s) { return (hipStream_t)s; }

// with_f32: also materialise the f32 input tensor when the stem reads the 8-bit form (the stage-debug entry: tests inspect the tensor)
int run_prep(bsx_ctx* c, const uint8_t* d_frames, int n, hipStream_t s, bool with_f32 = false, bool yuyv_in = false) {
  bsx_roctx::Range range("bsx:prep");
  float* f32 = (!c->in_u8 || with_f32) ? c->tensor_ptr(c->plan.input) : nullptr;
  uint32_t* u8 = c->in_u8 ? c->d_net_in_u8 : nullptr;
  BSX_HIP(c, launch_prep_fused(d_frames, c->width, c->height, c->roi, f32, u8, c->inW, c->inH, c->in_roi, c->tab_down.tab, c->bilateral, n, s, yuyv_in));
  return BSX_OK;
}
// logits = true: the network output tensor is written (stage tests, the stand-alone decode follows); false: the tail kernel of a
// segmented plan decodes straight into the temporal state of slots [slot, slot + n) and no logits exist
// per-launch path (DeepLab): the graph's final RESIZE_BILINEAR runs fused with the argmax decode + IIR — no full-resolution logits
bool argmax_tail(const bsx_ctx* c) {
  return !c->use_program && !c->keep_logits && c->model_type == BSX_MODEL_DEEPLAB && !c->plan.steps.empty() && c->plan.steps.back().out == c->plan.output &&
         resize_argmax_fusable(c->plan.steps.back());
}
// the middle of a segmented plan / the whole-network program: the specialised kernel when one was built, else the interpreter
hipError_t launch_program(bsx_ctx* c, int n, hipStream_t s, unsigned long long* timeline = nullptr) {
  long pf = (long)c->plan.arena_floats_per_stream;
  if (c->mid.fn) {
    float* arena = c->d_arena;
    const float* weights = c->d_weights;
    void* args[] = {&arena, &pf, &weights, &timeline};
    return hipModuleLaunchKernel(c->mid.fn, (unsigned)n, 1, 1, (unsigned)c->plan.mid_lanes, 1, 1, 0, s, args, nullptr);
  }
  return launch_frame_program(c->d_program, (int)c->plan.program.size(), c->plan.program_lds_floats, c->d_arena, pf, c->d_net_in, c->d_net_out, c->d_weights, n, s,
                              timeline);
}
// bsx_step_batch_pipelined: the launch that advances the temporal state (d_ofinal) waits until the composite of the previous batch — which reads that state on
// comp_stream — has finished.  Called right in front of every such launch; a no-op outside a pipelined call.
int state_write_fence(bsx_ctx* c, hipStream_t s) {
  if (!c->wait_before_state) return BSX_OK;
  hipEvent_t ev = c->wait_before_state;
  c->wait_before_state = nullptr;
  BSX_HIP(c, hipStreamWaitEvent(s, ev, 0));
  return BSX_OK;
}
bool infer_decodes(const bsx_ctx* c) { return (c->use_program && c->plan.seg.on && !c->keep_logits) || argmax_tail(c); }
// the four segment launches of a step: the graph-specialised hipRTC kernels when the context has them (the decode-fused tail only: the logits-writing variant of the stage
// tests stays with the ahead-of-time kernel), else the ahead-of-time kernels.  Same arguments, descriptor included (the specialised kernels ignore it).
hipError_t seg_launch(hipFunction_t fn, int tiles, int n, int lds_floats, hipStream_t s, void** args) {
  return hipModuleLaunchKernel(fn, (unsigned)tiles * (unsigned)n, 1, 1, kSegThreads, 1, 1, (unsigned)((size_t)lds_floats * sizeof(float)), s, args, nullptr);
}
hipError_t seg_head(bsx_ctx* c, int n, hipStream_t s) {
  const SegPlan& sp = c->plan.seg;
  long pf = (long)c->plan.arena_floats_per_stream;
  if (!c->seg_mod.fn)
    return launch_seg_head(sp.head, c->d_arena, pf, c->in_u8 ? (const void*)c->d_net_in_u8 : (const void*)c->d_net_in, c->d_weights, n, s, c->act16, c->in_u8, c->norm_scale, c->norm_offset);
  SegHead d = sp.head; float* arena = c->d_arena; const float* in = c->in_u8 ? reinterpret_cast<const float*>(c->d_net_in_u8) : c->d_net_in; const float* w = c->d_weights;
  float sc = c->norm_scale, of = c->norm_offset; int nf = n;
  void* args[] = {&d, &arena, &pf, &in, &w, &sc, &of, &nf};
  return seg_launch(c->seg_mod.fn, d.tiles_y * d.tiles_x, n, d.lds_floats, s, args);
}
hipError_t seg_k2(bsx_ctx* c, int n, hipStream_t s) {
  const SegPlan& sp = c->plan.seg;
  long pf = (long)c->plan.arena_floats_per_stream;
  if (!c->seg_fn[0]) return launch_seg_k2(sp.k2, c->d_arena, pf, c->d_weights, n, s, c->act16);
  SegK2 d = sp.k2; float* arena = c->d_arena; const float* w = c->d_weights; int nf = n;
  void* args[] = {&d, &arena, &pf, &w, &nf};
  return seg_launch(c->seg_fn[0], d.tiles_y * d.tiles_x, n, d.lds_floats, s, args);
}
hipError_t seg_k3(bsx_ctx* c, int n, hipStream_t s) {
  const SegPlan& sp = c->plan.seg;
  long pf = (long)c->plan.arena_floats_per_stream;
  if (!c->seg_fn[1]) return launch_seg_k3(sp.k3, c->d_arena, pf, c->d_weights, n, s, c->act16);
  SegK3 d = sp.k3; float* arena = c->d_arena; const float* w = c->d_weights; int nf = n;
  void* args[] = {&d, &arena, &pf, &w, &nf};
  return seg_launch(c->seg_fn[1], d.tiles_y * d.tiles_x, n, d.lds_floats, s, args);
}
hipError_t seg_tail(bsx_ctx* c, uint8_t* ofinal, bool logits, int n, hipStream_t s) {
  const SegPlan& sp = c->plan.seg;
  long pf = (long)c->plan.arena_floats_per_stream;
  if (logits || !c->seg_fn[2]) return launch_seg_tail(sp.tail, c->d_arena, pf, c->d_net_out, ofinal, c->d_weights, logits, n, s, c->act16);
  SegTail d = sp.tail; float* arena = c->d_arena; float* no = c->d_net_out; const float* w = c->d_weights; int nf = n;
  void* args[] = {&d, &arena, &pf, &no, &ofinal, &w, &nf};
  return seg_launch(c->seg_fn[2], d.tiles_y * d.tiles_x, n, d.lds_floats, s, args);
}

int run_infer(bsx_ctx* c, int n, hipStream_t s, bool logits = true, int slot = 0) {
  bsx_roctx::Range range("bsx:network");
  if (c->use_program && c->plan.seg.on) {
    const SegPlan& sp = c->plan.seg;
    const long pf = (long)c->plan.arena_floats_per_stream;
    BSX_HIP(c, seg_head(c, n, s));
    BSX_HIP(c, seg_k2(c, n, s));
    BSX_HIP(c, launch_program(c, n, s));
    BSX_HIP(c, seg_k3(c, n, s));
    if (sp.tail.pre_gate_off >= 0) BSX_HIP(c, launch_seg_gate(sp.tail.gate, c->d_arena, pf, c->d_weights, sp.tail.pre_gate_off, n, s));
    if (!logits) { const int frc = state_write_fence(c, s); if (frc) return frc; }       // the decoding tail reads and writes d_ofinal
    BSX_HIP(c, seg_tail(c, c->d_ofinal + (size_t)slot * c->outW * c->outH, logits, n, s));
    return BSX_OK;
  }
  if (c->use_program) {
    BSX_HIP(c, launch_program(c, n, s));
    return BSX_OK;
  }
  const bool fused_tail = !logits && argmax_tail(c);
  const size_t ns = c->plan.steps.size() - (fused_tail ? 1 : 0);
  for (size_t i = 0; i < ns; i++)
    BSX_HIP(c, launch_step(c->plan.steps[i], c->plan, c->d_arena, c->d_net_in, c->d_net_out, c->d_weights, n, c->n_streams, s, c->d_weights16, c->f16_terms, c->in_u8 ? c->d_net_in_u8 : nullptr, c->norm_scale, c->norm_offset));
  if (fused_tail) {
    const Step& last = c->plan.steps.back();
    { const int frc = state_write_fence(c, s); if (frc) return frc; }
    BSX_HIP(c, launch_resize_argmax_iir(last, c->d_arena + (size_t)c->plan.tensor_off[last.in0] * (size_t)c->n_streams,
                                        c->d_ofinal + (size_t)slot * c->outW * c->outH, n, s, c->tail_generic));
  }
  return BSX_OK;
}
// `slot` = first state slot (stream index) of the batch: frame i uses ofinal / mask slot `slot + i`
int run_decode(bsx_ctx* c, int n, hipStream_t s, int slot = 0) {
  bsx_roctx::Range range("bsx:decode");
  { const int frc = state_write_fence(c, s); if (frc) return frc; }
  BSX_HIP(c, launch_decode(c->model_type, c->tensor_ptr(c->plan.output), c->d_ofinal + (size_t)slot * c->outW * c->outH, c->outW * c->outH, c->outC, n, s));
  return BSX_OK;
}
// the mask up-scale table with its tile-class scratch re-based to stream `slot` (lanes run concurrently on disjoint slot ranges)
ResizeTab tab_up_at(const bsx_ctx* c, int slot) {
  ResizeTab t = c->tab_up.tab;
  if (t.tile_class) t.tile_class += (size_t)slot * c->tiles_per_frame;
  return t;
}
int run_mask(bsx_ctx* c, int n, hipStream_t s, int slot = 0) {
  bsx_roctx::Range range("bsx:mask");
  BSX_HIP(c, launch_mask_upscale_blur(c->d_ofinal + (size_t)slot * c->outW * c->outH, c->outW, c->outH, c->in_roi, tab_up_at(c, slot),
                                      c->d_masks + (size_t)slot * c->width * c->height, c->width, c->height, c->roi, n, s));
  return BSX_OK;
}
// bs_maskgen_process for n frames whose per-stream state lives in slots [slot, slot + n)
int process_impl(bsx_ctx* c, const uint8_t* d_frames, int n, int slot, hipStream_t s) {
  int rc;
  if ((rc = run_prep(c, d_frames, n, s))) return rc;
  if (c->onprep) { BSX_HIP(c, hipStreamSynchronize(s)); c->onprep(c->caller_ctx); }   // :303
  const bool fused_decode = infer_decodes(c);
  if ((rc = run_infer(c, n, s, !fused_decode, slot))) return rc;
  if (c->oninfer) { BSX_HIP(c, hipStreamSynchronize(s)); c->oninfer(c->caller_ctx); } // :311
  if (!fused_decode && (rc = run_decode(c, n, s, slot))) return rc;
  if (c->onmask) { BSX_HIP(c, hipStreamSynchronize(s)); c->onmask(c->caller_ctx); }   // :363
  return run_mask(c, n, s, slot);
}

}  // namespace

namespace {
// workgroup barriers one launch of the specialised middle kernel executes per frame: one in front of every micro-op, the ones inside squeeze-excite ops (pool | FC | FC)
// and between the channel chunks of staged depthwise ops — counted in the generated source itself (gen_mid.cpp), so the line cannot drift from the kernel
std::string mid_barrier_line(const Plan& p, bool act16) {
  std::string why;
  const std::string src = generate_mid_source(p, &why, act16);
  if (src.empty()) return "specialised middle kernel: none (" + why + ")\n";
  const size_t k0 = src.find("extern \"C\" __global__");
  auto count = [&](const char* pat) { size_t n = 0; for (size_t q = src.find(pat, k0); q != std::string::npos; q = src.find(pat, q + 1)) n++; return n; };
  const size_t ob = count("op_barrier();"), sy = count("__syncthreads();");
  char line[256];
  snprintf(line, sizeof line, "specialised middle kernel: %zu micro-ops, %zu workgroup barriers per launch (%zu between ops + %zu inside ops)\n", p.program.size(), ob + sy, ob, sy);
  return line;
}
}  // namespace

extern "C" {

const char* bsx_version(void) { return "bsx 0.1 (HIP gfx950, f32 NHWC)"; }

int bsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* bsx_last_error(const bsx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

bsx_ctx* bsx_new(const char* model_path, size_t threads, size_t width, size_t height, int n_streams, int device, bsx_debug_fn ondebug,
                 bsx_stage_fn onprep, bsx_stage_fn oninfer, bsx_stage_fn onmask, void* caller_ctx) {
  if (!model_path || !width || !height || n_streams <= 0 || n_streams > 65535) {   // grid.z of the mask kernel carries the stream index
    report(nullptr, ondebug, caller_ctx, "error: bad arguments to bsx_new\n"); return nullptr; }
  try {   // nothing may throw across the C ABI (std::bad_alloc / length_error from a hostile model file included)
  std::unique_ptr<bsx_ctx> c(new bsx_ctx);
  c->ondebug = ondebug; c->onprep = onprep; c->oninfer = oninfer; c->onmask = onmask; c->caller_ctx = caller_ctx;
  c->threads = threads; c->width = (int)width; c->height = (int)height; c->n_streams = n_streams; c->device = device;
  std::string err;
  if (!load_tflite(model_path, &c->graph, &err)) { report(nullptr, ondebug, caller_ctx, "error: %s\n", err.c_str()); return nullptr; }
  c->model_type = model_type_from_name(model_path);
  if (c->model_type == BSX_MODEL_UNKNOWN) { report(nullptr, ondebug, caller_ctx, "error: unknown model type '%s'.\n", model_path); return nullptr; }
  // normalisation (lib/libbackscrub.cc:132-148)
  if (c->model_type == BSX_MODEL_DEEPLAB) { c->norm_scale = (float)(1 / 127.5); c->norm_offset = -1.f; }
  else { c->norm_scale = (float)(1 / 255.0); c->norm_offset = 0.f; }
  const TensorInfo& ti = c->graph.tensors[c->graph.input];
  const TensorInfo& to = c->graph.tensors[c->graph.output];
  if (ti.shape.size() != 4 || to.shape.size() != 4 || ti.dims[0] != 1) {  // cf. getTensorMat, :85-112
    report(nullptr, ondebug, caller_ctx, "error: model input/output is not a single 4-D float tensor\n");
    return nullptr;
  }
  c->inH = ti.dims[1]; c->inW = ti.dims[2]; c->inC = ti.dims[3];
  c->outH = to.dims[1]; c->outW = to.dims[2]; c->outC = to.dims[3];
  if (c->inC != 3) { report(nullptr, ondebug, caller_ctx, "error: model input must have 3 channels\n"); return nullptr; }
  if ((c->model_type == BSX_MODEL_MEET && c->outC != 2) || (c->model_type == BSX_MODEL_DEEPLAB && c->outC < 16)) {
    report(nullptr, ondebug, caller_ctx, "error: model output has %d channels, unexpected for this model type\n", c->outC);
    return nullptr;
  }
  bool no_reuse = BSX_DBG_ENV("BSX_ARENA_NO_REUSE") != nullptr;
  // the per-launch path (BSX_NO_FRAME_PROGRAM) executes the plain step list: it needs the unsegmented plan
  const bool segments = BSX_DBG_ENV("BSX_NO_SEGMENTS") == nullptr && BSX_DBG_ENV("BSX_NO_FRAME_PROGRAM") == nullptr;
  if (!build_plan(c->graph, &c->plan, &err, !no_reuse, segments)) { report(nullptr, ondebug, caller_ctx, "error: unable to build GPU plan: %s\n", err.c_str()); return nullptr; }
  // ROI geometry, float arithmetic truncated to int exactly as lib/libbackscrub.cc:230-246
  float ratio = (float)c->inH / (float)c->inW;
  float frameratio = (float)height / (float)width;
  if (frameratio < ratio) {
    c->roi = Rect4{(int)((width - height / ratio) / 2), 0, (int)(height / ratio), (int)height};
    c->in_roi = Rect4{0, 0, c->inW, c->inH};
  } else {
    c->roi = Rect4{0, 0, (int)width, (int)height};
    c->in_roi = Rect4{(int)((c->inW - c->inH / frameratio) / 2), 0, (int)(c->inH / frameratio), c->inH};
  }
  if (c->roi.w <= 0 || c->roi.h <= 0 || c->roi.x < 0 || c->roi.x + c->roi.w > c->width || c->in_roi.w <= 0 || c->in_roi.x < 0 ||
      c->in_roi.x + c->in_roi.w > c->inW || c->in_roi.x + c->in_roi.w > c->outW || c->in_roi.h > c->outH) {
    report(nullptr, ondebug, caller_ctx, "error: frame/model geometry yields an empty or out-of-range ROI\n");
    return nullptr;
  }
  c->no_mask_blend_fusion = BSX_DBG_ENV("BSX_NO_MASK_BLEND_FUSION") != nullptr;
  c->no_bgblur_fusion = BSX_DBG_ENV("BSX_NO_BGBLUR_FUSION") != nullptr;
  c->no_mask_tile = BSX_DBG_ENV("BSX_NO_MASK_TILE") != nullptr;
  c->no_uniform_tiles = getenv("BSX_NO_UNIFORM_TILES") != nullptr;
  c->tail_generic = BSX_DBG_ENV("BSX_TAIL_GENERIC") != nullptr;
  c->keep_logits = BSX_DBG_ENV("BSX_KEEP_LOGITS") != nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    report(nullptr, ondebug, caller_ctx, "error: HIP device %d not available (%d visible)\n", device, ndev); return nullptr; }
  DeviceGuard guard(device);    // the caller's current device is restored on return
  if (init_device_state(c.get()) != BSX_OK) { bsx_ctx* raw = c.release(); bsx_delete(raw); return nullptr; }
  c->plan_text = c->plan.describe();
  {
    char line[256];
    snprintf(line, sizeof line, "frame program: %s, %zu micro-ops, LDS %d floats (%.1f KiB), %d tensors in LDS, %d in HBM\n",
             c->use_program ? "ON" : "off", c->plan.program.size(), c->plan.program_lds_floats, c->plan.program_lds_floats / 256.0,
             c->plan.program_lds_tensors, c->plan.program_global_tensors);
    c->plan_text += line;
    if (c->use_program) c->plan_text += "program execution: " + c->mid_note + "\n";
    if (c->use_program && c->plan.seg.on) c->plan_text += c->plan.seg_text;
    if (c->use_program && c->plan.seg.on) c->plan_text += "segment execution: " + c->seg_note + "\n";
    if (c->use_program && c->mid.fn) c->plan_text += mid_barrier_line(c->plan, c->act16);
    for (size_t i = 0; i < c->plan.program_labels.size(); i++) { c->plan_text += "P" + std::to_string(i) + " " + c->plan.program_labels[i] + "\n"; }
  }
  return c.release();
  } catch (const std::exception& e) {
    report(nullptr, ondebug, caller_ctx, "error: %s while creating the context\n", e.what());
    return nullptr;
  } catch (...) {
    report(nullptr, ondebug, caller_ctx, "error: unknown exception while creating the context\n");
    return nullptr;
  }
}

void bsx_delete(bsx_ctx* c) {
  if (!c) return;
  DeviceGuard guard(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto& kv : c->host_graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  rtc_unload(&c->mid);
  rtc_unload(&c->seg_mod);
  void* ptrs[] = {c->d_arena, c->d_net_in, c->d_net_in_u8, c->d_net_out, c->d_weights, c->d_ofinal, c->d_masks, c->d_host_frame, c->d_bgr_scratch, c->d_bgr_scratch2, c->d_bgblur_scratch, c->d_bgr_in_scratch, c->d_color_lut, c->tab_down.mem, c->tab_up.mem, c->d_program, c->d_weights16, c->d_tile_class};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& kv : c->bg_tabs) if (kv.second.mem) (void)hipFree(kv.second.mem);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  for (int k = 1; k < 4; k++) { if (c->lane_stream[k]) (void)hipStreamDestroy(c->lane_stream[k]); if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->comp_stream) { (void)hipStreamSynchronize(c->comp_stream); (void)hipStreamDestroy(c->comp_stream); }
  if (c->ev_pdone) (void)hipEventDestroy(c->ev_pdone);
  if (c->ev_pcomp) (void)hipEventDestroy(c->ev_pcomp);
  delete c;
}

int bsx_get_info(const bsx_ctx* c, bsx_info* o) {
  if (!c || !o) return BSX_EINVAL;
  memset(o, 0, sizeof *o);
  o->model_type = c->model_type; o->width = c->width; o->height = c->height; o->n_streams = c->n_streams;
  o->in_w = c->inW; o->in_h = c->inH; o->in_c = c->inC; o->out_w = c->outW; o->out_h = c->outH; o->out_c = c->outC;
  int r[4] = {c->roi.x, c->roi.y, c->roi.w, c->roi.h}, q[4] = {c->in_roi.x, c->in_roi.y, c->in_roi.w, c->in_roi.h};
  memcpy(o->roi, r, sizeof r); memcpy(o->in_roi, q, sizeof q);
  o->n_ops = c->graph.n_file_ops; o->n_steps = c->use_program ? (c->plan.seg.on ? 5 : 1) : (int)c->plan.steps.size(); o->device = c->device;
  o->norm_scale = c->norm_scale; o->norm_offset = c->norm_offset;
  o->nn_flops_per_frame = 2.0 * c->plan.macs_per_frame;
  o->act_bytes_per_stream = c->plan.arena_floats_per_stream * sizeof(float);
  return BSX_OK;
}

int bsx_reset(bsx_ctx* c, void* stream) {
  if (!c) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  hipStream_t s = pick(c, stream);
  const size_t N = (size_t)c->n_streams;
  c->pend.active = false;                                          // a composite still pending in the two-deep pipeline is dropped with the state it belongs to
  BSX_HIP(c, hipMemsetAsync(c->d_ofinal, 0, N * c->outW * c->outH, s));
  BSX_HIP(c, hipMemsetAsync(c->d_masks, 255, N * c->width * c->height, s));
  return BSX_OK;
}

uint8_t* bsx_masks_device(bsx_ctx* c) { return c ? c->d_masks : nullptr; }

int bsx_process_batch(bsx_ctx* c, const uint8_t* d_frames, int n, uint8_t* d_masks, void* stream) {
  if (!c || !d_frames || n <= 0 || n > c->n_streams) return BSX_EINVAL;
  if (c->pend.active) { c->last_error = "error: a pipelined composite is pending (flush with bsx_step_batch_pipelined(ctx, NULL, ...) first)\n"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  hipStream_t s = pick(c, stream);
  int rc;
  if ((rc = process_impl(c, d_frames, n, 0, s))) return rc;
  if (d_masks) BSX_HIP(c, hipMemcpyAsync(d_masks, c->d_masks, (size_t)n * c->width * c->height, hipMemcpyDeviceToDevice, s));
  return BSX_OK;
}

int bsx_process_host(bsx_ctx* c, int stream_idx, const uint8_t* h_bgr, size_t bgr_stride, uint8_t* h_mask, size_t mask_stride) {
  if (!c || !h_bgr || !h_mask || stream_idx < 0 || stream_idx >= c->n_streams) return BSX_EINVAL;
  if (bgr_stride < (size_t)c->width * 3 || mask_stride < (size_t)c->width) return BSX_ESIZE;
  if (c->pend.active) { c->last_error = "error: a pipelined composite is pending (flush with bsx_step_batch_pipelined(ctx, NULL, ...) first)\n"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  hipStream_t s = c->own_stream;
  const size_t fbytes = (size_t)c->width * c->height * 3;
  if (!c->d_host_frame) BSX_HIP(c, hipMalloc(&c->d_host_frame, fbytes));
  BSX_HIP(c, hipMemcpy2DAsync(c->d_host_frame, (size_t)c->width * 3, h_bgr, bgr_stride, (size_t)c->width * 3, c->height, hipMemcpyHostToDevice, s));
  // the single frame runs against the temporal state of slot `stream_idx` (the context itself is never modified)
  int rc = BSX_OK;
  hipGraphExec_t exec = nullptr;
  if (c->graph_state >= 0 && !c->onprep && !c->oninfer && !c->onmask) {
    auto it = c->host_graphs.find(stream_idx);
    if (it != c->host_graphs.end()) exec = it->second;
    else if (c->host_graphs.size() < 64) {                          // capture this slot's launch chain (the kernels' arguments carry the slot's state pointers)
      hipGraph_t g = nullptr;
      bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        rc = process_impl(c, c->d_host_frame, 1, stream_idx, s);
        ok = hipStreamEndCapture(s, &g) == hipSuccess && rc == BSX_OK && g != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
      if (ok) { c->host_graphs[stream_idx] = exec; c->graph_state = 1; }
      else { exec = nullptr; c->graph_state = -1; rc = BSX_OK; (void)hipGetLastError(); }      // nothing ran during the capture: fall through to the direct launches
    }
  }
  if (exec) BSX_HIP(c, hipGraphLaunch(exec, s));
  else rc = process_impl(c, c->d_host_frame, 1, stream_idx, s);
  if (rc) return rc;
  const uint8_t* my_mask = c->d_masks + (size_t)stream_idx * c->width * c->height;
  BSX_HIP(c, hipMemcpy2DAsync(h_mask, mask_stride, my_mask, (size_t)c->width, (size_t)c->width, c->height, hipMemcpyDeviceToHost, s));
  BSX_HIP(c, hipStreamSynchronize(s));
  return BSX_OK;
}

int bsx_composite_batch(bsx_ctx* c, const uint8_t* d_bg, size_t bg_frame_stride, const uint8_t* d_frames, const uint8_t* d_masks, uint8_t* d_out,
                        int n, void* stream) {
  if (!c || !d_bg || !d_frames || !d_out || n <= 0) return BSX_EINVAL;
  if (!d_masks) { if (n > c->n_streams) return BSX_EINVAL; d_masks = c->d_masks; }
  DeviceGuard guard(c->device);
  bsx_roctx::Range range("bsx:blend");
  BSX_HIP(c, launch_blend(d_bg, bg_frame_stride, d_frames, d_masks, d_out, (size_t)c->width * c->height, n, pick(c, stream)));
  return BSX_OK;
}

namespace {
// The context's batch buffers re-based to streams [f0, f0 + cap) for the launches enqueued inside the scope (one host thread enqueues a context's work).
// Frame-major buffers simply start at stream f0; the batch-major arena of the per-launch path becomes the group's own compact arena laid out for `cap` streams.
struct LaneView {
  bsx_ctx* c; float* net_in; uint32_t* net_in_u8; float* net_out; float* arena; int n_streams;
  LaneView(bsx_ctx* ctx, int f0, int cap) : c(ctx), net_in(ctx->d_net_in), net_in_u8(ctx->d_net_in_u8), net_out(ctx->d_net_out), arena(ctx->d_arena),
                                            n_streams(ctx->n_streams) {
    c->d_net_in += (size_t)f0 * c->inW * c->inH * c->inC;
    c->d_net_in_u8 += (size_t)f0 * c->inW * c->inH;
    c->d_net_out += (size_t)f0 * c->outW * c->outH * c->outC;
    c->d_arena += (size_t)f0 * c->plan.arena_floats_per_stream;
    c->n_streams = cap;
  }
  ~LaneView() { c->d_net_in = net_in; c->d_net_in_u8 = net_in_u8; c->d_net_out = net_out; c->d_arena = arena; c->n_streams = n_streams; }
};

// flags (bsx.h): BSX_STEP_YUYV — the composite leaves as YUYV 4:2:2 (2 B/px), convert_rgb_to_yuyv (deepseg.cc:87-106) applied in the blend's epilogue;
// BSX_STEP_FLIP_H / _V — cv::flip of the composite (deepseg.cc:667-673) folded into the epilogue's store addresses
int step_impl(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out, int n, void* stream, unsigned flags) {
  const int bgblur = (int)((flags >> 8) & 255u);                // BSX_STEP_BGBLUR(ksize): background = blur of the stream's own frame, d_bg unused
  if (!c || !d_frames || (!d_bg && !bgblur) || !d_out || n <= 0 || n > c->n_streams || (flags & ~(31u | 0xFF00u))) return BSX_EINVAL;
  if (bgblur && (bgblur > 31 || !(bgblur & 1) || d_frames == d_out)) return BSX_EINVAL;
  if (c->pend.active) { c->last_error = "error: a pipelined composite is pending (flush with bsx_step_batch_pipelined(ctx, NULL, ...) first)\n"; return BSX_EINVAL; }
  const int yuyv = (int)(flags & BSX_STEP_YUYV);
  const bool yin = (flags & BSX_STEP_YUYV_IN) != 0;             // the camera's raw 4:2:2 frames (cv::COLOR_YUV2BGR_YUYV, app/deepseg.cc:553,725) instead of BGR
  const unsigned flip = flags & (BSX_STEP_FLIP_H | BSX_STEP_FLIP_V);
  if ((yuyv || yin) && (c->width & 1)) return BSX_EINVAL;       // 4:2:2 pairs pixels horizontally
  DeviceGuard guard(c->device);
  const size_t px_frame = (size_t)c->width * c->height;
  if (yin) {
    // fused form: prep_fused_k and the mask tile kernel convert the pixels they read (2 B/px from HBM, no BGR frame in between).  Everything the fused kernels do
    // not cover — a background blurred from the frame itself, a geometry the tile kernel or the 8-byte tap window does not take, buffers that overlap, stage
    // callbacks — converts the batch into a context-owned BGR scratch first (bsx_yuyv_to_bgr's kernel) and runs the BGR step on it: same bytes, one more pass.
    const size_t in_b = (size_t)n * px_frame * 2, out_b = (size_t)n * px_frame * (yuyv ? 2 : 3);
    const bool ovl = d_out < d_frames + in_b && d_frames < d_out + out_b;
    const bool direct = !bgblur && !ovl && !c->onmask && !c->no_mask_blend_fusion && (!yuyv || ((uintptr_t)d_out & 3) == 0) && prep_yuyv_fusable(c->width, c->roi, c->tab_down.tab) &&
                        mask_blend_fusable(c->width, c->height, c->roi, d_bg, bg_frame_stride, d_frames, yuyv ? d_frames : d_out);
    if (!direct) {
      if (!c->d_bgr_in_scratch) BSX_HIP(c, hipMalloc(&c->d_bgr_in_scratch, (size_t)c->n_streams * px_frame * 3));
      BSX_HIP(c, launch_yuyv_to_bgr(d_frames, c->d_bgr_in_scratch, c->width, c->height, n, pick(c, stream)));
      return step_impl(c, c->d_bgr_in_scratch, d_bg, bg_frame_stride, d_out, n, stream, flags & ~BSX_STEP_YUYV_IN);
    }
  }
  if (bgblur) {
    if (!(flags & 15u) && !c->no_bgblur_fusion && gauss_blend_fusable(d_frames, c->d_masks, d_out, c->width, bgblur)) {
      // masks as usual (prep → network → decode → upscale + blur), then ONE pass over the frames: blur tile → blend with the frame and the mask → composite
      int rc = bsx_process_batch(c, d_frames, n, nullptr, stream);
      if (rc) return rc;
      BSX_HIP(c, launch_gauss_blend(d_frames, c->d_masks, d_out, c->width, c->height, bgblur, n, pick(c, stream)));
      return BSX_OK;
    }
    const size_t fb = (size_t)c->width * c->height * 3;
    if (!c->d_bgblur_scratch) BSX_HIP(c, hipMalloc(&c->d_bgblur_scratch, (size_t)c->n_streams * fb));
    BSX_HIP(c, launch_gauss_blur(d_frames, c->d_bgblur_scratch, c->width, c->height, bgblur, n, pick(c, stream)));
    return step_impl(c, d_frames, c->d_bgblur_scratch, fb, d_out, n, stream, flags & 15u);
  }
  // Aliasing (bsx.h): the reference flips `raw` in place (app/deepseg.cc:667-673), so a caller following it passes d_out == d_frames.  The fused tile kernel reads a
  // frame pixel at (x, y) and stores the flipped (or YUYV-packed: 2 B/px) result at ANOTHER address, which a different tile may not have read yet — with overlapping
  // buffers those forms take the unfused sequence (composite into the context's scratch first).  A plain composite in place (same address read, then written, by the
  // same lane) is fine; partially overlapping buffers are refused.
  const size_t in_bytes = (size_t)n * c->width * c->height * (yin ? 2 : 3), out_bytes = (size_t)n * c->width * c->height * (yuyv ? 2 : 3);
  const bool overlap = d_out < d_frames + in_bytes && d_frames < d_out + out_bytes;      // (never with yin: overlapping YUYV input took the scratch route above)
  if (overlap && !yuyv && !flip && d_out != d_frames) return BSX_EINVAL;
  const bool fuse = !c->onmask && !c->no_mask_blend_fusion && (!yuyv || ((uintptr_t)d_out & 3) == 0) && !(overlap && (yuyv || flip)) &&
                    mask_blend_fusable(c->width, c->height, c->roi, d_bg, bg_frame_stride, d_frames, yuyv ? d_frames : d_out);
  if (!fuse) {
    if (yin) return BSX_EINVAL;                                   // unreachable: `direct` above implies `fuse`
    int rc = bsx_process_batch(c, d_frames, n, nullptr, stream);
    if (rc) return rc;
    if (!yuyv && !flip) return bsx_composite_batch(c, d_bg, bg_frame_stride, d_frames, nullptr, d_out, n, stream);
    // unfused geometry: composite into a context-owned BGR scratch, then flip and / or pack as separate passes
    const size_t need = (size_t)c->n_streams * c->width * c->height * 3;
    if (!c->d_bgr_scratch) BSX_HIP(c, hipMalloc(&c->d_bgr_scratch, need));
    rc = bsx_composite_batch(c, d_bg, bg_frame_stride, d_frames, nullptr, c->d_bgr_scratch, n, stream);
    if (rc) return rc;
    const uint8_t* bgr = c->d_bgr_scratch;
    if (flip) {
      const int code = flip == (BSX_STEP_FLIP_H | BSX_STEP_FLIP_V) ? -1 : (flip == BSX_STEP_FLIP_H ? 1 : 0);
      uint8_t* dst = d_out;
      if (yuyv) { if (!c->d_bgr_scratch2) BSX_HIP(c, hipMalloc(&c->d_bgr_scratch2, need)); dst = c->d_bgr_scratch2; }
      if ((rc = bsx_flip_bgr(c, bgr, dst, c->width, c->height, n, code, stream))) return rc;
      bgr = dst;
    }
    return yuyv ? bsx_bgr_to_yuyv(c, bgr, d_out, c->width, c->height, n, stream) : BSX_OK;
  }
  // process (prep → network → decode), then mask-upscale+blur and alpha blend of each tile in ONE launch
  hipStream_t s = pick(c, stream);
  int rc;
  // (the per-launch path's arena is batch-major: a lane's compact arena is laid out for `per` streams starting at stream f0, so the LAST lane ends at
  //  K * per streams — lanes are only taken when that still fits the allocation, e.g. not for n = n_streams = 66, K = 4: 4 * 17 = 68)
  const int lane_per = (n + c->lanes - 1) / std::max(c->lanes, 1);
  if (c->lanes > 1 && n >= 16 * c->lanes && !c->onprep && !c->oninfer && !c->keep_logits && (c->use_program || c->lanes * lane_per <= c->n_streams)) {
    const int K = c->lanes, per = lane_per;
    const bool fd = infer_decodes(c);
    const size_t fb = (size_t)c->width * c->height * (yin ? 2 : 3), ob = (size_t)c->width * c->height * (yuyv ? 2 : 3), sm = (size_t)c->outW * c->outH;
    BSX_HIP(c, hipEventRecord(c->ev_fork, s));
    int lane_rc = BSX_OK;
    for (int k = 0; k < K && lane_rc == BSX_OK; k++) {
      const int f0 = k * per, nb = std::min(per, n - f0);
      if (nb <= 0) break;
      hipStream_t ls = k == 0 ? s : c->lane_stream[k];
      if (k > 0 && hipStreamWaitEvent(ls, c->ev_fork, 0) != hipSuccess) { lane_rc = BSX_EDEVICE; break; }
      {
        LaneView view(c, f0, per);
        lane_rc = run_prep(c, d_frames + (size_t)f0 * fb, nb, ls, false, yin);
        if (!lane_rc) lane_rc = run_infer(c, nb, ls, !fd, f0);
        if (!lane_rc && !fd) lane_rc = run_decode(c, nb, ls, f0);
      }
      if (!lane_rc && launch_mask_blend(c->d_ofinal + (size_t)f0 * sm, c->outW, c->outH, c->in_roi, tab_up_at(c, f0), c->d_masks + (size_t)f0 * c->width * c->height, c->width, c->height,
                                        c->roi, d_bg + (size_t)f0 * bg_frame_stride, bg_frame_stride, d_frames + (size_t)f0 * fb, d_out + (size_t)f0 * ob, nb, ls, (int)flags) != hipSuccess)
        lane_rc = BSX_EDEVICE;
      // forked lanes are ALWAYS joined, also after an error: the caller's stream must not be left with work in flight on streams it cannot see
      if (k > 0 && (hipEventRecord(c->ev_join[k], ls) != hipSuccess || hipStreamWaitEvent(s, c->ev_join[k], 0) != hipSuccess)) lane_rc = lane_rc ? lane_rc : BSX_EDEVICE;
    }
    if (lane_rc == BSX_EDEVICE && c->last_error.empty()) c->last_error = "error: HIP failure while enqueuing a lane of the step\n";
    return lane_rc;
  }
  if ((rc = run_prep(c, d_frames, n, s, false, yin))) return rc;
  if (c->onprep) { BSX_HIP(c, hipStreamSynchronize(s)); c->onprep(c->caller_ctx); }
  const bool fused_decode = infer_decodes(c);
  if ((rc = run_infer(c, n, s, !fused_decode, 0))) return rc;
  if (c->oninfer) { BSX_HIP(c, hipStreamSynchronize(s)); c->oninfer(c->caller_ctx); }
  if (!fused_decode && (rc = run_decode(c, n, s))) return rc;
  bsx_roctx::Range range("bsx:mask+blend");
  BSX_HIP(c, launch_mask_blend(c->d_ofinal, c->outW, c->outH, c->in_roi, c->tab_up.tab, c->d_masks, c->width, c->height, c->roi, d_bg,
                               bg_frame_stride, d_frames, d_out, n, s, (int)flags));
  return BSX_OK;
}
}  // namespace

int bsx_step_batch(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out, int n, void* stream) {
  return step_impl(c, d_frames, d_bg, bg_frame_stride, d_out, n, stream, 0u);
}
int bsx_step_batch_yuyv(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out_yuyv, int n, void* stream) {
  return step_impl(c, d_frames, d_bg, bg_frame_stride, d_out_yuyv, n, stream, BSX_STEP_YUYV);
}
int bsx_step_batch_ex(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out, int n, void* stream, unsigned flags) {
  return step_impl(c, d_frames, d_bg, bg_frame_stride, d_out, n, stream, flags);
}

// ---- two-deep pipeline: mask pipeline of batch k  ||  composite of batch k - 1 ------------------------------------------------------------------
// The reference overlaps exactly these two halves of its main loop: CalcMask::run() segments on a worker thread (app/deepseg.cc:182-216) while the capture loop
// blends and writes (:634-681).  Here both halves are GPU work of one context: the composite (HBM-bound: mask tiles + alpha blend) of the batch handed over
// by the previous call goes to comp_stream, the mask pipeline (latency-bound network kernels) of this call's batch to the caller's stream, and the only shared
// object — the model-resolution temporal state — is protected by ev_pcomp in front of the launch that advances it.  Unlike the reference's loop, which blends a
// frame with whatever mask is newest, every frame is composited with ITS OWN mask: results are bit-identical to bsx_step_batch_ex, one call later.
namespace {
int pipelined_objects(bsx_ctx* c) {
  if (c->comp_stream) return BSX_OK;
  // the composite fills the gaps of the network kernels, not the other way round: lowest priority the device offers (BSX_PIPE_PRIO=0: default priority)
  int lo = 0, hi = 0;
  const char* pe = BSX_DBG_ENV("BSX_PIPE_PRIO");
  if (!(pe && atoi(pe) == 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
    BSX_HIP(c, hipStreamCreateWithPriority(&c->comp_stream, hipStreamNonBlocking, lo));
  } else {
    BSX_HIP(c, hipStreamCreateWithFlags(&c->comp_stream, hipStreamNonBlocking));
  }
  BSX_HIP(c, hipEventCreateWithFlags(&c->ev_pdone, hipEventDisableTiming));
  BSX_HIP(c, hipEventCreateWithFlags(&c->ev_pcomp, hipEventDisableTiming));
  return BSX_OK;
}
int composite_pending(bsx_ctx* c, hipStream_t s, bool concurrent) {
  const bsx_ctx::PendingComposite& p = c->pend;
  // next to the network kernels the composite holds at most pipe_wgs workgroups per CU (an unused LDS pad: static 16.7 KB + pad <= 64 KB, i.e. >= 2 per CU)
  int pad = 0;
  if (concurrent && c->pipe_wgs > 0) pad = std::max(0, std::min(47 * 1024, (160 * 1024 / c->pipe_wgs - 17 * 1024) & ~255));
  bsx_roctx::Range range("bsx:mask+blend");
  BSX_HIP(c, launch_mask_blend(c->d_ofinal, c->outW, c->outH, c->in_roi, c->tab_up.tab, c->d_masks, c->width, c->height, c->roi, p.bg, p.bg_stride, p.frames, p.out, p.n, s,
                               (int)p.flags, pad));
  return BSX_OK;
}
}  // namespace

int bsx_step_batch_pipelined(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out, int n, void* stream, unsigned flags) {
  if (!c) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  hipStream_t s = pick(c, stream);
  if (!d_frames) {                                                  // flush: the composite of the last batch, on the caller's stream
    if (!c->pend.active) return BSX_OK;
    BSX_HIP(c, hipStreamWaitEvent(s, c->ev_pdone, 0));               // the pending batch's network may have run on another stream than this call's
    const int rc = composite_pending(c, s, false);
    c->pend.active = false;
    return rc;
  }
  if (!d_bg || !d_out || n <= 0 || n > c->n_streams || (flags & ~31u)) return BSX_EINVAL;
  const int yuyv = (int)(flags & BSX_STEP_YUYV);
  const bool yin = (flags & BSX_STEP_YUYV_IN) != 0;
  if ((yuyv || yin) && (c->width & 1)) return BSX_EINVAL;
  // the pipeline exists for the fused tile kernel only; out(k) is written while frames(k + 1) are read, so the buffers of a call must not overlap at all
  const size_t in_bytes = (size_t)n * c->width * c->height * (yin ? 2 : 3), out_bytes = (size_t)n * c->width * c->height * (yuyv ? 2 : 3);
  const bool overlap = d_out < d_frames + in_bytes && d_frames < d_out + out_bytes;
  const bool fuse = !c->onprep && !c->oninfer && !c->onmask && !c->no_mask_blend_fusion && !overlap && (!yuyv || ((uintptr_t)d_out & 3) == 0) &&
                    (!yin || prep_yuyv_fusable(c->width, c->roi, c->tab_down.tab)) &&
                    mask_blend_fusable(c->width, c->height, c->roi, d_bg, bg_frame_stride, d_frames, yuyv ? d_frames : d_out);
  if (!fuse) { c->last_error = "error: bsx_step_batch_pipelined needs the fused mask + blend geometry, no stage callbacks and non-overlapping buffers\n"; return BSX_EINVAL; }
  int rc = pipelined_objects(c);
  if (rc) return rc;
  bool forked = false;
  if (c->pend.active) {
    BSX_HIP(c, hipStreamWaitEvent(c->comp_stream, c->ev_pdone, 0)); // behind the network of the pending batch, whatever stream the call that enqueued it was given
    BSX_HIP(c, hipStreamWaitEvent(s, c->ev_pdone, 0));              // this call's network reuses the arena of the previous one: same order if the caller changed streams
    rc = composite_pending(c, c->comp_stream, true);
    // forked work is ALWAYS joined, also after an error: the caller's stream must not be left with work in flight on a stream it cannot see
    if (hipEventRecord(c->ev_pcomp, c->comp_stream) != hipSuccess) rc = rc ? rc : BSX_EDEVICE;
    c->pend.active = false;
    forked = true;
    if (!rc) c->wait_before_state = c->ev_pcomp;
  }
  if (!rc) rc = run_prep(c, d_frames, n, s, false, yin);
  const bool fused_decode = infer_decodes(c);
  if (!rc) rc = run_infer(c, n, s, !fused_decode, 0);
  if (!rc && !fused_decode) rc = run_decode(c, n, s);
  if (forked && (c->wait_before_state || rc)) {                     // fence not consumed (error on the way): join here
    c->wait_before_state = nullptr;
    if (hipStreamWaitEvent(s, c->ev_pcomp, 0) != hipSuccess) rc = rc ? rc : BSX_EDEVICE;
  }
  if (rc) return rc;
  BSX_HIP(c, hipEventRecord(c->ev_pdone, s));
  c->pend.active = true; c->pend.frames = d_frames; c->pend.bg = d_bg; c->pend.bg_stride = bg_frame_stride; c->pend.out = d_out; c->pend.n = n; c->pend.flags = flags;
  return BSX_OK;
}

int bsx_resize_bgr(bsx_ctx* c, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh, int n, void* stream) {
  if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || n <= 0) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  auto key = std::make_pair(std::make_pair(sw, sh), std::make_pair(dw, dh));
  auto it = c->bg_tabs.find(key);
  if (it == c->bg_tabs.end()) {
    DevResizeTab d;
    int rc = upload_tab(c, make_resize_tab(sw, sh, dw, dh), &d);
    if (rc) return rc;
    it = c->bg_tabs.emplace(key, d).first;
  }
  BSX_HIP(c, launch_resize_bgr(d_src, d_dst, it->second.tab, n, pick(c, stream)));
  return BSX_OK;
}

int bsx_flip_bgr(bsx_ctx* c, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int code, void* stream) {
  if (!c || !d_src || !d_dst || d_src == d_dst || w <= 0 || h <= 0 || n <= 0) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  BSX_HIP(c, launch_flip_bgr(d_src, d_dst, w, h, code, n, pick(c, stream)));
  return BSX_OK;
}

int bsx_gaussian_blur_bgr(bsx_ctx* c, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int ksize, void* stream) {
  if (!c || !d_src || !d_dst || d_src == d_dst || w <= 0 || h <= 0 || n <= 0 || ksize < 1 || ksize > 31 || !(ksize & 1)) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  BSX_HIP(c, launch_gauss_blur(d_src, d_dst, w, h, ksize, n, pick(c, stream)));
  return BSX_OK;
}

int bsx_bgr_to_yuyv(bsx_ctx* c, const uint8_t* d_bgr, uint8_t* d_yuyv, int w, int h, int n, void* stream) {
  if (!c || !d_bgr || !d_yuyv || w <= 0 || h <= 0 || n <= 0) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  BSX_HIP(c, launch_bgr_to_yuyv(d_bgr, d_yuyv, w, h, n, pick(c, stream)));
  return BSX_OK;
}

int bsx_yuyv_to_bgr(bsx_ctx* c, const uint8_t* d_yuyv, uint8_t* d_bgr, int w, int h, int n, void* stream) {
  if (!c || !d_yuyv || !d_bgr || w <= 0 || h <= 0 || n <= 0 || (((long)w * h) & 1)) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  BSX_HIP(c, launch_yuyv_to_bgr(d_yuyv, d_bgr, w, h, n, pick(c, stream)));
  return BSX_OK;
}

int bsx_debug_buffer(bsx_ctx* c, int which, void** d_ptr, size_t* bytes) {
  if (!c || !d_ptr || !bytes) return BSX_EINVAL;
  const size_t N = (size_t)c->n_streams;
  switch (which) {
    case 0: *d_ptr = c->tensor_ptr(c->plan.input); *bytes = N * c->inW * c->inH * c->inC * 4; break;
    case 1: *d_ptr = c->tensor_ptr(c->plan.output); *bytes = N * c->outW * c->outH * c->outC * 4; break;
    case 2: *d_ptr = c->d_ofinal; *bytes = N * c->outW * c->outH; break;
    case 3: *d_ptr = c->d_masks; *bytes = N * c->width * c->height; break;
    default: return BSX_EINVAL;
  }
  return BSX_OK;
}

int bsx_debug_run_stage(bsx_ctx* c, int stage, const uint8_t* d_frames, int n, void* stream) {
  if (!c || n <= 0 || n > c->n_streams) return BSX_EINVAL;
  if (stage >= 1 && stage <= 3 && c->pend.active) { c->last_error = "error: a pipelined composite is pending (flush with bsx_step_batch_pipelined(ctx, NULL, ...) first)\n"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  hipStream_t s = pick(c, stream);
  switch (stage) {
    case 0: if (!d_frames) return BSX_EINVAL; return run_prep(c, d_frames, n, s, true);
    case 1: return run_infer(c, n, s);
    case 2: return run_decode(c, n, s);
    case 3: return run_mask(c, n, s);
    case 4:                                                       // stage 0 on raw YUYV 4:2:2 frames (BSX_STEP_YUYV_IN's prep): the f32 network input, for the stage tests
      if (!d_frames || !prep_yuyv_fusable(c->width, c->roi, c->tab_down.tab)) return BSX_EINVAL;
      return run_prep(c, d_frames, n, s, true, true);
    default: return BSX_EINVAL;
  }
}

int bsx_profile_batch(bsx_ctx* c, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_stride, uint8_t* d_out, int n, int iters,
                      bsx_launch_stat* out, int cap, void* stream) {
  if (!c || !d_frames || !d_bg || !d_out || !out || n <= 0 || n > c->n_streams || iters <= 0) return BSX_EINVAL;
  if (c->pend.active) { c->last_error = "error: a pipelined composite is pending (flush with bsx_step_batch_pipelined(ctx, NULL, ...) first)\n"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  hipStream_t s = pick(c, stream);
  const bool seg = c->use_program && c->plan.seg.on;
  const bool fused_decode = infer_decodes(c);
  const bool atail = argmax_tail(c);
  const int n_net = c->use_program ? (seg ? (c->plan.seg.tail.pre_gate_off >= 0 ? 6 : 5) : 1) : (int)c->plan.steps.size();
  const bool fuse_tail = !c->onmask && !c->no_mask_blend_fusion &&
                         mask_blend_fusable(c->width, c->height, c->roi, d_bg, bg_stride, d_frames, d_out);
  const int L = 1 + n_net + (fused_decode ? 0 : 1) + (fuse_tail ? 1 : 2) + (fuse_tail ? 1 : 0);   // + a stand-alone blend launch when the step's tail is fused
  if (cap < L) return BSX_EINVAL;
  // ONE event between consecutive launches (round 6: L + 1 events, not 2 L): launch k is timed from the event behind launch k - 1 to the event behind itself, so
  // the per-launch figures add up to the pass and carry one event's cost each instead of two (round 5's bracketing pairs read 11 % over the un-instrumented step)
  std::vector<hipEvent_t> ev((size_t)L + 1);
  for (auto& e : ev) BSX_HIP(c, hipEventCreate(&e));
  std::vector<double> sum(L, 0.0);
  const double N = n, px = (double)c->width * c->height;
  for (int it = 0; it < iters; it++) {
    int k = 0;
    BSX_HIP(c, hipEventRecord(ev[0], s));
#define BSX_TIMED(call)                                    \
    do {                                                   \
      BSX_HIP(c, (call));                                  \
      BSX_HIP(c, hipEventRecord(ev[k + 1], s));            \
      k++;                                                 \
    } while (0)
    BSX_TIMED(launch_prep_fused(d_frames, c->width, c->height, c->roi, c->in_u8 ? nullptr : c->tensor_ptr(c->plan.input), c->in_u8 ? c->d_net_in_u8 : nullptr, c->inW, c->inH, c->in_roi,
                                c->tab_down.tab, c->bilateral, n, s));
    const long pf = (long)c->plan.arena_floats_per_stream;
    if (seg) {
      const SegPlan& sp = c->plan.seg;
      BSX_TIMED(seg_head(c, n, s));
      BSX_TIMED(seg_k2(c, n, s));
      BSX_TIMED(launch_program(c, n, s));
      BSX_TIMED(seg_k3(c, n, s));
      if (sp.tail.pre_gate_off >= 0) BSX_TIMED(launch_seg_gate(sp.tail.gate, c->d_arena, pf, c->d_weights, sp.tail.pre_gate_off, n, s));
      BSX_TIMED(seg_tail(c, c->d_ofinal, !fused_decode, n, s));
    } else if (c->use_program)
      BSX_TIMED(launch_program(c, n, s));
    else {
      for (size_t si = 0; si + (atail ? 1 : 0) < c->plan.steps.size(); si++)
        BSX_TIMED(launch_step(c->plan.steps[si], c->plan, c->d_arena, c->d_net_in, c->d_net_out, c->d_weights, n, c->n_streams, s, c->d_weights16, c->f16_terms, c->in_u8 ? c->d_net_in_u8 : nullptr, c->norm_scale, c->norm_offset));
      if (atail) BSX_TIMED(launch_resize_argmax_iir(c->plan.steps.back(), c->d_arena + (size_t)c->plan.tensor_off[c->plan.steps.back().in0] * (size_t)c->n_streams,
                                                    c->d_ofinal, n, s, c->tail_generic));
    }
    if (!fused_decode) BSX_TIMED(launch_decode(c->model_type, c->tensor_ptr(c->plan.output), c->d_ofinal, c->outW * c->outH, c->outC, n, s));
    if (fuse_tail) {
      BSX_TIMED(launch_mask_blend(c->d_ofinal, c->outW, c->outH, c->in_roi, c->tab_up.tab, c->d_masks, c->width, c->height, c->roi, d_bg, bg_stride,
                                  d_frames, d_out, n, s));
      // not part of the step: the plain alpha-blend kernel (bsx_composite_batch) on the same buffers, for its own roofline line
      BSX_TIMED(launch_blend(d_bg, bg_stride, d_frames, c->d_masks, d_out, (size_t)c->width * c->height, n, s));
    } else {
      BSX_TIMED(launch_mask_upscale_blur(c->d_ofinal, c->outW, c->outH, c->in_roi, c->tab_up.tab, c->d_masks, c->width, c->height, c->roi, n, s));
      BSX_TIMED(launch_blend(d_bg, bg_stride, d_frames, c->d_masks, d_out, (size_t)c->width * c->height, n, s));
    }
#undef BSX_TIMED
    BSX_HIP(c, hipStreamSynchronize(s));
    for (int j = 0; j < L; j++) { float ms = 0; BSX_HIP(c, hipEventElapsedTime(&ms, ev[j], ev[j + 1])); sum[j] += ms; }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  auto put = [&](int j, const std::string& name, double bytes, double flops) {
    memset(&out[j], 0, sizeof out[j]);
    snprintf(out[j].name, sizeof out[j].name, "%s", name.c_str());
    out[j].avg_ms = sum[j] / iters; out[j].bytes = bytes; out[j].flops = flops;
  };
  int j = 0;
  const double canvas = (double)c->inW * c->inH;
  // prep_resize: reads the TOUCHED source pixels of the ROI — a bilinear tap pair per destination column / row, i.e. at most 2 x 2 source
  // pixels per canvas pixel of in_roi (a 5x down-scale touches 16 % of the ROI)
  const double touched = (double)std::min(c->roi.w, 2 * c->in_roi.w) * (double)std::min(c->roi.h, 2 * c->in_roi.h);
  put(j++, "prep", N * (3.0 * touched + (c->in_u8 ? 4.0 : 12.0) * canvas), 0);       // touched source pixels in, the network input out
  if (seg) {
    // algorithmic bytes of each segment = the tensors it must read once + write once (f32); flops from the fused steps it covers
    const std::vector<Step>& S = c->plan.steps;
    const int NS = (int)S.size();
    auto macs = [&](int a, int b) { double m = 0; for (int i = a; i <= b; i++) m += S[i].macs; return m; };
    const SegPlan& sp = c->plan.seg;
    const double eA = 16.0 * sp.head.H1 * sp.head.W1, eb0 = 16.0 * sp.head.H2 * sp.head.W2, ec0 = (double)sp.k2.dw.C * sp.k2.H3 * sp.k2.W3;
    const double elo2 = 16.0 * sp.k3.HL * sp.k3.WL, elo = 16.0 * sp.k3.H2 * sp.k3.W2;
    const double ein = (double)c->inW * c->inH * c->inC, eout = (double)c->outW * c->outH * c->outC;
    put(j++, "seg_head", N * ((c->in_u8 ? 4.0 * c->inW * c->inH : 4.0 * ein) + 4.0 * (eA + eb0)), N * 2.0 * macs(0, 2));
    put(j++, "seg_k2", N * 4.0 * (eb0 + eb0 + ec0), N * 2.0 * macs(3, 8));
    put(j++, "frame_program", N * 4.0 * (ec0 + elo2) + 4.0 * c->plan.weights.size(), N * 2.0 * macs(9, NS - 11));
    put(j++, "seg_k3", N * 4.0 * (eb0 + elo2 + elo), N * 2.0 * macs(NS - 10, NS - 8));
    if (sp.tail.pre_gate_off >= 0) put(j++, "seg_gate", N * 4.0 * (16.0 * (sp.tail.gate.part[0].n + sp.tail.gate.part[1].n) + 16.0), 0);      // the tail's gate, once per frame
    put(j++, fused_decode ? "seg_tail+decode" : "seg_tail", N * (4.0 * (eA + elo) + (fused_decode ? 2.0 * c->outW * c->outH : 4.0 * eout)), N * 2.0 * macs(NS - 7, NS - 1));
  } else if (c->use_program) {
    // algorithmic bytes of the fused network = its input tensor + its output tensor + the weights once
    double io = (double)c->inW * c->inH * c->inC + (double)c->outW * c->outH * c->outC;
    put(j++, "frame_program", N * io * 4.0 + 4.0 * c->plan.weights.size(), N * 2.0 * c->plan.macs_per_frame);
  } else
  for (const Step& st : c->plan.steps) {
    double in = (double)st.H * st.W * st.Cin, o = (double)st.OH * st.OW * st.Cout, b = 0;
    switch (st.kind) {
      case StepKind::Eltwise: b = in + o + (st.elt == kEltUnary ? 0 : (st.bcast1 ? st.Cin : in)) + (st.elt == kEltMulAdd ? in : 0); break;
      case StepKind::Concat: b = 2 * o; break;
      case StepKind::Gap: b = in + st.Cin; break;
      default: b = in + o + (st.residual >= 0 ? o : 0) + (st.in_scale >= 0 ? st.Cin : 0); break;
    }
    const bool is_tail = atail && &st == &c->plan.steps.back();
    const bool ir_on = c->d_weights16 && c->f16_terms > 0;
    const bool h0 = c->plan.steps[0].fuse_head0;
    const bool h0_member = h0 && (&st == &c->plan.steps[1] || &st == &c->plan.steps[2]);
    if (h0_member || (st.fused_away && !(h0 && &st <= &c->plan.steps[2]) && ir_on)) { put(j++, st.label + " (inside the launch before)", 0, 0); continue; }   // no-op slot of a fused group
    if (st.chain_mid >= 0 && chain3_on(c->plan, st.chain_mid, n, c->d_weights16, c->f16_terms)) { put(j++, st.label + " (inside the chained launch)", 0, 0); continue; }
    if (st.chain_first >= 0 && chain3_on(c->plan, (int)(&st - c->plan.steps.data()), n, c->d_weights16, c->f16_terms)) {      // three 1x1 convolutions: reads the first one's input, writes the last one's output
      const Step& ca = c->plan.steps[st.chain_first];
      const Step& cc = c->plan.steps[st.chain_last];
      put(j++, ca.label + "+" + st.label + "+" + cc.label, N * 4.0 * ((double)ca.H * ca.W * ca.Cin + (double)cc.OH * cc.OW * cc.Cout), N * 2.0 * (ca.macs + st.macs + cc.macs));
      continue;
    }
    if (st.fuse_head0 && h0) {                                   // stem + depthwise + 1x1: reads the network input, writes the 1x1's output
      const Step& d1 = c->plan.steps[1];
      const Step& p2 = c->plan.steps[2];
      put(j++, st.label + "+" + d1.label + "+" + p2.label, N * ((c->in_u8 ? 4.0 * st.H * st.W : 4.0 * in) + 4.0 * (double)p2.OH * p2.OW * p2.Cout), N * 2.0 * (st.macs + d1.macs + p2.macs));
      continue;
    }
    if (st.fuse_dw >= 0 && ir_on) {                              // expand + depthwise in one launch: reads the expand's input, writes the depthwise's output
      const Step& dd = c->plan.steps[st.fuse_dw];
      put(j++, st.label + "+" + dd.label, N * 4.0 * (in + (double)dd.OH * dd.OW * dd.Cout), N * 2.0 * (st.macs + dd.macs));
      continue;
    }
    put(j++, is_tail ? st.label + "+argmax" : st.label, is_tail ? N * (4.0 * in + 2.0 * st.OH * st.OW) : N * b * 4.0, N * 2.0 * st.macs);
  }
  if (!fused_decode) put(j++, "decode_iir", N * ((double)c->outW * c->outH * c->outC * 4.0 + 2.0 * c->outW * c->outH), 0);
  if (fuse_tail) {
    // fused: model-res mask in, full-res mask out (1 B/px), bg + frame in (6 B/px), composite out (3 B/px)
    put(j++, "mask_blend", N * ((double)c->in_roi.w * c->in_roi.h + 10.0 * px), 0);
    put(j++, "blend(standalone)", N * 10.0 * px, 0);
  } else {
    put(j++, "mask_upscale_blur", N * ((double)c->in_roi.w * c->in_roi.h + (double)c->roi.w * c->roi.h), 0);
    put(j++, "blend", N * 10.0 * px, 0);
  }
  return j;
}

int bsx_debug_mask_tile_stats(bsx_ctx* c, int n, long* out4) {
  if (!c || !out4 || n <= 0 || n > c->n_streams) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  out4[0] = (long)c->tiles_per_frame * n; out4[1] = out4[2] = 0; out4[3] = out4[0];
  if (!c->d_tile_class) return BSX_OK;                          // generic kernel / shortcut off: every tile is general
  // the classifier the launches themselves run, on the current temporal state
  BSX_HIP(c, launch_tile_class(c->d_ofinal, c->outW, c->outH, c->in_roi, c->tab_up.tab, c->roi, n, nullptr));
  std::vector<uint8_t> cls((size_t)out4[0]);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(cls.data(), c->d_tile_class, cls.size(), hipMemcpyDeviceToHost) != hipSuccess) return BSX_EDEVICE;
  out4[3] = 0;
  for (uint8_t v : cls) out4[v == 1 ? 1 : (v == 2 ? 2 : 3)]++;
  return BSX_OK;
}

int bsx_debug_gauss_coeffs(int ksize, int shift, uint32_t* c4, uint32_t* c2) {
  return (c4 && c2 && gauss_coeff_words(ksize, shift, c4, c2)) ? BSX_OK : BSX_EINVAL;
}

int bsx_debug_program_timeline(bsx_ctx* c, int n, unsigned long long* ticks, int cap, void* stream) {
  if (!c || !ticks || n <= 0 || n > c->n_streams) return BSX_EINVAL;
  DeviceGuard guard(c->device);
  if (!c->use_program) return 0;
  const int L = (int)c->plan.program.size();
  if (cap < L + 1 || L + 1 > 256) return BSX_EINVAL;
  unsigned long long* d = nullptr;
  const size_t kTl = 1024 + 64 * 16 * 4;       // coarse stamps [0,256), sub-phase accumulators [256,512), fine per-wave stamps [1024, ..)
  BSX_HIP(c, hipMalloc(&d, kTl * sizeof(unsigned long long)));
  BSX_HIP(c, hipMemset(d, 0, kTl * sizeof(unsigned long long)));
  hipStream_t s = pick(c, stream);
  BSX_HIP(c, launch_program(c, n, s, d));
  BSX_HIP(c, hipStreamSynchronize(s));
  BSX_HIP(c, hipMemcpy(ticks, d, std::min((size_t)cap, kTl) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return L;
}

int bsx_model_describe(const char* model_path, char* buf, size_t cap) {
  if (!model_path || !buf || !cap) return BSX_EINVAL;
  std::string err, out;
  int rc = BSX_OK;
  try {
  Graph g; Plan p;
  if (!load_tflite(model_path, &g, &err) || !build_plan(g, &p, &err)) { out = err; rc = BSX_EMODEL; }
  else {
    char head[256];
    snprintf(head, sizeof head, "ops=%d nodes=%d steps=%d macs=%.0f arena_floats=%zu\n", g.n_file_ops, (int)g.nodes.size(), (int)p.steps.size(),
             p.macs_per_frame, p.arena_floats_per_stream);
    out = head + p.describe();
    snprintf(head, sizeof head, "program micro-ops=%zu lds_floats=%d lds_tensors=%d hbm_tensors=%d arena_bytes_per_frame=%ld placement_policy=%u lds_blocks=%zu lds_check=%s\n", p.program.size(),
             p.program_lds_floats, p.program_lds_tensors, p.program_global_tensors, p.program_arena_bytes, p.program_policy, p.program_blocks.size(), p.program_check.c_str());
    out += head;
    for (size_t i = 0; i < p.program_labels.size(); i++) out += "P" + std::to_string(i) + " " + p.program_labels[i] + "\n";
    if (p.seg.on) out += p.seg_text;
    if (!p.program.empty()) out += mid_barrier_line(p, false);
    if (BSX_DBG_ENV("BSX_PLAN_BLOCKS"))           // debugging: the LDS reservations of the program (float offset, length, first / last step, owner)
      for (const auto& b : p.program_blocks) { snprintf(head, sizeof head, "block off=%d len=%d steps=[%d,%d] %s\n", b.off, b.len, b.from, b.until, b.what.c_str()); out += head; }
  }
  } catch (const std::exception& e) { out = std::string("exception while reading the model: ") + e.what(); rc = BSX_EMODEL; }
  catch (...) { out = "unknown exception while reading the model"; rc = BSX_EMODEL; }
  snprintf(buf, cap, "%s", out.c_str());
  return rc;
}

long bsx_model_kernel_source(const char* model_path, char* buf, size_t cap) {
  if (!model_path || !buf || !cap) return BSX_EINVAL;
  try {
    Graph g; Plan p;
    std::string err, why;
    if (!load_tflite(model_path, &g, &err) || !build_plan(g, &p, &err)) { snprintf(buf, cap, "%s", err.c_str()); return BSX_EMODEL; }
    const char* a16 = getenv("BSX_ACT16");
    if (p.program.empty()) { snprintf(buf, cap, "no program"); return 0; }
    MidBuild mb;                                                    // the source of the kernel a context would RUN (the form build_mid_kernel picks: compiles, no GPU needed)
    const std::string e = build_mid_kernel(p, a16 && atoi(a16) != 0 && p.seg.on, "gfx950", &mb);
    if (!e.empty()) {
      const std::string src = generate_mid_source(p, &why, a16 && atoi(a16) != 0 && p.seg.on);      // a compiler failure still shows the source; a graph without a body shows why
      if (src.empty()) { snprintf(buf, cap, "%s", why.c_str()); return 0; }
      snprintf(buf, cap, "%s", src.c_str());
      return (long)src.size();
    }
    snprintf(buf, cap, "%s", mb.source.c_str());
    return (long)mb.source.size();
  } catch (...) { snprintf(buf, cap, "exception while reading the model"); return BSX_EMODEL; }
}

long bsx_model_seg_source(const char* model_path, char* buf, size_t cap) {
  if (!model_path || !buf || !cap) return BSX_EINVAL;
  try {
    Graph g; Plan p;
    std::string err, why;
    if (!load_tflite(model_path, &g, &err) || !build_plan(g, &p, &err)) { snprintf(buf, cap, "%s", err.c_str()); return BSX_EMODEL; }
    const char* a16 = getenv("BSX_ACT16");
    const std::string src = generate_seg_source(p, a16 && atoi(a16) != 0 && p.seg.on, BSX_DBG_ENV("BSX_F32_INPUT") == nullptr, &why);
    if (src.empty()) { snprintf(buf, cap, "%s", why.c_str()); return 0; }
    snprintf(buf, cap, "%s", src.c_str());
    return (long)src.size();
  } catch (...) { snprintf(buf, cap, "exception while reading the model"); return BSX_EMODEL; }
}

int bsx_model_precompile(const char* model_path, const char* arch, char* msg, size_t cap) {
  if (!model_path || !msg || !cap) return BSX_EINVAL;
  try {
    Graph g; Plan p;
    std::string err, why, log;
    if (!load_tflite(model_path, &g, &err) || !build_plan(g, &p, &err)) { snprintf(msg, cap, "%s", err.c_str()); return BSX_EMODEL; }
    const char* a16 = getenv("BSX_ACT16");                         // the variant a context created under the same environment would ask for
    if (p.program.empty()) { snprintf(msg, cap, "interpreted (no program)"); return BSX_OK; }
    MidBuild mb;
    const std::string e = build_mid_kernel(p, a16 && atoi(a16) != 0 && p.seg.on, arch ? arch : "gfx950", &mb);
    if (!e.empty()) { snprintf(msg, cap, "%s", e.c_str()); return e.compare(0, 19, "interpreted (hipRTC") == 0 ? BSX_EMODEL : BSX_OK; }
    // the segment kernels of the same graph (gen_seg.cpp), in the variant a context created under this environment would load
    std::string seg_msg;
    if (p.seg.on) {
      std::vector<char> scode;
      bool scached = false;
      size_t sbytes = 0;
      const std::string se = build_seg_kernels(p, a16 && atoi(a16) != 0, BSX_DBG_ENV("BSX_F32_INPUT") == nullptr, arch ? arch : "gfx950", &scode, &scached, &sbytes);
      char sm[256];
      if (se.empty()) {
        long worst = 0;
        for (const char* k : {"bsx_seg_head", "bsx_seg_k2", "bsx_seg_k3", "bsx_seg_tail"}) worst = std::max(worst, code_object_scratch_bytes(scode, k));
        snprintf(sm, sizeof sm, "; segment kernels %s (%zu bytes of source, %zu bytes of code object, %ld B of scratch)", scached ? "cached" : "compiled", sbytes, scode.size(), worst);
        seg_msg = sm;
      } else { seg_msg = "; segment kernels: " + se; if (se.find("hipRTC:") != std::string::npos) { snprintf(msg, cap, "%s", seg_msg.c_str()); return BSX_EMODEL; } }
    }
    snprintf(msg, cap, "%s (%zu bytes of source, %zu bytes of code object, %ld B of scratch%s, cache %s)%s", mb.cached ? "cached" : "compiled", mb.source.size(), mb.code.size(), mb.scratch,
             mb.opaque_tid ? ", lane indices re-derived per op" : "", rtc_cache_dir().c_str(), seg_msg.c_str());
    return BSX_OK;
  } catch (...) { snprintf(msg, cap, "exception while reading the model"); return BSX_EMODEL; }
}

const char* bsx_plan_describe(bsx_ctx* c) { return c ? c->plan_text.c_str() : ""; }

// BSX_ACT16 stores the arena tensors of the segmented networks as packed halves at their own strides: reading them back as f32 would return garbage without an
// error, so the inspection entry points refuse arena tensors in that mode (network input / output keep their f32 buffers).
static bool debug_tensor_readable(const bsx_ctx* c, int t) { return !c->act16 || t == c->plan.input || t == c->plan.output; }

long bsx_debug_tensor_of(bsx_ctx* c, int t, int stream_idx, float* h_out, long cap) {
  if (!c || t < 0 || t >= (int)c->graph.tensors.size() || c->plan.tensor_off[t] < 0 || stream_idx < 0 || stream_idx >= c->n_streams) return BSX_EINVAL;
  if (h_out && cap < 0) return BSX_EINVAL;
  if (!debug_tensor_readable(c, t)) { c->last_error = "bsx_debug_tensor: arena tensors are stored as f16 under BSX_ACT16 and are not readable through this entry"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  const long n = (long)c->graph.tensors[t].elems();
  if (!h_out) return n;
  // network input / output and the per-launch arena are batch-major (stream i at + i * elems); the per-frame program's arena is frame-major
  const float* p = (t == c->plan.input || t == c->plan.output || !c->use_program) ? c->tensor_ptr(t) + (size_t)stream_idx * (size_t)n
                                                                                 : c->tensor_ptr(t) + (size_t)stream_idx * c->plan.arena_floats_per_stream;
  if (hipDeviceSynchronize() != hipSuccess) return BSX_EDEVICE;
  if (hipMemcpy(h_out, p, sizeof(float) * (size_t)std::min(n, cap), hipMemcpyDeviceToHost) != hipSuccess) return BSX_EDEVICE;
  return n;
}

long bsx_debug_tensor(bsx_ctx* c, int t, float* h_out, long cap) {
  if (!c || t < 0 || t >= (int)c->graph.tensors.size() || c->plan.tensor_off[t] < 0) return BSX_EINVAL;
  if (h_out && cap < 0) return BSX_EINVAL;
  if (!debug_tensor_readable(c, t)) { c->last_error = "bsx_debug_tensor: arena tensors are stored as f16 under BSX_ACT16 and are not readable through this entry"; return BSX_EINVAL; }
  DeviceGuard guard(c->device);
  long n = (long)c->graph.tensors[t].elems();
  if (!h_out) return n;
  if (hipDeviceSynchronize() != hipSuccess) return BSX_EDEVICE;
  if (hipMemcpy(h_out, c->tensor_ptr(t), sizeof(float) * (size_t)std::min(n, cap), hipMemcpyDeviceToHost) != hipSuccess) return BSX_EDEVICE;
  return n;
}

}  // extern "C"
