/* bsx.h — C ABI of libbsx.so, the MI355X-native backscrub hot path.
 *
 * Plain C: opaque handle, raw pointers, sizes.  No C++ / OpenCV / torch types cross this
 * boundary.  `d_` pointers are HIP device pointers on the context's GPU; `h_` pointers are
 * host memory.  `stream` is a hipStream_t passed as void* (NULL = the HIP default stream, as in every HIP API).
 * All functions return 0 on success or a negative BSX_E* code; nothing throws.
 *
 * What each entry point replaces in the reference (/root/reference):
 *
 *   bsx_version            bs_tensorflow_version()        lib/libbackscrub.h:13,  .cc:150-152
 *   bsx_new                bs_maskgen_new()               lib/libbackscrub.h:16-33, .cc:161-259
 *   bsx_delete             bs_maskgen_delete()            lib/libbackscrub.h:36,  .cc:261-277
 *   bsx_process_host       bs_maskgen_process()           lib/libbackscrub.h:39,  .cc:279-376
 *                          (one stream, host frame in, host mask out — what the C++ shim
 *                          csrc/bs_maskgen_shim.cpp forwards cv::Mat data to)
 *   bsx_process_batch      the same function for n_streams device-resident frames at once
 *   bsx_composite_batch    alpha_blend()                  app/deepseg.cc:108-134 (file-static)
 *   bsx_step_batch         one main-loop iteration        app/deepseg.cc:634-661
 *   bsx_step_batch_yuyv    … with convert_rgb_to_yuyv fused app/deepseg.cc:634-681
 *   bsx_step_batch_ex      … with cv::flip (and YUYV) fused  app/deepseg.cc:667-681; BSX_STEP_YUYV_IN: … and VideoCapture's YUYV->BGR  app/deepseg.cc:553,725
 *   bsx_step_batch_pipelined  … with the CalcMask worker's overlap of segmentation and blending  app/deepseg.cc:159-285, 634-661
 *                          (set_input_frame → mask → alpha_blend), batched
 *   bsx_resize_bgr         grab_background() cv::resize   app/background.cc:178-194
 *   bsx_bgr_to_yuyv        convert_rgb_to_yuyv()          app/deepseg.cc:87-106
 *   bsx_yuyv_to_bgr        VideoCapture's YUYV->BGR       app/deepseg.cc:553,725 (cv::COLOR_YUV2BGR_YUYV)
 *   bsx_flip_bgr           cv::flip of the output frame   app/deepseg.cc:667-673 (flipHorizontal / flipVertical)
 *   bsx_gaussian_blur_bgr  cv::GaussianBlur of the background app/deepseg.cc:415-431,652-658 (-p bgblur:<n>: blur the camera frame itself
 *                          (or the background image) and composite over it)
 *   bsx_background_*       load_background / grab_background + reader thread   app/background.cc:29-104,126-194
 *   bsx_live_*             class CalcMask (worker thread, double buffering)     app/deepseg.cc:159-286
 *   bsx_profile_batch      the per-stage timers           app/deepseg.cc:137-156,701-720 (timinginfo_t)
 *   bsx_get_info           the geometry of backscrub_ctx_t lib/libbackscrub.cc:28-54,234-246
 *
 * Threading: a context is NOT thread-safe (same as the reference: one context per caller
 * thread, lib/libbackscrub.cc has no locks).  Callbacks fire synchronously on the calling
 * thread in the order prep → infer → mask, once per process call (per batch for the batched
 * calls), see lib/libbackscrub.cc:303,311,363.
 */
#ifndef BSX_H_
#define BSX_H_

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY exported symbols (tests/test_cabi.py checks `nm -D`), so
 * linking it into an application adds nothing but `bsx_*` to that application's symbol namespace. */
#ifndef BSX_API
#define BSX_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsx_ctx bsx_ctx;

typedef void (*bsx_debug_fn)(void* caller_ctx, const char* msg);
typedef void (*bsx_stage_fn)(void* caller_ctx);

enum {
  BSX_OK = 0,
  BSX_EINVAL = -1,   /* NULL context / bad argument (reference: `return false`, .cc:280) */
  BSX_EMODEL = -2,   /* model file unreadable / unsupported op / unknown model type (.cc:191-203) */
  BSX_EDEVICE = -3,  /* HIP error (message via ondebug or stderr) */
  BSX_ESIZE = -4     /* frame geometry does not match the one given to bsx_new */
};

/* model types, as sniffed from the file name by the reference (lib/libbackscrub.cc:116-130) */
enum { BSX_MODEL_UNKNOWN = 0, BSX_MODEL_DEEPLAB = 1, BSX_MODEL_MLKIT = 2, BSX_MODEL_MEET = 3, BSX_MODEL_BODYPIX = 4 };

typedef struct bsx_info {
  int model_type;
  int width, height;              /* frame geometry */
  int n_streams;                  /* batch capacity */
  int in_w, in_h, in_c;           /* model input tensor */
  int out_w, out_h, out_c;        /* model output tensor */
  int roi[4];                     /* roidim   x,y,w,h in the frame       (.cc:234-246) */
  int in_roi[4];                  /* in_roidim x,y,w,h in the model canvas */
  int n_ops, n_steps;             /* graph operators / fused GPU launches per batch */
  int device;
  float norm_scale, norm_offset;  /* .cc:132-148 */
  double nn_flops_per_frame;      /* 2*MAC of the loaded graph */
  size_t act_bytes_per_stream;    /* activation arena per stream */
} bsx_info;

/* "bsx <ver> (HIP gfx950)"; static storage. */
BSX_API const char* bsx_version(void);

/* Number of visible HIP devices (0 if none / runtime missing). */
BSX_API int bsx_device_count(void);

/* Create a mask generator for `n_streams` independent camera streams of width x height BGR
 * frames on HIP device `device`.  `threads` is accepted for signature parity with
 * bs_maskgen_new (intra-op CPU threads there) and recorded only.  Callbacks may be NULL.
 * 1 <= n_streams <= 65535 (the stream index rides in a grid dimension).
 * Returns NULL on failure after reporting through ondebug (or stderr), like the reference. */
BSX_API bsx_ctx* bsx_new(const char* model_path, size_t threads, size_t width, size_t height,
                 int n_streams, int device,
                 bsx_debug_fn ondebug, bsx_stage_fn onprep, bsx_stage_fn oninfer, bsx_stage_fn onmask,
                 void* caller_ctx);

/* NULL-safe. */
BSX_API void bsx_delete(bsx_ctx* ctx);

BSX_API int bsx_get_info(const bsx_ctx* ctx, bsx_info* out);

/* Last error text for this context (or the global one if ctx==NULL); static/ctx storage. */
BSX_API const char* bsx_last_error(const bsx_ctx* ctx);

/* Reset the per-stream temporal state (`ofinal` → 0, `mask` → 255) of all streams. */
BSX_API int bsx_reset(bsx_ctx* ctx, void* stream);

/* Drop-in single-frame path.  h_bgr: height rows of width*3 bytes, `bgr_stride` bytes apart
 * (CV_8UC3 cv::Mat data/step).  h_mask: height rows of width bytes, `mask_stride` apart;
 * receives the full-frame mask (255 = background).  Uses stream slot `stream_idx`.
 * Synchronous: returns after the mask is in h_mask. */
BSX_API int bsx_process_host(bsx_ctx* ctx, int stream_idx, const uint8_t* h_bgr, size_t bgr_stride,
                     uint8_t* h_mask, size_t mask_stride);

/* Batched device path: frames [n][height][width][3] u8 contiguous (n <= n_streams; frame i
 * belongs to stream i).  Updates each stream's temporal state and its persistent full-frame
 * mask.  If d_masks != NULL the masks are also copied there ([n][height][width]).
 * Asynchronous on `stream` unless callbacks are set (each callback needs a stream sync). */
BSX_API int bsx_process_batch(bsx_ctx* ctx, const uint8_t* d_frames, int n, uint8_t* d_masks, void* stream);

/* Device pointer of the persistent masks [n_streams][height][width] (valid until bsx_delete;
 * contents valid after the process call that produced them has completed on its stream) —
 * the analogue of `mask = ctx.mask` aliasing the lib-owned buffer (lib/libbackscrub.cc:374). */
BSX_API uint8_t* bsx_masks_device(bsx_ctx* ctx);

/* out = (bg*m + frame*(255-m))/255 per byte (C truncating divide).  d_bg is one
 * [height][width][3] image shared by all frames when bg_frame_stride == 0, else frame i uses
 * d_bg + i*bg_frame_stride.  d_masks == NULL means "use the context's persistent masks". */
BSX_API int bsx_composite_batch(bsx_ctx* ctx, const uint8_t* d_bg, size_t bg_frame_stride,
                        const uint8_t* d_frames, const uint8_t* d_masks, uint8_t* d_out, int n, void* stream);

/* bsx_process_batch followed by bsx_composite_batch on the same stream. */
BSX_API int bsx_step_batch(bsx_ctx* ctx, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride,
                   uint8_t* d_out, int n, void* stream);

/* The same main-loop iteration with the composite leaving as YUYV 4:2:2 [n][height][width][2] (bytes Y0 V Y1 U): convert_rgb_to_yuyv
 * (app/deepseg.cc:87-106, applied at :681 right after alpha_blend) runs in the blend's epilogue — 2 B/px written instead of 3, and no separate
 * pass over the composite — for callers that feed a V4L2 YUYV sink.  Bit-identical to bsx_step_batch followed by bsx_bgr_to_yuyv.  width even. */
BSX_API int bsx_step_batch_yuyv(bsx_ctx* ctx, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride,
                        uint8_t* d_out_yuyv, int n, void* stream);

/* The same iteration with the post steps of the main loop that sit between alpha_blend and the device write folded into WHERE the blend stores
 * its result (app/deepseg.cc:667-681): flags = BSX_STEP_FLIP_H | BSX_STEP_FLIP_V (cv::flip(raw, raw, 1 / 0 / -1) of the composite: the four pixels
 * a lane composites go to the mirrored column group in reverse order, the row to the mirrored row — no extra pass over the frame) and / or
 * BSX_STEP_YUYV (convert_rgb_to_yuyv of the — flipped — composite, as bsx_step_batch_yuyv).  flags = 0 is bsx_step_batch.  The persistent masks
 * are those of the unflipped camera frame, as in the reference.  Bit-identical to bsx_step_batch + bsx_flip_bgr [+ bsx_bgr_to_yuyv]. */
/* Aliasing: d_out == d_frames (the reference flips and composites `raw` in place, app/deepseg.cc:661-673) is allowed for every flag combination — with a flip or
 * YUYV flag and overlapping buffers the library composites into its own scratch first (the fused kernel would store to addresses another tile has not read
 * yet), so in-place costs one extra pass there; a plain composite runs in place at full speed.  Buffers that overlap PARTIALLY return BSX_EINVAL, and so does
 * BSX_STEP_BGBLUR with d_out == d_frames (every output pixel needs a neighbourhood of input pixels). */
#define BSX_STEP_YUYV 1u
#define BSX_STEP_FLIP_H 2u
#define BSX_STEP_FLIP_V 4u
/* BSX_STEP_NO_MASK: composite only — the full-resolution mask of this frame is formed in registers, blended and NOT stored (1 of the step's 7 HBM bytes per pixel).
 * Nothing later depends on it: every frame's mask is rebuilt from the model-resolution temporal state, which advances as usual; bsx_masks_device() then keeps
 * the last mask a call WITHOUT this flag stored.  For main loops that only need the composite (app/deepseg.cc:661 uses the mask for nothing else unless -d -d). */
#define BSX_STEP_NO_MASK 8u
/* BSX_STEP_YUYV_IN: d_frames holds the camera's RAW YUYV 4:2:2 frames, [n][height][width][2] bytes Y0 U Y1 V — what a V4L2 webcam delivers and what the reference's
 * capture converts with cv::COLOR_YUV2BGR_YUYV (CAP_PROP_CONVERT_RGB, app/deepseg.cc:553; explicit form :725) — instead of packed BGR.  The conversion (the integers
 * of bsx_yuyv_to_bgr) is folded into the two kernels that read the frame: the prep kernel converts the two taps of every resize sample, the mask tile kernel the four
 * pixels it composites, so the 3 B/px BGR frame is never written or read.  With BSX_STEP_YUYV the step is YUYV in -> YUYV out: 7 B/px of device traffic instead of 9
 * (frame 2 + background 3 in, composite 2 out) and 4 B/px over PCIe instead of 6.  Results are bit-identical to bsx_yuyv_to_bgr followed by the same step on the BGR
 * frames; that two-pass sequence is what runs, through a context-owned scratch, where the fused kernels do not apply (BSX_STEP_BGBLUR, a ROI that starts on an odd
 * column, width % 4 != 0, overlapping or unaligned buffers, an onmask callback).  width must be even.  Also accepted by bsx_step_batch_pipelined. */
#define BSX_STEP_YUYV_IN 16u
/* BSX_STEP_BGBLUR(ksize): the background of every stream is cv::GaussianBlur(its own camera frame, Size(ksize, ksize), 0) — `-p bgblur:<ksize>` without `-b`,
 * the reference's default way to run (app/deepseg.cc:652-661).  d_bg is ignored (may be NULL).  The blur and the alpha blend are ONE pass over the frames: each
 * tile of the blurred frame is composited while it is still in LDS, so the blurred image is never written or read back and the frame is read once, not twice.
 * Bit-identical to bsx_gaussian_blur_bgr into a per-stream background + bsx_step_batch_ex(flags) with bg_frame_stride = one frame; that two-call sequence is what
 * runs, on a context-owned scratch background, when the single pass does not apply (YUYV / flip flags, width % 4 != 0, unaligned buffers, ksize 1).  ksize odd, 1..31. */
#define BSX_STEP_BGBLUR(ksize) (((unsigned)(ksize) & 255u) << 8)
BSX_API int bsx_step_batch_ex(bsx_ctx* ctx, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride,
                      uint8_t* d_out, int n, void* stream, unsigned flags);

/* Throughput mode — the same main-loop iteration as a TWO-DEEP PIPELINE.  The reference runs its two halves concurrently: CalcMask::run() segments on a
 * worker thread (app/deepseg.cc:182-216) while the capture loop blends and writes (:634-681).  Call k enqueues the mask pipeline (prep → network → decode /
 * temporal filter) of batch k on `stream` and, concurrently on a context-owned low-priority stream, the mask up-scale + blur + composite of batch k - 1 — the
 * HBM-bound half fills the gaps of the latency-bound half instead of waiting behind it.  Every frame is still composited with ITS OWN mask (the reference's
 * loop blends with whatever mask is newest): outputs, persistent masks and temporal state are bit-identical to bsx_step_batch_ex(flags), one call later.
 *   - d_out(k) and bsx_masks_device() hold batch k's results once call k + 1 (or the flush) has completed on `stream`;
 *   - d_frames(k), d_bg(k) and d_out(k) must stay valid and unmodified until then; d_out must not overlap d_frames;
 *   - d_frames == NULL flushes: the pending composite runs on `stream` (all other arguments ignored); bsx_reset drops it;
 *   - `stream` may differ from call to call (double-buffered callers): every call records the end of its mask pipeline on its own stream, and the composite of
 *     that batch, the next call's network (it reuses the arena) and the flush wait for THAT event — the caller orders nothing across its streams;
 *   - flags: BSX_STEP_YUYV | BSX_STEP_FLIP_H | BSX_STEP_FLIP_V | BSX_STEP_NO_MASK | BSX_STEP_YUYV_IN (no BSX_STEP_BGBLUR); the geometry must be the fused tile kernel's
 *     (width, roi.x, roi.w multiples of 4, 4-byte aligned buffers) and no stage callback may be set — otherwise BSX_EINVAL, as does every other entry point
 *     that advances the temporal state while a composite is pending (bsx_process_batch / _host, bsx_step_batch*, bsx_profile_batch, bsx_debug_run_stage 1-3). */
BSX_API int bsx_step_batch_pipelined(bsx_ctx* ctx, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride,
                             uint8_t* d_out, int n, void* stream, unsigned flags);

/* cv::resize(src, dst, Size(dw,dh)) with INTER_LINEAR on packed BGR u8 (device pointers, n images). */
BSX_API int bsx_resize_bgr(bsx_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh, int n, void* stream);

/* BGR u8 [n][h][w][3] -> YUYV 4:2:2 [n][h][w][2] exactly as convert_rgb_to_yuyv (byte order Y0 V Y1 U). */
BSX_API int bsx_bgr_to_yuyv(bsx_ctx* ctx, const uint8_t* d_bgr, uint8_t* d_yuyv, int w, int h, int n, void* stream);

/* YUYV 4:2:2 (bytes Y0 U Y1 V) [n][h][w][2] -> BGR u8 [n][h][w][3], exactly cv::cvtColor(COLOR_YUV2BGR_YUYV): the conversion
 * cv::VideoCapture applies to raw camera frames for the reference (app/deepseg.cc:553, :725).  Lets a caller upload 2 B/px. */
BSX_API int bsx_yuyv_to_bgr(bsx_ctx* ctx, const uint8_t* d_yuyv, uint8_t* d_bgr, int w, int h, int n, void* stream);

/* cv::flip(src, dst, code) on packed BGR u8 [n][h][w][3] (device pointers, dst != src): code 0 flips around the x axis
 * (-v / flipVertical), code > 0 around the y axis (-h / flipHorizontal), code < 0 both (app/deepseg.cc:667-673). */
BSX_API int bsx_flip_bgr(bsx_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int code, void* stream);

/* cv::GaussianBlur(src, dst, Size(ksize, ksize), 0) on packed BGR u8 [n][h][w][3] (device pointers, dst != src), BORDER_REFLECT_101,
 * OpenCV's 8-bit fixed-point coefficients; ksize odd, 1 <= ksize <= 31 (the reference's default strength is 25, app/deepseg.cc:429).
 * The "blur my own room" mode of the reference = this on the camera frames, then bsx_step_batch with bg_frame_stride = one frame. */
BSX_API int bsx_gaussian_blur_bgr(bsx_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int ksize, void* stream);

/* ---- background source (app/background.cc) ----
 * load_background(): a still image or an animation (GIF87a/89a, 8-bit non-interlaced PNG, binary PPM decoded in this library; other
 * formats through bsx_background_from_frames with frames the caller decoded).  The frames live on the context's GPU; an animation gets
 * the FPS-paced reader thread of background.cc:29-104 (advances one frame per 1/fps, wraps to 0 at the end).  NULL on error. */
typedef struct bsx_background bsx_background;
BSX_API bsx_background* bsx_background_load(bsx_ctx* ctx, const char* path, int debug);
BSX_API bsx_background* bsx_background_from_frames(bsx_ctx* ctx, const uint8_t* h_bgr, int width, int height, int n_frames, double fps, int debug);
BSX_API void bsx_background_free(bsx_background* bg);
BSX_API int bsx_background_info(const bsx_background* bg, int* width, int* height, int* n_frames, double* fps, int* is_video);
/* grab_background(): the current frame resized (cv::resize INTER_LINEAR) to width x height into d_bgr_out [height][width][3].
 * Returns the frame number or -1 on error: 1 for a still image; for an animation the reference's count of pictures read since the last rewind, i.e.
 * picture c (= floor(t * fps) mod n, t since the background was created: real-time playback, looping at the end) is reported as c + 1. */
BSX_API int bsx_background_grab(bsx_background* bg, int width, int height, uint8_t* d_bgr_out, void* stream);
/* host-only decode of the same formats (no GPU): frames → malloc'ed [n][h][w][3] BGR; returns n (> 0) or a negative BSX_E* code */
BSX_API int bsx_media_decode(const char* path, int* width, int* height, double* fps, uint8_t** h_bgr, char* errbuf, size_t errcap);
BSX_API void bsx_media_free(uint8_t* h_bgr);

/* ---- live single-camera mode: class CalcMask (app/deepseg.cc:159-286) ----
 * set_input_frame clones the frame into pinned memory and enqueues upload + mask pipeline (stream slot 0) + mask download on a private HIP
 * stream; it never waits for the GPU (up to two submissions in flight; a frame that finds both busy replaces the one already waiting).
 * get_output_mask polls the completion events: it copies the newest finished mask (returns 1) or leaves h_mask untouched (0).
 * timings: how long the queue sat idle before the last submission / upload + pipeline + download of the last mask handed out (HIP events). */
typedef struct bsx_live bsx_live;
BSX_API bsx_live* bsx_live_new(bsx_ctx* ctx);
BSX_API void bsx_live_delete(bsx_live* live);
BSX_API int bsx_live_set_input_frame(bsx_live* live, const uint8_t* h_bgr, size_t bgr_stride);
BSX_API int bsx_live_get_output_mask(bsx_live* live, uint8_t* h_mask, size_t mask_stride);
BSX_API int bsx_live_timings(const bsx_live* live, long* waitns, long* loopns);

/* ---- introspection used by the parity tests and the bench (stage-by-stage checks) ---- */
/* Device pointer + element count of: 0 = model input tensor [n_streams][in_h][in_w][in_c] f32,
 * 1 = model output tensor f32, 2 = ofinal u8 [n_streams][out_h][out_w], 3 = masks u8. */
BSX_API int bsx_debug_buffer(bsx_ctx* ctx, int which, void** d_ptr, size_t* bytes);
/* Run single stages on the current buffers (n streams): 0 = prep, 1 = infer, 2 = decode+IIR, 3 = upscale+blur, 4 = prep on raw YUYV 4:2:2 frames
 * (BSX_STEP_YUYV_IN's form of stage 0; BSX_EINVAL where the fused form does not apply). */
BSX_API int bsx_debug_run_stage(bsx_ctx* ctx, int stage, const uint8_t* d_frames, int n, void* stream);
/* Per-launch description of the fused plan, one line per GPU launch; returned string is owned by ctx. */
BSX_API const char* bsx_plan_describe(bsx_ctx* ctx);
/* Copy out the value of graph tensor `tensor_idx` for stream 0 after an infer (only tensors that survive
 * fusion are available); returns element count or negative error.  h_out may be NULL to query the size. */
BSX_API long bsx_debug_tensor(bsx_ctx* ctx, int tensor_idx, float* h_out, long cap);
/* The same for stream `stream_idx` of the last batch (full-batch parity tests: every stream against its twin). */
BSX_API long bsx_debug_tensor_of(bsx_ctx* ctx, int tensor_idx, int stream_idx, float* h_out, long cap);

/* How the fused mask + blend launch would classify its tiles for the CURRENT temporal state of the first n streams (host-side, exact same extents as the kernel):
 * out4 = {tiles, tiles whose whole source block is 0xFF, ... 0x00, tiles on the general path}.  A uniform tile skips the up-scale / blur phases and reads only the
 * operand its composite is a copy of (kernels_img.hip: mask_tile_k); bench.py uses the counts to state the bytes the launch really has to move. */
BSX_API int bsx_debug_mask_tile_stats(bsx_ctx* ctx, int n, long* out4);

/* Per-frame-program timeline: runs the network once for n streams and returns, for workgroup 0, the wall-clock
 * (100 MHz constant-rate counter) ticks at the start of every micro-op plus one final tick; ticks[i+1]-ticks[i] = op i.
 * Returns the number of micro-ops (cap must be >= that + 1), 0 if the program path is off, negative on error. */
BSX_API int bsx_debug_program_timeline(bsx_ctx* ctx, int n, unsigned long long* ticks, int cap, void* stream);

/* Host only, no GPU: the coefficient words gauss_blur_k multiplies with for cv::GaussianBlur(ksize, sigma 0) on 8-bit images (app/deepseg.cc:657-658) when its LDS
 * planes start `shift` (0..3) pixels left of the tile — c4[4][9]: the u8 taps packed 4 per word, delayed by j + shift bytes for output phase j; c2[2][17]: the same taps
 * as u16 pairs delayed by h halves.  The parity tests check them against the oracle's taps (the kernels' arithmetic is only as right as these tables).  Returns 0 or BSX_EINVAL. */
BSX_API int bsx_debug_gauss_coeffs(int ksize, int shift, uint32_t* c4 /* 36 */, uint32_t* c2 /* 34 */);

/* Parse a .tflite file and build the fused plan WITHOUT touching a GPU; writes a text description
 * ("ops=<n> nodes=<n> steps=<n> macs=<per frame> arena_floats=<per stream>" then one line per launch)
 * into buf (NUL-terminated, truncated to cap).  Returns 0, or BSX_EMODEL with the reason in buf. */
BSX_API int bsx_model_describe(const char* model_path, char* buf, size_t cap);
/* Host only, no GPU: build the plan of `model_path`, emit the kernel specialised to that graph (the per-frame program of the Meet / MLKit
 * family as straight-line code: csrc/gen_mid.cpp) and compile it with hipRTC for `arch` (NULL = "gfx950") into the code-object cache, so
 * that bsx_new on the GPU box only loads it — and the same for the graph's segment kernels (csrc/gen_seg.cpp).  This replaces what InterpreterBuilder / AllocateTensors do when the reference creates its
 * context (lib/libbackscrub.cc:205-217).  msg receives "compiled" / "cached" / why the graph stays interpreted.  Returns 0, or BSX_EMODEL. */
BSX_API int bsx_model_precompile(const char* model_path, const char* arch, char* msg, size_t cap);
/* The generated source itself (tests, inspection): returns its length (without NUL), copies at most cap - 1 bytes; 0 when the graph has no
 * specialised form (reason in buf), negative on a model error. */
BSX_API long bsx_model_kernel_source(const char* model_path, char* buf, size_t cap);
/* The same for the SEGMENT kernels of the graph (csrc/gen_seg.cpp: the high-resolution ends of the Meet / MLKit networks — the source of csrc/kernels_seg.hip with this
 * plan's descriptors and template arguments as compile-time constants; bsx_model_precompile compiles it next to the program's kernel, bsx_new loads it). */
BSX_API long bsx_model_seg_source(const char* model_path, char* buf, size_t cap);

/* ---- measurement ---- */
typedef struct bsx_launch_stat {
  char name[64];      /* kernel / fused step label */
  double avg_ms;      /* mean duration of this launch over `iters` repetitions: from the hipEvent behind the launch in front of it to the one behind itself */
  double bytes;       /* ALGORITHMIC bytes this launch must move for n streams (inputs once + outputs once) */
  double flops;       /* 2*MAC for n streams (0 for byte kernels) */
} bsx_launch_stat;

/* Runs the whole per-batch sequence (prep, every fused network step, decode, upscale+blur, blend)
 * `iters` times with ONE hipEvent on `stream` between consecutive launches (the per-launch figures add up to the pass), and writes one record per
 * launch (in launch order) into out[0..cap).  Returns the number of launches, or a negative error.
 * The temporal state advances exactly as `iters` calls of bsx_step_batch would. */
BSX_API int bsx_profile_batch(bsx_ctx* ctx, const uint8_t* d_frames, const uint8_t* d_bg, size_t bg_frame_stride, uint8_t* d_out,
                      int n, int iters, bsx_launch_stat* out, int cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BSX_H_ */
