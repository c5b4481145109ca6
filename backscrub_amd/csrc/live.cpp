// live.cpp — the host-side pieces that turn the batch engine back into a webcam tool, built around the GPU's own queue.
//
//   bsx_background_*   what app/background.cc provides (/root/reference/app/background.cc:126-194): a still image or an animation as the
//                      compositing background, played in real time, looping at the end of the stream, and `grab` = the current picture resized to
//                      the camera size together with its frame number.  Here the pictures are decoded once (media.cpp) and kept ON THE GPU; which
//                      one is current is a pure function of the clock — picture floor(t * fps) mod n, reported as 1..n like the reference's counter
//                      of pictures read — so there is no reader thread, no lock and no per-frame upload: a grab is one resize launch.
//   bsx_live_*         what class CalcMask provides (/root/reference/app/deepseg.cc:159-286): the camera loop hands over frames and polls for
//                      masks and never waits for the segmentation; a mask may lag its frame.  Here the worker is the GPU queue itself: a frame
//                      is copied into a pinned slot and its upload, the whole mask pipeline and the mask's download are enqueued on a private
//                      HIP stream with an event behind them; `get_output_mask` polls that event.  Two submissions may be in flight; a frame that
//                      arrives while both are busy replaces the waiting one (the newest frame wins, as with the reference's double buffer).
//
// Only the C ABI of bsx.h and the HIP runtime are used here; nothing touches a kernel directly.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsx.h"
#include "media.hpp"

using namespace bsx;

namespace {

struct OnDevice {                      // the caller's current device is restored on scope exit (HIP's current device is per thread)
  int prev = -1, dev;
  explicit OnDevice(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~OnDevice() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

using Clock = std::chrono::steady_clock;

}  // namespace

struct bsx_background {
  bsx_ctx* ctx = nullptr;
  int debug = 0, device = 0;
  bool animated = false;
  int width = 0, height = 0, n_pictures = 0;
  double fps = 0;
  uint8_t* d_pictures = nullptr;        // [n_pictures][height][width][3] BGR on the context's GPU
  Clock::time_point t0;                 // start of playback
};

struct bsx_live {
  static constexpr int kInFlight = 2;
  struct Slot {
    uint8_t* h_frame = nullptr;         // pinned: the caller's frame, cloned (deepseg.cc:273)
    uint8_t* h_mask = nullptr;          // pinned: the finished mask
    uint8_t* d_frame = nullptr;
    hipEvent_t begun = nullptr, done = nullptr;
    bool busy = false;                  // submitted, mask not handed out yet
    unsigned long seq = 0;              // submission order
  };
  bsx_ctx* ctx = nullptr;
  int width = 0, height = 0, device = 0;
  hipStream_t stream = nullptr;
  Slot slot[kInFlight];
  uint8_t* h_waiting = nullptr;         // pinned: the newest frame that found both slots busy
  bool waiting = false;
  unsigned long next_seq = 1;
  bool failed = false;
  long idle_ns = 0, gpu_ns = 0;         // between the last completion and the next submission / upload + pipeline + download of the last mask
  Clock::time_point last_done;
  std::mutex mu;
};

namespace {

// enqueue: upload, mask pipeline for stream slot 0 of the context, download — all asynchronous on the live stream
bool submit(bsx_live* l, bsx_live::Slot& s) {
  const size_t fb = (size_t)l->width * l->height * 3, mb = (size_t)l->width * l->height;
  if (hipEventRecord(s.begun, l->stream) != hipSuccess) return false;
  if (hipMemcpyAsync(s.d_frame, s.h_frame, fb, hipMemcpyHostToDevice, l->stream) != hipSuccess) return false;
  if (bsx_process_batch(l->ctx, s.d_frame, 1, nullptr, l->stream) != BSX_OK) return false;
  if (hipMemcpyAsync(s.h_mask, bsx_masks_device(l->ctx), mb, hipMemcpyDeviceToHost, l->stream) != hipSuccess) return false;
  if (hipEventRecord(s.done, l->stream) != hipSuccess) return false;
  s.busy = true;
  s.seq = l->next_seq++;
  l->idle_ns = (long)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - l->last_done).count();
  return true;
}

bsx_live::Slot* free_slot(bsx_live* l) {
  for (auto& s : l->slot) if (!s.busy) return &s;
  return nullptr;
}

void release(bsx_live* l) {
  OnDevice dev(l->device);
  if (l->stream) (void)hipStreamSynchronize(l->stream);
  for (auto& s : l->slot) {
    if (s.h_frame) (void)hipHostFree(s.h_frame);
    if (s.h_mask) (void)hipHostFree(s.h_mask);
    if (s.d_frame) (void)hipFree(s.d_frame);
    if (s.begun) (void)hipEventDestroy(s.begun);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (l->h_waiting) (void)hipHostFree(l->h_waiting);
  if (l->stream) (void)hipStreamDestroy(l->stream);
  delete l;
}

}  // namespace

extern "C" {

bsx_background* bsx_background_from_frames(bsx_ctx* ctx, const uint8_t* h_bgr, int width, int height, int n_frames, double fps, int debug) {
  if (!ctx || !h_bgr || width <= 0 || height <= 0 || n_frames <= 0) return nullptr;
  bsx_info info;
  if (bsx_get_info(ctx, &info) != BSX_OK) return nullptr;
  std::unique_ptr<bsx_background> b(new bsx_background);
  b->ctx = ctx; b->debug = debug; b->device = info.device; b->width = width; b->height = height; b->n_pictures = n_frames; b->fps = fps;
  const size_t bytes = (size_t)n_frames * width * height * 3;
  {
    OnDevice dev(info.device);
    if (hipMalloc(&b->d_pictures, bytes) != hipSuccess || hipMemcpy(b->d_pictures, h_bgr, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      if (debug) fprintf(stderr, "background: cannot place %zu bytes on the GPU\n", bytes);
      if (b->d_pictures) (void)hipFree(b->d_pictures);
      return nullptr;
    }
  }
  b->animated = n_frames >= 2 && fps > 0;                       // "if: can read 2 video frames => it's a video" (background.cc:143-153)
  b->t0 = Clock::now();
  if (debug) fprintf(stderr, "background properties:\n\tvid: %s\n\tfps: %f\n\tcnt: %d\n", b->animated ? "yes" : "no", fps, n_frames);
  return b.release();
}

bsx_background* bsx_background_load(bsx_ctx* ctx, const char* path, int debug) {
  if (!ctx || !path) return nullptr;
  try {
    Media m;
    std::string err;
    if (!media_load(path, &m, &err)) { if (debug) fprintf(stderr, "background: %s\n", err.c_str()); return nullptr; }
    std::vector<uint8_t> all;
    all.reserve(m.frames.size() * m.frames[0].size());
    for (const auto& f : m.frames) all.insert(all.end(), f.begin(), f.end());
    return bsx_background_from_frames(ctx, all.data(), m.width, m.height, (int)m.frames.size(), m.fps, debug);
  } catch (...) {
    if (debug) fprintf(stderr, "background: exception while loading\n");
    return nullptr;
  }
}

void bsx_background_free(bsx_background* b) {
  if (!b) return;
  { OnDevice dev(b->device); if (b->d_pictures) (void)hipFree(b->d_pictures); }
  delete b;
}

int bsx_background_info(const bsx_background* b, int* width, int* height, int* n_frames, double* fps, int* is_video) {
  if (!b) return BSX_EINVAL;
  if (width) *width = b->width;
  if (height) *height = b->height;
  if (n_frames) *n_frames = b->n_pictures;
  if (fps) *fps = b->fps;
  if (is_video) *is_video = b->animated ? 1 : 0;
  return BSX_OK;
}

// grab_background (background.cc:178-194): the current picture resized to width x height; returns its frame number — 1 for a still, and for an
// animation the reference's count of pictures read since the last rewind (picture c is reported as c + 1; the reference's 0 exists only for the
// instant between the failed read at the end of the stream and the first read after the rewind, :91-95)
int bsx_background_grab(bsx_background* b, int width, int height, uint8_t* d_bgr_out, void* stream) {
  if (!b || !d_bgr_out || width <= 0 || height <= 0) return -1;
  long c = 0;
  if (b->animated) {
    const double t = std::chrono::duration<double>(Clock::now() - b->t0).count();
    c = (long)(t * b->fps) % b->n_pictures;
  }
  const uint8_t* src = b->d_pictures + (size_t)c * b->width * b->height * 3;
  if (bsx_resize_bgr(b->ctx, src, b->width, b->height, d_bgr_out, width, height, 1, stream) != BSX_OK) return -1;
  return (int)c + 1;
}

bsx_live* bsx_live_new(bsx_ctx* ctx) {
  bsx_info info;
  if (!ctx || bsx_get_info(ctx, &info) != BSX_OK) return nullptr;
  bsx_live* l = new bsx_live;
  l->ctx = ctx; l->width = info.width; l->height = info.height; l->device = info.device;
  const size_t fb = (size_t)info.width * info.height * 3, mb = (size_t)info.width * info.height;
  OnDevice dev(info.device);
  bool ok = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) == hipSuccess && hipHostMalloc((void**)&l->h_waiting, fb, hipHostMallocDefault) == hipSuccess;
  for (auto& s : l->slot)
    ok = ok && hipHostMalloc((void**)&s.h_frame, fb, hipHostMallocDefault) == hipSuccess && hipHostMalloc((void**)&s.h_mask, mb, hipHostMallocDefault) == hipSuccess &&
         hipMalloc(&s.d_frame, fb) == hipSuccess && hipEventCreate(&s.begun) == hipSuccess && hipEventCreate(&s.done) == hipSuccess;
  if (!ok) { release(l); return nullptr; }
  l->last_done = Clock::now();
  return l;
}

void bsx_live_delete(bsx_live* l) {
  if (l) release(l);
}

// CalcMask::set_input_frame (deepseg.cc:272-277): the frame is cloned; never waits for the GPU
int bsx_live_set_input_frame(bsx_live* l, const uint8_t* h_bgr, size_t stride) {
  if (!l || !h_bgr || stride < (size_t)l->width * 3) return BSX_EINVAL;
  std::lock_guard<std::mutex> hold(l->mu);
  if (l->failed) return BSX_EDEVICE;
  OnDevice dev(l->device);
  bsx_live::Slot* s = free_slot(l);
  uint8_t* dst = s ? s->h_frame : l->h_waiting;
  const size_t row = (size_t)l->width * 3;
  for (int y = 0; y < l->height; y++) memcpy(dst + (size_t)y * row, h_bgr + (size_t)y * stride, row);
  if (!s) { l->waiting = true; return BSX_OK; }               // both submissions in flight: this frame waits (and is replaced by a newer one)
  l->waiting = false;                                           // an older waiting frame is superseded
  if (!submit(l, *s)) { l->failed = true; return BSX_EDEVICE; }
  return BSX_OK;
}

// CalcMask::get_output_mask (deepseg.cc:279-285): 1 = a new mask was copied out, 0 = none yet (the caller's buffer is left alone)
int bsx_live_get_output_mask(bsx_live* l, uint8_t* h_mask, size_t stride) {
  if (!l || !h_mask || stride < (size_t)l->width) return BSX_EINVAL;
  std::lock_guard<std::mutex> hold(l->mu);
  if (l->failed) return BSX_EDEVICE;
  OnDevice dev(l->device);
  bsx_live::Slot* newest = nullptr;
  for (auto& s : l->slot) {
    if (!s.busy) continue;
    const hipError_t q = hipEventQuery(s.done);
    if (q == hipErrorNotReady) continue;
    if (q != hipSuccess) { l->failed = true; return BSX_EDEVICE; }
    if (!newest || s.seq > newest->seq) newest = &s;
  }
  int got = 0;
  if (newest) {
    for (int y = 0; y < l->height; y++) memcpy(h_mask + (size_t)y * stride, newest->h_mask + (size_t)y * l->width, (size_t)l->width);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, newest->begun, newest->done) == hipSuccess) l->gpu_ns = (long)(ms * 1e6);
    for (auto& s : l->slot) if (s.busy && s.seq <= newest->seq) s.busy = false;      // an older finished mask is superseded by the one handed out
    l->last_done = Clock::now();
    got = 1;
  }
  if (l->waiting) {                                             // a frame that found both slots busy goes out as soon as one is free
    if (bsx_live::Slot* s = free_slot(l)) {
      std::swap(s->h_frame, l->h_waiting);
      l->waiting = false;
      if (!submit(l, *s)) { l->failed = true; return BSX_EDEVICE; }
    }
  }
  return got;
}

// the reference's two loop timers (deepseg.cc:186,213): time spent waiting for a frame, time of a whole iteration.  Here: how long the GPU queue
// sat idle before the last submission, and upload + mask pipeline + download of the last mask handed out (HIP events).
int bsx_live_timings(const bsx_live* l, long* waitns, long* loopns) {
  if (!l) return BSX_EINVAL;
  if (waitns) *waitns = l->idle_ns;
  if (loopns) *loopns = l->gpu_ns;
  return BSX_OK;
}

// host-only decode (no GPU, no context): what bsx_background_load would place on the GPU.  Returns the number of frames (> 0) or a
// negative BSX_E* code; *h_bgr (malloc'ed, [n][h][w][3]) is the caller's to free with bsx_media_free.
int bsx_media_decode(const char* path, int* width, int* height, double* fps, uint8_t** h_bgr, char* errbuf, size_t errcap) {
  if (!path || !width || !height || !h_bgr) return BSX_EINVAL;
  try {
    Media m;
    std::string err;
    if (!media_load(path, &m, &err)) { if (errbuf && errcap) snprintf(errbuf, errcap, "%s", err.c_str()); return BSX_EMODEL; }
    const size_t fb = (size_t)m.width * m.height * 3;
    uint8_t* out = (uint8_t*)malloc(fb * m.frames.size());
    if (!out) return BSX_EDEVICE;
    for (size_t i = 0; i < m.frames.size(); i++) memcpy(out + i * fb, m.frames[i].data(), fb);
    *width = m.width; *height = m.height; *h_bgr = out;
    if (fps) *fps = m.fps;
    return (int)m.frames.size();
  } catch (...) { if (errbuf && errcap) snprintf(errbuf, errcap, "exception while decoding"); return BSX_EMODEL; }
}
void bsx_media_free(uint8_t* h_bgr) { free(h_bgr); }

}  // extern "C"
