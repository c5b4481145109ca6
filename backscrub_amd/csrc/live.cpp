// live.cpp — the host-side pieces that turn the batch engine back into a webcam tool:
//
//   bsx_background_*   the background source of /root/reference/app/background.cc: load_background (:126-176, still image or
//                      video), the FPS-paced reader thread that loops on end of stream (:29-104), grab_background (:178-194:
//                      current frame resized to the camera size, under the frame mutex; returns the frame number, 1 for a still,
//                      and "can loop round to 0").  Frames are decoded once (media.cpp) and kept ON THE GPU, so a grab is one
//                      bsx_resize_bgr launch — no per-frame host decode or upload.
//   bsx_live_*         CalcMask of /root/reference/app/deepseg.cc:159-286: one worker thread, double-buffered frame in /
//                      mask out, condition-variable wake-up, the three stage timers; the camera loop never waits for the mask
//                      (it may lag by a frame, exactly as in the reference).
//
// Plain C++ threads over the C ABI of bsx.h; nothing here touches a kernel directly.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsx.h"
#include "media.hpp"

using namespace bsx;

struct bsx_background {
  bsx_ctx* ctx = nullptr;
  int debug = 0;
  bool video = false;
  std::atomic<bool> run{false};
  int width = 0, height = 0, n_frames = 0;
  double fps = 0;
  uint8_t* d_frames = nullptr;          // [n_frames][height][width][3] BGR on the context's GPU
  int frame = 0;                        // current frame of the reader (guarded by rawmux)
  std::mutex rawmux;
  std::thread thread;
};

struct bsx_live {
  bsx_ctx* ctx = nullptr;
  int width = 0, height = 0;
  std::atomic<bool> running{false};
  std::vector<uint8_t> frame1, frame2, mask1, mask2;
  std::vector<uint8_t>*frame_current, *frame_next, *mask_current, *mask_out;
  std::mutex lock_frame, lock_mask;
  std::condition_variable condition_new_frame;
  bool new_frame = false;                 // guarded by lock_frame
  std::atomic<bool> new_mask{false};      // set under lock_mask, polled without it by get_output_mask (as the reference does, deepseg.cc:280)
  std::atomic<int> failed{0};
  std::thread thread;
  std::atomic<long> waitns{0}, loopns{0};
};

namespace {

long since_ns(std::chrono::steady_clock::time_point t0) {
  return (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

// background.cc:29-104 — advance one frame per 1/fps, wrap to frame 0 at the end of the stream
void reader_thread(bsx_background* b) {
  if (b->debug) fprintf(stderr, "background: thread start\n");
  auto next = std::chrono::steady_clock::now();
  while (b->run) {
    {
      std::unique_lock<std::mutex> hold(b->rawmux);
      b->frame += 1;
      if (b->frame >= b->n_frames) b->frame = 0;          // no more frames: reset position and go again (:91-95)
    }
    next += std::chrono::nanoseconds((long)(1e9 / b->fps));
    while (b->run && std::chrono::steady_clock::now() < next) {
      const auto left = next - std::chrono::steady_clock::now();
      std::this_thread::sleep_for(left < std::chrono::milliseconds(20) ? left : std::chrono::milliseconds(20));   // stays responsive to `run`
    }
  }
  if (b->debug) fprintf(stderr, "background: thread stop\n");
}

// deepseg.cc:182-216
void live_thread(bsx_live* l) {
  while (l->running) {
    const auto tloop = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> hold(l->lock_frame);
      while (!l->new_frame) l->condition_new_frame.wait(hold);
      l->new_frame = false;
      std::swap(l->frame_next, l->frame_current);
    }
    l->waitns = since_ns(tloop);
    if (!l->running) break;
    if (bsx_process_host(l->ctx, 0, l->frame_current->data(), (size_t)l->width * 3, l->mask_current->data(), (size_t)l->width) != BSX_OK) {
      l->failed = 1;                                        // the reference exits the process here (:203-206); a library reports instead
      break;
    }
    {
      std::unique_lock<std::mutex> hold(l->lock_mask);
      std::swap(l->mask_out, l->mask_current);
      l->new_mask = true;
    }
    l->loopns = since_ns(tloop);
  }
}

}  // namespace

extern "C" {

bsx_background* bsx_background_from_frames(bsx_ctx* ctx, const uint8_t* h_bgr, int width, int height, int n_frames, double fps, int debug) {
  if (!ctx || !h_bgr || width <= 0 || height <= 0 || n_frames <= 0) return nullptr;
  bsx_info info;
  if (bsx_get_info(ctx, &info) != BSX_OK) return nullptr;
  std::unique_ptr<bsx_background> b(new bsx_background);
  b->ctx = ctx; b->debug = debug; b->width = width; b->height = height; b->n_frames = n_frames; b->fps = fps;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(info.device);
  const size_t bytes = (size_t)n_frames * width * height * 3;
  bool ok = hipMalloc(&b->d_frames, bytes) == hipSuccess && hipMemcpy(b->d_frames, h_bgr, bytes, hipMemcpyHostToDevice) == hipSuccess;
  if (prev >= 0) (void)hipSetDevice(prev);
  if (!ok) { if (debug) fprintf(stderr, "background: cannot place %zu bytes on the GPU\n", bytes); if (b->d_frames) (void)hipFree(b->d_frames); return nullptr; }
  // "if: can read 2 video frames => it's a video" (background.cc:143-153)
  b->video = n_frames >= 2 && fps > 0;
  if (b->video) { b->run = true; b->thread = std::thread(reader_thread, b.get()); }
  if (debug) fprintf(stderr, "background properties:\n\tvid: %s\n\tfps: %f\n\tcnt: %d\n", b->video ? "yes" : "no", fps, n_frames);
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

void bsx_background_free(bsx_background* b) {                 // drop_background, background.cc:106-124
  if (!b) return;
  if (b->video && b->run) { b->run = false; b->thread.join(); }
  if (b->d_frames) (void)hipFree(b->d_frames);
  delete b;
}

int bsx_background_info(const bsx_background* b, int* width, int* height, int* n_frames, double* fps, int* is_video) {
  if (!b) return BSX_EINVAL;
  if (width) *width = b->width;
  if (height) *height = b->height;
  if (n_frames) *n_frames = b->n_frames;
  if (fps) *fps = b->fps;
  if (is_video) *is_video = b->video ? 1 : 0;
  return BSX_OK;
}

int bsx_background_grab(bsx_background* b, int width, int height, uint8_t* d_bgr_out, void* stream) {
  if (!b || !d_bgr_out || width <= 0 || height <= 0) return -1;
  int frm = 1;
  const uint8_t* src = b->d_frames;
  if (b->video) {                                             // grab frame & frame no. under the mutex (:183-188)
    std::unique_lock<std::mutex> hold(b->rawmux);
    frm = b->frame;
    src = b->d_frames + (size_t)frm * b->width * b->height * 3;
  }
  if (bsx_resize_bgr(b->ctx, src, b->width, b->height, d_bgr_out, width, height, 1, stream) != BSX_OK) return -1;
  return frm;
}

bsx_live* bsx_live_new(bsx_ctx* ctx) {
  bsx_info info;
  if (!ctx || bsx_get_info(ctx, &info) != BSX_OK) return nullptr;
  std::unique_ptr<bsx_live> l(new bsx_live);
  l->ctx = ctx; l->width = info.width; l->height = info.height;
  const size_t fb = (size_t)info.width * info.height * 3, mb = (size_t)info.width * info.height;
  l->frame1.assign(fb, 0); l->frame2.assign(fb, 0); l->mask1.assign(mb, 255); l->mask2.assign(mb, 255);
  l->frame_next = &l->frame1; l->frame_current = &l->frame2; l->mask_current = &l->mask1; l->mask_out = &l->mask2;
  l->running = true;
  l->thread = std::thread(live_thread, l.get());
  return l.release();
}

void bsx_live_delete(bsx_live* l) {                            // ~CalcMask, deepseg.cc:261-270
  if (!l) return;
  l->running = false;
  { std::lock_guard<std::mutex> hold(l->lock_frame); l->new_frame = true; }
  l->condition_new_frame.notify_all();
  l->thread.join();
  delete l;
}

int bsx_live_set_input_frame(bsx_live* l, const uint8_t* h_bgr, size_t stride) {   // deepseg.cc:272-277 (frame.clone())
  if (!l || !h_bgr || stride < (size_t)l->width * 3) return BSX_EINVAL;
  if (l->failed) return BSX_EDEVICE;
  std::lock_guard<std::mutex> hold(l->lock_frame);
  for (int y = 0; y < l->height; y++) memcpy(l->frame_next->data() + (size_t)y * l->width * 3, h_bgr + (size_t)y * stride, (size_t)l->width * 3);
  l->new_frame = true;
  l->condition_new_frame.notify_all();
  return BSX_OK;
}

int bsx_live_get_output_mask(bsx_live* l, uint8_t* h_mask, size_t stride) {        // deepseg.cc:279-285: 1 = a new mask was copied out
  if (!l || !h_mask || stride < (size_t)l->width) return BSX_EINVAL;
  if (l->failed) return BSX_EDEVICE;
  if (!l->new_mask) return 0;
  std::lock_guard<std::mutex> hold(l->lock_mask);
  for (int y = 0; y < l->height; y++) memcpy(h_mask + (size_t)y * stride, l->mask_out->data() + (size_t)y * l->width, (size_t)l->width);
  l->new_mask = false;
  return 1;
}

int bsx_live_timings(const bsx_live* l, long* waitns, long* loopns) {
  if (!l) return BSX_EINVAL;
  if (waitns) *waitns = l->waitns;
  if (loopns) *loopns = l->loopns;
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
