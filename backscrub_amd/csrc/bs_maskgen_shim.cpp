// bs_maskgen_shim.cpp — the C++ face of the drop-in: bs_tensorflow_version / bs_maskgen_new /
// bs_maskgen_delete / bs_maskgen_process with the exact signatures of
// /root/reference/lib/libbackscrub.h:13-39, forwarding to the C ABI of libbsx.so (include/bsx.h).
//
// Compile with the application's OpenCV on the include path (or tests/cv_stub here) and link the
// object together with libbsx.so instead of the reference's libbackscrub.a:
//     g++ -std=c++17 -I include -c backscrub_amd/csrc/bs_maskgen_shim.cpp
//     g++ deepseg.o background.o loopback.o bs_maskgen_shim.o -L backscrub_amd -lbsx `pkg-config --libs opencv4`
//
// Behaviour kept from lib/libbackscrub.cc: nullptr on any creation failure after a message through
// ondebug/stderr (:191-233); process() returns false on a null context (:280); the output mask header
// aliases context-owned memory (:374) and stays valid until the next call; callbacks fire in the
// order prep, infer, mask on the calling thread (:303,311,363).  Added check: a frame whose size or
// type differs from the geometry given to new() returns false instead of throwing cv::Exception.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/bs_maskgen.h"
#include "../../include/bsx.h"

namespace {
struct ShimCtx {
  bsx_ctx* core = nullptr;
  int width = 0, height = 0;
  std::vector<unsigned char> host_mask;  // what `mask` aliases after process()
};
}  // namespace

const char* bs_tensorflow_version(void) { return bsx_version(); }

void* bs_maskgen_new(const std::string& modelname, size_t threads, size_t width, size_t height,
                     void (*ondebug)(void* ctx, const char* msg), void (*onprep)(void* ctx), void (*oninfer)(void* ctx),
                     void (*onmask)(void* ctx), void* caller_ctx) {
  ShimCtx* s = new ShimCtx;
  int device = 0;
  if (const char* e = getenv("BSX_DEVICE")) device = atoi(e);
  s->core = bsx_new(modelname.c_str(), threads, width, height, /*n_streams=*/1, device, ondebug, onprep, oninfer, onmask, caller_ctx);
  if (!s->core) { delete s; return nullptr; }
  s->width = (int)width;
  s->height = (int)height;
  s->host_mask.assign(width * height, 255);  // ctx.mask starts all-background (lib/libbackscrub.cc:248)
  return s;
}

void bs_maskgen_delete(void* context) {
  if (!context) return;
  ShimCtx* s = static_cast<ShimCtx*>(context);
  bsx_delete(s->core);
  delete s;
}

bool bs_maskgen_process(void* context, cv::Mat& frame, cv::Mat& mask) {
  if (!context) return false;
  ShimCtx* s = static_cast<ShimCtx*>(context);
  if (frame.empty() || frame.type() != CV_8UC3 || frame.cols != s->width || frame.rows != s->height) return false;
  int rc = bsx_process_host(s->core, 0, frame.data, frame.step[0], s->host_mask.data(), (size_t)s->width);
  if (rc != BSX_OK) return false;
  mask = cv::Mat(s->height, s->width, CV_8UC1, s->host_mask.data());
  return true;
}
