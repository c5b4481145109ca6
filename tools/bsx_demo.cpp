// bsx_demo.cpp — a tiny C++ application written against the reference's library interface
// (bs_maskgen_new / process / delete, as CalcMask does in app/deepseg.cc:203,246,269).
//   bsx_demo <model.tflite> <width> <height> <frames.bgr> <n_frames> <masks.out>
// Reads n raw BGR frames, feeds them through ONE context in order, writes the n masks.
// Used by tests/test_shim.py to check the C++ drop-in path without Python/torch in the process.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "bs_maskgen.h"

static int g_events = 0;
static void on_debug(void*, const char* msg) { fprintf(stderr, "[debug] %s", msg); }
static void on_stage(void* p) { ++*static_cast<int*>(p); }

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: %s model w h frames.bgr n masks.out\n", argv[0]); return 2; }
  int w = atoi(argv[2]), h = atoi(argv[3]), n = atoi(argv[5]);
  printf("engine: %s\n", bs_tensorflow_version());
  void* ctx = bs_maskgen_new(argv[1], 2, w, h, on_debug, on_stage, on_stage, on_stage, &g_events);
  if (!ctx) { fprintf(stderr, "bs_maskgen_new failed\n"); return 3; }
  FILE* fi = fopen(argv[4], "rb");
  FILE* fo = fopen(argv[6], "wb");
  if (!fi || !fo) { fprintf(stderr, "cannot open files\n"); return 4; }
  std::vector<unsigned char> buf((size_t)w * h * 3);
  for (int i = 0; i < n; i++) {
    if (fread(buf.data(), 1, buf.size(), fi) != buf.size()) { fprintf(stderr, "short read\n"); return 5; }
    cv::Mat frame(h, w, CV_8UC3, buf.data());
    cv::Mat mask;
    if (!bs_maskgen_process(ctx, frame, mask)) { fprintf(stderr, "process failed\n"); return 6; }
    if (mask.rows != h || mask.cols != w || mask.type() != CV_8UC1) { fprintf(stderr, "bad mask header\n"); return 7; }
    for (int y = 0; y < h; y++) fwrite(mask.data + (size_t)y * mask.step[0], 1, w, fo);
  }
  fclose(fi); fclose(fo);
  cv::Mat wrong(h / 2, w, CV_8UC3), m2;
  if (bs_maskgen_process(ctx, wrong, m2)) { fprintf(stderr, "size mismatch not rejected\n"); return 8; }
  bs_maskgen_delete(ctx);
  bs_maskgen_delete(nullptr);
  if (bs_maskgen_process(nullptr, wrong, m2)) return 9;
  printf("ok frames=%d callbacks=%d\n", n, g_events);
  return g_events == 3 * n ? 0 : 10;
}
