// debug: yuyv4_to_bgr3 and blend_quad on the GPU against host arithmetic
#include "../backscrub_amd/csrc/kernels_img.hip"
#include <cstdio>
#include <vector>
namespace bsx { namespace {
__global__ void conv_k(const uint32_t* in, uint32_t* out, const uint32_t* bgw, const uint32_t* mw, uint32_t* outb, uint32_t* outc, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  uint32_t o[3]; yuyv4_to_bgr3(in[2 * i], in[2 * i + 1], o);
  out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
  uint32_t a[3] = {bgw[3 * i], bgw[3 * i + 1], bgw[3 * i + 2]}, r[3];
  blend_quad(a, o, mw[i], r);
  outb[3 * i] = r[0]; outb[3 * i + 1] = r[1]; outb[3 * i + 2] = r[2];
  // the same blend with the frame words taken from memory (what the BGR path does)
  uint32_t f[3] = {outc[3 * i], outc[3 * i + 1], outc[3 * i + 2]}, r2[3];
  blend_quad(a, f, mw[i], r2);
  outc[3 * i] = r2[0]; outc[3 * i + 1] = r2[1]; outc[3 * i + 2] = r2[2];
}
} }
static void host_conv(uint32_t p, uint8_t* o6) {
  const int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
  int y0 = p & 255, u = (int)((p >> 8) & 255) - 128, y1 = (p >> 16) & 255, v = (int)(p >> 24) - 128;
  int ruv = (1 << 19) + CVR * v, guv = (1 << 19) + CVG * v + CUG * u, buv = (1 << 19) + CUB * u;
  int ya = std::max(0, y0 - 16) * CY, yb = std::max(0, y1 - 16) * CY;
  auto c = [](int x) { return (uint8_t)std::min(std::max(x >> 20, 0), 255); };
  o6[0] = c(ya + buv); o6[1] = c(ya + guv); o6[2] = c(ya + ruv); o6[3] = c(yb + buv); o6[4] = c(yb + guv); o6[5] = c(yb + ruv);
}
int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> in(2 * n), bg(3 * n), mw(n), fr(3 * n);
  unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  for (auto& x : in) x = rnd() ^ (rnd() >> 7); for (auto& x : bg) x = rnd() ^ (rnd() >> 9); for (auto& x : mw) x = rnd() ^ (rnd() >> 5);
  std::vector<uint8_t> want(12 * n);
  for (int i = 0; i < n; i++) { host_conv(in[2 * i], &want[12 * i]); host_conv(in[2 * i + 1], &want[12 * i + 6]); }
  memcpy(fr.data(), want.data(), 12 * n);
  uint32_t *d_in, *d_out, *d_bg, *d_mw, *d_ob, *d_oc;
  hipMalloc(&d_in, 8 * n); hipMalloc(&d_out, 12 * n); hipMalloc(&d_bg, 12 * n); hipMalloc(&d_mw, 4 * n); hipMalloc(&d_ob, 12 * n); hipMalloc(&d_oc, 12 * n);
  hipMemcpy(d_in, in.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(d_bg, bg.data(), 12 * n, hipMemcpyHostToDevice); hipMemcpy(d_mw, mw.data(), 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(d_oc, fr.data(), 12 * n, hipMemcpyHostToDevice);
  bsx::conv_k<<<n / 256, 256>>>(d_in, d_out, d_bg, d_mw, d_ob, d_oc, n);
  std::vector<uint8_t> got(12 * n), gb(12 * n), gc(12 * n);
  hipMemcpy(got.data(), d_out, 12 * n, hipMemcpyDeviceToHost); hipMemcpy(gb.data(), d_ob, 12 * n, hipMemcpyDeviceToHost); hipMemcpy(gc.data(), d_oc, 12 * n, hipMemcpyDeviceToHost);
  long bad = 0, badb = 0; long hist[12] = {0}, histb[12] = {0};
  for (long i = 0; i < 12L * n; i++) { if (got[i] != want[i]) { bad++; hist[i % 12]++; if (bad < 6) printf("conv byte %ld: got %d want %d\n", i % 12, got[i], want[i]); } if (gb[i] != gc[i]) { badb++; histb[i % 12]++; if (badb < 6) printf("blend byte %ld: yuyv %d bgr %d\n", i % 12, gb[i], gc[i]); } }
  printf("conversion mismatches %ld, blend mismatches %ld\n", bad, badb);
  for (int k = 0; k < 12; k++) printf("%ld/%ld ", hist[k], histb[k]); printf("\n");
  return 0;
}
