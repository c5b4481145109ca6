// What is the HBM ceiling for a READ/WRITE MIX on this part?  (run on the GPU box; profiles/r04*_microbench_mix.txt)
//   tools/microbench_rows.hip showed 6.2-6.4 TB/s for pure reads and pure writes, yet the two kernels of this repository that stream a read/write mix at full
//   tilt (blend16_k: 7 B read + 3 B written per pixel; the fused mask + blend: 7 read + 4 written) both stop at ~4.7-4.8 TB/s of counted traffic.  This file
//   measures plain streaming kernels with the same mixes and nothing to compute: if THEY stop at the same rate, the compositor is at the memory system's
//   mixed-traffic ceiling and no kernel work can move it; if they run at 6.3, the compositor has headroom.
// Patterns (16-byte accesses, grid-stride over 2.4 GB arrays, 256-lane workgroups, nontemporal stores like the product kernels):
//   read1        s = Σ a[i]                          R:W = 1:0
//   write1       o[i] = c                            0:1
//   copy11       o[i] = a[i]                         1:1
//   mix21        o[i] = a[i] ^ b[i]                  2:1   (the blend: frame + background in, composite out — background per stream)
//   mix74        o[i] = a..; 7 units read, 4 written 7:4   (the fused mask + blend with a per-stream background)
//   mix41s       o[i] = a[i] ^ b[i % small]          the shared 0.9 MB background: 4 B/px from HBM, 3 from L2, 4 written
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_mix tools/microbench_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, bool SHARED_B>
__global__ __launch_bounds__(256) void mix_k(const u4* __restrict__ a, const u4* __restrict__ b, u4* __restrict__ o, long n, long small_n, unsigned* sink) {
  // unit = one 16-byte vector per array slot; a lane handles, per iteration, NR read units and NW written units of its own contiguous region
  const long stride = (long)gridDim.x * 256;
  u4 acc = {0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    u4 v[NR > 0 ? NR : 1];
#pragma unroll
    for (int r = 0; r < NR; r++) {
      if (SHARED_B && r == NR - 1) v[r] = b[i % small_n];
      else v[r] = __builtin_nontemporal_load(&a[i + (long)r * n]);
    }
    u4 x = {(unsigned)i, 1u, 2u, 3u};
#pragma unroll
    for (int r = 0; r < NR; r++) x ^= v[r];
    if (NW == 0) acc ^= x;
#pragma unroll
    for (int w = 0; w < NW; w++) __builtin_nontemporal_store(x, &o[i + (long)w * n]);
  }
  if (NW == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc.x;
}

template <int NR, int NW, bool SHARED_B>
void run(const char* name, const u4* a, const u4* b, u4* o, long n, long small_n, unsigned* sink, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) mix_k<NR, NW, SHARED_B><<<grid, 256>>>(a, b, o, n, small_n, sink);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; i++) mix_k<NR, NW, SHARED_B><<<grid, 256>>>(a, b, o, n, small_n, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  const double hbm_r = (double)(NR - (SHARED_B ? 1 : 0)) * n * 16, all_r = (double)NR * n * 16, w = (double)NW * n * 16;
  printf("%-10s grid %6d  %8.3f ms   algorithmic %7.1f GB/s   HBM-side (shared operand from L2) %7.1f GB/s   R:W = %d:%d\n", name, grid, ms, (all_r + w) / ms * 1e-6,
         (hbm_r + w) / ms * 1e-6, NR, NW);
}

int main() {
  const long n = 12L << 20;                          // 12 Mi units of 16 B = 192 MiB per array slot
  u4 *a, *b, *o;
  unsigned* sink;
  hipMalloc(&a, (size_t)n * 16 * 7);
  hipMalloc(&b, 1 << 20);
  hipMalloc(&o, (size_t)n * 16 * 4);
  hipMalloc(&sink, 4);
  hipMemset(a, 1, (size_t)n * 16 * 7);
  hipMemset(b, 2, 1 << 20);
  const long small_n = (900 * 1024) / 16;
  for (int grid : {2048, 8192, 32768}) {
    run<1, 0, false>("read1", a, b, o, n, small_n, sink, grid);
    run<0, 1, false>("write1", a, b, o, n, small_n, sink, grid);
    run<1, 1, false>("copy11", a, b, o, n, small_n, sink, grid);
    run<2, 1, false>("mix21", a, b, o, n, small_n, sink, grid);
    run<7, 4, false>("mix74", a, b, o, n / 2, small_n, sink, grid);
    run<2, 1, true>("mix21s", a, b, o, n, small_n, sink, grid);
    run<4, 2, false>("mix42", a, b, o, n, small_n, sink, grid);
  }
  return 0;
}
