// HBM read rate of the GEMM kernels' A-operand access pattern, without the GEMM (run on the GPU box):
//   a workgroup owns a block of 128 rows x K floats and reads it in passes of 16 KB = R rows x PW floats (R = 4096 / PW).
//   PW = 32 is what pw_gemm_f16s_k / pw_gemm_ring_k do (128 rows x 128 B per K slab); PW = K reads the block front to back.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_rows tools/microbench_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcd_wg() {
  const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, local = orig >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int PW, bool BARRIER>
__global__ __launch_bounds__(256) void rows_k(const float* __restrict__ x, float* __restrict__ out, long M, int K) {
  extern __shared__ float pad[];
  const long m_base = (long)xcd_wg() * 128;
  constexpr int CPR = PW / 4, R = 4096 / PW;        // float4 chunks per row and rows per pass (16 KB per pass, 4 float4 per lane)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rg = 0; rg < 128; rg += R)
    for (int k0 = 0; k0 + PW <= K; k0 += PW) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int f = threadIdx.x + i * 256, row = f / CPR, c = f - row * CPR;
        const long m = min(m_base + rg + row, M - 1);
        v[i] = *reinterpret_cast<const float4*>(x + m * K + k0 + 4 * c);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      if (BARRIER) __syncthreads();
    }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = acc.x + pad[0];
}

template <int PW, bool BARRIER>
void run(const char* name, const float* x, float* out, long M, int K, size_t lds) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(rows_k<PW, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)((M + 127) / 128);
  const int Ku = K / PW * PW;
  for (int w = 0; w < 2; w++) rows_k<PW, BARRIER><<<grid, 256, lds>>>(x, out, M, K);
  hipEventRecord(a);
  const int it = 5;
  for (int i = 0; i < it; i++) rows_k<PW, BARRIER><<<grid, 256, lds>>>(x, out, M, K);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  ms /= it;
  printf("%-44s K=%4d lds/WG=%3zu KB  %8.3f ms  %7.1f GB/s\n", name, K, lds >> 10, ms, (double)M * Ku * 4 / ms * 1e-6);
}

// ---- store side: a workgroup writes a tile of 128 rows x TC floats of an [M][N] f32 matrix, the way the GEMM epilogue does (16-byte stores;
// one wave instruction = RPI rows x (64 / RPI) lanes x 16 B):  RPI = 16 → 64-byte pieces (the epilogue), 8 → 128-byte pieces, 4 → 256-byte pieces
template <int RPI>
__global__ __launch_bounds__(256) void tile_store_k(float* __restrict__ y, long M, int N, int TC) {
  const unsigned ncol = N / TC, wg = xcd_wg();
  const long row_blk = wg / ncol;
  const int col0 = (int)(wg - row_blk * ncol) * TC;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int LPR = 64 / RPI;                      // lanes per row
  const int r = lane / LPR, c = (lane % LPR) * 4;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
  // wave: 32 rows; instruction (i, j): rows 32 wave + RPI i + r, columns col0 + 4 LPR j + c
  for (int i = 0; i < 32 / RPI; i++)
    for (int j = 0; j < TC / (4 * LPR); j++) {
      const long m = row_blk * 128 + 32 * wave + RPI * i + r;
      if (m < M) *reinterpret_cast<float4*>(y + m * N + col0 + 4 * LPR * j + c) = v;
    }
}
template <int RPI>
void run_store(const char* name, float* y, long M, int N, int TC) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)((M + 127) / 128) * (N / TC);
  for (int w = 0; w < 2; w++) tile_store_k<RPI><<<grid, 256>>>(y, M, N, TC);
  hipEventRecord(a);
  const int it = 5;
  for (int i = 0; i < it; i++) tile_store_k<RPI><<<grid, 256>>>(y, M, N, TC);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  ms /= it;
  printf("%-44s N=%4d tile=%3d  %8.3f ms  %7.1f GB/s\n", name, N, TC, ms, (double)M * N * 4 / ms * 1e-6);
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 480;
  const long M = 1024L * 1089;
  float *x, *out;
  hipMalloc(&x, (size_t)M * K * 4 + 4096);
  hipMalloc(&out, 64 << 20);
  hipMemset(x, 0, (size_t)M * K * 4 + 4096);
  for (size_t lds : {(size_t)38 << 10, (size_t)78 << 10}) {         // 4 or 2 workgroups per CU
    run<32, true>("128 rows x 128 B per pass (the GEMMs), barrier", x, out, M, K, lds);
    run<32, false>("128 rows x 128 B per pass, no barrier", x, out, M, K, lds);
    if (K % 96 == 0) run<96, true>("  rows x 384 B per pass, barrier", x, out, M, K, lds);
    if (K % 160 == 0) run<160, true>("  rows x 640 B per pass, barrier", x, out, M, K, lds);
    if (K == 480) run<480, true>("whole rows, front to back, barrier", x, out, M, K, lds);
    if (K == 480) run<480, false>("whole rows, front to back, no barrier", x, out, M, K, lds);
  }
  run<32, false>("128 rows x 128 B per pass, no barrier, 8 WG/CU", x, out, M, K, 16 << 10);
  if (K == 480) run<480, false>("whole rows, no barrier, 8 WG/CU", x, out, M, K, 16 << 10);
  // stores (x is reused as the output: M x N floats with N <= K)
  for (int N : {256, 80}) {
    const int TC = N == 256 ? 64 : 80;
    if (N > K) continue;
    run_store<16>("store 16 rows x 64 B per instruction", x, M, N, TC);
    if (TC % 32 == 0) run_store<8>("store 8 rows x 128 B per instruction", x, M, N, TC);
    if (TC % 64 == 0) run_store<4>("store 4 rows x 256 B per instruction", x, M, N, TC);
  }
  run_store<4>("store whole rows, N = tile = 256", x, M, 256, 256);
  return 0;
}
